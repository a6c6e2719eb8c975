#!/bin/bash
# round 6: the round's new tests first (fail fast), then the whole -m gpu suite, then the default bench line
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "hand_off_words or scheduling_fault or split" > gpurun_out/r06_new_tests.log 2>&1 || { tail -60 gpurun_out/r06_new_tests.log; exit 1; }
tail -3 gpurun_out/r06_new_tests.log
timeout -k 10 200 python -m pytest tests/test_mim_step_gpu.py -x -q -m gpu -k "nothing_is_saved" >> gpurun_out/r06_new_tests.log 2>&1 || { tail -60 gpurun_out/r06_new_tests.log; exit 1; }
tail -3 gpurun_out/r06_new_tests.log
START=$(date +%s)
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r06_gpu_tests_full.log 2>&1 || { tail -60 gpurun_out/r06_gpu_tests_full.log; exit 1; }
END=$(date +%s)
echo "wall $((END - START)) s" >> gpurun_out/r06_gpu_tests_full.log
tail -4 gpurun_out/r06_gpu_tests_full.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_start.json 2> gpurun_out/r06_bench_start.err || { tail -30 gpurun_out/r06_bench_start.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/r06_bench_start.json')); print(d['ms_per_step'], d['value'], d['roofline']['launch_ms'], d['config'].get('vjepa_ms_per_step'), d['config'].get('vit_large_fp8_recompute_ms_per_step'))"
