#!/bin/bash
# scratch: attention forward with the ragged tile peeled - op tests, step tests, kernel timing, step time
set -e
timeout -k 10 800 python3 -m pytest tests/test_ops_gpu.py tests/test_mim_step_gpu.py tests/test_attention_plugin_gpu.py -m gpu -x -q > gpurun_out/r05ad_tests.log 2>&1 || { tail -30 gpurun_out/r05ad_tests.log; exit 1; }
tail -1 gpurun_out/r05ad_tests.log
PYTHONPATH=. python3 tools/attn_fwd_bench.py > gpurun_out/r05ad_fwd_bench.txt 2>&1 || true
tail -12 gpurun_out/r05ad_fwd_bench.txt
for PASS in 1 2; do
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('mim ms_per_step %.3f loss %.6f' % (d['ms_per_step'], d['config']['final_loss']))"
done | tee gpurun_out/r05ad_step.txt
python3 bench.py --vjepa --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('vjepa ms_per_step %.3f' % d['ms_per_step'])" | tee -a gpurun_out/r05ad_step.txt
