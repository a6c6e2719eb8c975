#!/bin/bash
# scratch: reduce_cols rewrite - op + step tests, the step profile
set -e
timeout -k 10 800 python3 -m pytest tests/test_ops_gpu.py tests/test_mim_step_gpu.py -m gpu -x -q > gpurun_out/r05ab_tests.log 2>&1 || { tail -30 gpurun_out/r05ab_tests.log; exit 1; }
tail -1 gpurun_out/r05ab_tests.log
tools/profile_step.sh r05ab > gpurun_out/r05ab_profile.log 2>&1
grep "reduce_cols\|ln_bwd" gpurun_out/r05ab_step_kernel_by_shape.txt
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % d['ms_per_step'])"
