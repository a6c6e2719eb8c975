#!/bin/bash
# round 6: race screen + random-shape fuzz + wide configuration sweep on the build with the re-keyed split-launch state
set -e
mkdir -p gpurun_out
(timeout -k 10 500 python tools/step_soak.py 24 > gpurun_out/r06_step_soak.txt 2>&1; echo "rc $?" >> gpurun_out/r06_step_soak.txt) || true
tail -12 gpurun_out/r06_step_soak.txt
(timeout -k 10 300 python tools/fuzz_round5.py 60 6 > gpurun_out/r06_fuzz_round5.txt 2>&1; echo "rc $?" >> gpurun_out/r06_fuzz_round5.txt) || true
tail -5 gpurun_out/r06_fuzz_round5.txt
(SMBV_SWEEP_SCALE=4 timeout -k 10 600 python -m pytest tests/test_step_sweep_gpu.py -x -q -m gpu > gpurun_out/r06_sweep_wide.txt 2>&1; echo "rc $?" >> gpurun_out/r06_sweep_wide.txt) || true
tail -4 gpurun_out/r06_sweep_wide.txt
