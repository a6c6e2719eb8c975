#!/bin/bash
# scratch: head / encoder_to_decoder weight gradients on the side stream - step + DP tests, then the A/B (SMBV_DEBUG_AB bit 16 = on the caller's stream)
set -e
timeout -k 10 800 python3 -m pytest tests/test_mim_step_gpu.py tests/test_accum_dp_gpu.py tests/test_step_sweep_gpu.py -m gpu -x -q > gpurun_out/r05aa_tests.log 2>&1 || { tail -30 gpurun_out/r05aa_tests.log; exit 1; }
tail -1 gpurun_out/r05aa_tests.log
STEPS=30 tools/step_ab2.sh 0:0 16:0 | tee gpurun_out/r05aa_ab.txt
