#!/bin/bash
set -e
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_nt or bricks or splitk" > gpurun_out/r05d_tests.log 2>&1 || { tail -40 gpurun_out/r05d_tests.log; exit 1; }
tail -3 gpurun_out/r05d_tests.log
timeout -k 10 900 python -m pytest tests/test_mim_step_gpu.py tests/test_encoder_gpu.py tests/test_step_sweep_gpu.py -x -q -m gpu > gpurun_out/r05d_tests2.log 2>&1 || { tail -40 gpurun_out/r05d_tests2.log; exit 1; }
tail -3 gpurun_out/r05d_tests2.log
STEPS=20 timeout -k 10 900 tools/step_ab2.sh 0:0 2:0 4:0 6:0 > gpurun_out/r05d_ab.txt 2>&1
cat gpurun_out/r05d_ab.txt
tools/profile_step.sh r05d > gpurun_out/r05d_profile.log 2>&1
TRACE=$(find gpurun_out/prof_r05d -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $TRACE > gpurun_out/r05d_timeline.txt
rm -rf gpurun_out/prof_r05d
tail -8 gpurun_out/r05d_profile.log
