#!/bin/bash
# round 5, first in-box A/B: (16) last encoder block deferred like the others vs inline; (1) scalar weight-preparation jobs vs one-read 64x64 jobs
set -e
python -m pytest tests/test_ops_gpu.py tests/test_mim_step_gpu.py tests/test_encoder_gpu.py -x -q -m gpu > gpurun_out/r05b_tests.log 2>&1 || { tail -30 gpurun_out/r05b_tests.log; exit 1; }
tail -3 gpurun_out/r05b_tests.log
STEPS=20 tools/step_ab.sh 0 16 1 17 > gpurun_out/r05b_ab.txt 2>&1
cat gpurun_out/r05b_ab.txt
tools/profile_step.sh r05b > gpurun_out/r05b_profile.log 2>&1
TRACE=$(find gpurun_out/prof_r05b -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $TRACE > gpurun_out/r05b_timeline.txt
rm -rf gpurun_out/prof_r05b
tail -8 gpurun_out/r05b_profile.log
