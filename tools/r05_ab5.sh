#!/bin/bash
set -e
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_mim_step_gpu.py -x -q -m gpu -k "bricks or golden or mid_size or determin or resum" > gpurun_out/r05f_tests.log 2>&1 || { tail -40 gpurun_out/r05f_tests.log; exit 1; }
tail -3 gpurun_out/r05f_tests.log
STEPS=20 timeout -k 10 900 tools/step_ab2.sh 0:0 12:0 > gpurun_out/r05f_ab.txt 2>&1
cat gpurun_out/r05f_ab.txt
