#!/bin/bash
set -e
STEPS=20 tools/step_ab2.sh 0:0 1:0 0:16 > gpurun_out/r05c_ab.txt 2>&1
cat gpurun_out/r05c_ab.txt
tools/gemm_copy_rate.sh > gpurun_out/r05_gemm_copy_rate.txt 2>&1
cat gpurun_out/r05_gemm_copy_rate.txt
rm -rf gpurun_out/pmc_gemm_r05
