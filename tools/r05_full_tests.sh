#!/bin/bash
# the whole -m gpu suite with captured prints (measured worst values beside the bounds: DESIGN.md section 2), wall time reported
set -e
START=$(date +%s)
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu -rP --durations=25 > gpurun_out/r05_gpu_tests_full.log 2>&1 || { tail -60 gpurun_out/r05_gpu_tests_full.log; exit 1; }
END=$(date +%s)
echo "wall $((END - START)) s" >> gpurun_out/r05_gpu_tests_full.log
tail -4 gpurun_out/r05_gpu_tests_full.log
grep -n "worst\|full-size\|loss " gpurun_out/r05_gpu_tests_full.log | head -60
