#!/bin/bash
# scratch: V-JEPA rotary fold - tests, A/B, fused profile
set -e
timeout -k 10 600 python3 -m pytest tests/test_vjepa_gpu.py -m gpu -x -q > gpurun_out/r05z_tests.log 2>&1 || { tail -30 gpurun_out/r05z_tests.log; exit 1; }
tail -1 gpurun_out/r05z_tests.log
STEPS=16 tools/vjepa_ab2.sh 0 8 | tee gpurun_out/r05z_ab.txt
tools/profile_vjepa.sh r05z_fused > gpurun_out/r05z_fused.log 2>&1
grep -h "gemm_nt_ring_kernel<3, [03]\|rope3d" gpurun_out/r05z_fused_vjepa_kernel_by_shape.txt | head -12
