#!/bin/bash
set -e
STEPS=20 timeout -k 10 900 tools/step_ab2.sh 0:0 2:0 4:0 6:0 > gpurun_out/r05e_ab.txt 2>&1
cat gpurun_out/r05e_ab.txt
timeout -k 10 600 tools/attn_bwd_traffic.sh > gpurun_out/r05_attn_bwd_traffic.log 2>&1 || true
cat gpurun_out/attn_bwd_traffic.json 2>/dev/null | head -30
rm -rf gpurun_out/pmc_attn_bwd
