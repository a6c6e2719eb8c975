#!/bin/bash
# round-5 A/B of the token split of the last grouped weight-gradient launch (SMBV_DEBUG_AB bit 4 = off): MIM step and V-JEPA step, interleaved, two passes
set -e
STEPS=30 tools/step_ab2.sh 0:0 4:0 | tee gpurun_out/r05v_ab.txt
for PASS in 1 2; do for AB in 0 4; do
  SMBV_DEBUG_AB=$AB python3 bench.py --vjepa --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $PASS vjepa ab $AB ms_per_step %.3f' % d['ms_per_step'])"
done; done | tee -a gpurun_out/r05v_ab.txt
