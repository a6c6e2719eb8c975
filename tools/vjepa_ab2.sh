#!/bin/bash
# in-box A/B of the V-JEPA step over SMBV_DEBUG_AB words (GPU box, repo root): tools/vjepa_ab2.sh 0 8 ...   (4 = last grouped weight-gradient launch whole,
# 8 = rotary embedding as launches of its own); two interleaved passes
set -e
for PASS in 1 2; do for AB in "$@"; do
  SMBV_DEBUG_AB=$AB python3 bench.py --vjepa --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $PASS vjepa ab $AB ms_per_step %.3f loss %.6f' % (d['ms_per_step'], d['config'].get('final_loss', float('nan'))))"
done; done
