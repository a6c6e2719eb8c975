#!/bin/bash
# in-box A/B of whole-step time between debug variants (GPU box, repo root): tools/step_ab.sh <variant> <variant> ...
# every variant is run twice, interleaved, so box-to-box and warm-up drift shows up as a difference between the two passes
set -e
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4 || true
for PASS in 1 2; do
  for V in "$@"; do
    python3 bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs --debug-variant $V 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $PASS variant $V ms_per_step %.3f' % d['ms_per_step'])"
  done
done
