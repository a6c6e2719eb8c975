#!/bin/bash
# in-box A/B of whole-step time over (SMBV_DEBUG_AB, --debug-variant) pairs "ab:variant" (GPU box, repo root): tools/step_ab2.sh 0:0 1:0 0:16 ...
# every pair is run twice, interleaved (tools/step_ab.sh for the first word alone)
set -e
for PASS in 1 2; do
  for P in "$@"; do
    AB=${P%%:*}; V=${P##*:}
    SMBV_DEBUG_AB=$AB python3 bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs --debug-variant $V 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $PASS ab $AB variant $V ms_per_step %.3f' % d['ms_per_step'])"
  done
done
