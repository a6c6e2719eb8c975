#!/bin/bash
# in-box A/B of the attention backward's split sweeps (SMBV_PAIR_SPLIT = tiles of each role run as two half sweeps; 0 = none): tools/pair_split_ab.sh 0 128 192 256
set -e
for PASS in 1 2; do
  for X in "$@"; do
    SMBV_PAIR_SPLIT=$X python3 bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $PASS split $X ms_per_step %.3f loss %.6f' % (d['ms_per_step'], d['config']['final_loss']))"
  done
done
