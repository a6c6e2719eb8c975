#!/bin/bash
# in-box A/B of a runtime environment knob on the whole step (GPU box, repo root): tools/env_ab.sh
run() { python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 ms_per_step %.3f' % d['ms_per_step'])"; }
for P in 1 2; do
  run "pass $P default            "
  HIP_FORCE_DEV_KERNARG=1 run "pass $P HIP_FORCE_DEV_KERNARG=1"
  HIP_FORCE_DEV_KERNARG=0 run "pass $P HIP_FORCE_DEV_KERNARG=0"
done
env | grep -i "HIP_\|HSA_\|ROC\|AMD_" | head -20
