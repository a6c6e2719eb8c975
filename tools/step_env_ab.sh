#!/bin/bash
# in-box A/B of whole-step time between ENVIRONMENT settings (GPU box, repo root): tools/step_env_ab.sh "SMBV_ATTN_RESID=0" "SMBV_ATTN_RESID=1" ...
# every setting is run twice, interleaved (as tools/step_ab.sh does for debug variants); EXTRA="--vjepa" etc. selects another bench leg
for PASS in 1 2; do
  for E in "$@"; do
    env $E python3 bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs ${EXTRA} 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $PASS [$E] ms_per_step %.3f' % d['ms_per_step'])"
  done
done
