#!/bin/bash
# in-box A/B of two BUILDS of the library (GPU box, repo root): tools/ab_libs/libsmbv_A.so / libsmbv_B.so are copied over smb_vision_amd/libsmbv.so in turn.
#   tools/lib_ab.sh [bench.py flags ...]      e.g.  tools/lib_ab.sh --vjepa
set -e
cp smb_vision_amd/libsmbv.so /tmp/libsmbv_keep.so
for PASS in 1 2; do for L in A B; do
  cp tools/ab_libs/libsmbv_$L.so smb_vision_amd/libsmbv.so
  python3 bench.py --steps ${STEPS:-20} --warmup 4 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs "$@" 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $PASS lib $L ms_per_step %.3f' % d['ms_per_step'])"
done; done
cp /tmp/libsmbv_keep.so smb_vision_amd/libsmbv.so
