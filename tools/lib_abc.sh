#!/bin/bash
# in-box A/B of several BUILDS of the library (GPU box, repo root): tools/ab_libs/libsmbv_<X>.so copied over smb_vision_amd/libsmbv.so in turn, two interleaved passes.
#   LIBS="A B C" tools/lib_abc.sh [bench.py flags ...]
set -e
cp smb_vision_amd/libsmbv.so /tmp/libsmbv_keep.so
for PASS in 1 2; do for L in ${LIBS:-A B}; do
  cp tools/ab_libs/libsmbv_$L.so smb_vision_amd/libsmbv.so
  if [ "$PASS" = 1 ] && [ -n "$CHECK" ]; then
    timeout -k 10 400 python -m pytest tests/test_ops_gpu.py tests/test_mim_step_gpu.py -x -q -m gpu -k "$CHECK" 2>&1 | tail -1 | sed "s/^/lib $L tests: /"
  fi
  python3 bench.py --steps ${STEPS:-20} --warmup 4 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs "$@" 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $PASS lib $L ms_per_step %.3f loss %.6f' % (d['ms_per_step'], d['config']['final_loss']))"
done; done
cp /tmp/libsmbv_keep.so smb_vision_amd/libsmbv.so
