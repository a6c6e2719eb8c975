#!/bin/bash
# in-box A/B of several BUILDS of the library on a kernel bench tool (GPU box, repo root):  LIBS="A B" TOOL="tools/attn_fwd_bench.py" tools/lib_kernel_ab.sh
set -e
cp smb_vision_amd/libsmbv.so /tmp/libsmbv_keep.so
for PASS in 1 2; do for L in ${LIBS:-A B}; do
  cp tools/ab_libs/libsmbv_$L.so smb_vision_amd/libsmbv.so
  echo "== pass $PASS lib $L"
  timeout -k 10 300 python3 ${TOOL:-tools/attn_fwd_bench.py} 2>&1 | grep -v Warning | head -${LINES_MAX:-12}
done; done
cp /tmp/libsmbv_keep.so smb_vision_amd/libsmbv.so
