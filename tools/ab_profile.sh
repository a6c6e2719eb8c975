#!/bin/bash
# kernel-stat comparison of two debug variants of the bench step (GPU box, repo root): tools/ab_profile.sh <variantA> <variantB>
set -e
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  OUT=$ROOT/gpurun_out/prof_ab_$V
  mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o step -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs --debug-variant $V > "$OUT/bench.json" 2> "$OUT/bench.err"
  python3 "$ROOT/tools/kernel_by_shape.py" "$(find "$OUT" -name '*kernel_trace.csv' | head -1)" 4 > "$ROOT/gpurun_out/ab_by_shape_$V.txt"
  echo "== variant $V"; head -24 "$ROOT/gpurun_out/ab_by_shape_$V.txt"; tail -9 "$ROOT/gpurun_out/ab_by_shape_$V.txt"
done
