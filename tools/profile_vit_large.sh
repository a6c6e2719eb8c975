#!/bin/bash
# rocprofv3 kernel trace of BASELINE config 5's step (ViT-large, fp8 attention, per-layer recompute) on the GPU box, repo root:  tools/profile_vit_large.sh <tag>
set -e
TAG=${1:-r05}
OUT=$PWD/gpurun_out/prof_vitl_$TAG
mkdir -p "$OUT"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o step -- python3 "$ROOT/bench.py" --vit-large --fp8 --recompute --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/gpurun_out/${TAG}_vitl_kernel_stats.csv"
TRACE=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/kernel_by_shape.py" "$TRACE" 4 > "$ROOT/gpurun_out/${TAG}_vitl_kernel_by_shape.txt" || true
python3 "$ROOT/tools/step_gaps.py" "$TRACE" 2 >> "$ROOT/gpurun_out/${TAG}_vitl_kernel_by_shape.txt" || true
rm -rf "$OUT"
head -50 "$ROOT/gpurun_out/${TAG}_vitl_kernel_by_shape.txt"; tail -12 "$ROOT/gpurun_out/${TAG}_vitl_kernel_by_shape.txt"
