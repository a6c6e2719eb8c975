#!/bin/bash
# rocprofv3 kernel trace + stats of the V-JEPA bench step (GPU box, repo root):  tools/profile_vjepa.sh <tag>
set -e
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_vjepa_$TAG
mkdir -p "$OUT"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o step -- python3 "$ROOT/bench.py" --vjepa --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/gpurun_out/${TAG}_vjepa_kernel_stats.csv"
python3 - "$ROOT/gpurun_out/${TAG}_vjepa_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6/4:.2f} ms per step (4 steps)")
for r in rows[:26]:
    print(f'{int(r["TotalDurationNs"])/1e6/4:8.3f} ms/step  {int(r["Calls"])/4:7.1f} calls/step  {float(r["AverageNs"])/1e3:9.1f} us avg  {r["Name"][:90]}')
PY
TRACE=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/kernel_by_shape.py" "$TRACE" 4 > "$ROOT/gpurun_out/${TAG}_vjepa_kernel_by_shape.txt" || true
head -40 "$ROOT/gpurun_out/${TAG}_vjepa_kernel_by_shape.txt"; tail -9 "$ROOT/gpurun_out/${TAG}_vjepa_kernel_by_shape.txt"
