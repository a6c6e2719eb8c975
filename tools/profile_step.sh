#!/bin/bash
# rocprofv3 kernel trace + stats of the bench step (run on the GPU box from the repo root):  tools/profile_step.sh <tag>
# writes gpurun_out/prof_<tag>/ ; copy <tag>_step_kernel_stats.csv into profiles/ for the record.
set -e
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o step -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/gpurun_out/${TAG}_step_kernel_stats.csv"
python3 - "$ROOT/gpurun_out/${TAG}_step_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6/4:.2f} ms per step (4 steps)")
for r in rows[:22]:
    print(f'{int(r["TotalDurationNs"])/1e6/4:8.3f} ms/step  {int(r["Calls"])//4:5d} calls/step  {float(r["AverageNs"])/1e3:9.1f} us avg  {r["Name"][:90]}')
PY
TRACE=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/kernel_by_shape.py" "$TRACE" 4 > "$ROOT/gpurun_out/${TAG}_step_kernel_by_shape.txt" || true
tail -12 "$ROOT/gpurun_out/${TAG}_step_kernel_by_shape.txt"
