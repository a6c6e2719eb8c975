#!/bin/bash
# rocprofv3 kernel trace + stats of bench.py's roofline leg ALONE (the decoder attention backward launches bench.py times with HIP events), so that the
# profiler's average duration of attn_bwd_pair_kernel and the bench's `roofline.launch_ms` come from the same process on the same box:
#   tools/profile_roofline.sh <tag>   ->  gpurun_out/<tag>_roofline_kernel_stats.csv, gpurun_out/<tag>_roofline.txt
set -e
TAG=${1:-r04}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_roofline_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o roofline -- python3 "$ROOT/tools/roofline_leg.py" > "$OUT/leg.json" 2> "$OUT/leg.err"
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/gpurun_out/${TAG}_roofline_kernel_stats.csv"
python3 - "$ROOT/gpurun_out/${TAG}_roofline_kernel_stats.csv" "$OUT/leg.json" > "$ROOT/gpurun_out/${TAG}_roofline.txt" <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
leg = json.load(open(sys.argv[2]))
print("# bench.py kernel_roofline() under rocprofv3 --kernel-trace --stats (tools/profile_roofline.sh): HIP-event times of the leg, then the profiler's per-kernel averages")
print("# (pair = parts 6: ONE attn_bwd_pair_kernel launch per call; dkv / dq: the two standalone kernels; 10 warm-up + 20 timed calls each)")
print(json.dumps(leg))
for r in rows:
    if "attn_" in r["Name"]:
        print(f'{float(r["AverageNs"]) / 1e3:9.1f} us avg  {int(r["Calls"]):5d} calls  min {float(r["MinNs"]) / 1e3:9.1f}  max {float(r["MaxNs"]) / 1e3:9.1f}  {r["Name"][:70]}')
PY
cat "$ROOT/gpurun_out/${TAG}_roofline.txt"
