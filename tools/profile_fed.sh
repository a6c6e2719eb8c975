#!/bin/bash
# rocprofv3 kernel stats of bench.py INCLUDING the fed leg (new volume + mask per step through VolumeFeeder): tools/profile_fed.sh <tag>
set -e
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_fed_$TAG
mkdir -p "$OUT"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o step -- python3 "$ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-trainer > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("prepare", "volume", "copyBuffer", "feed", "crop", "permute")):
        print(f'{int(r["TotalDurationNs"])/1e6:9.3f} ms total  {int(r["Calls"]):5d} calls  {float(r["AverageNs"])/1e3:9.1f} us avg  {r["Name"][:100]}')
PY
