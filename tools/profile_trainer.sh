#!/bin/bash
# rocprofv3 kernel stats of bench.py with the HF-Trainer leg: which torch kernels the harness adds per step.  tools/profile_trainer.sh <tag>
set -e
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_trainer_$TAG
mkdir -p "$OUT"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o step -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-feed > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "smbv::" not in r["Name"]:
        print(f'{int(r["TotalDurationNs"])/1e6:9.3f} ms total  {int(r["Calls"]):6d} calls  {float(r["AverageNs"])/1e3:9.1f} us avg  {r["Name"][:110]}')
PY
