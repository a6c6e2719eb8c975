#!/bin/bash
# Per-kernel counter table of one MIM bench step (GPU box, repo root):  tools/step_budget_pmc.sh <tag>
# Two rocprofv3 passes over the same short bench run (PMC + kernel trace only; counters serialise the dispatches, so durations are the kernels' own):
#   pass 1  SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES
#   pass 2  GRBM_GUI_ACTIVE  (effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration, MI355X guide 'DVFS give-back')
# Table: gpurun_out/<tag>_step_budget.txt  (per kernel + launch grid: calls/step, us per launch, clock, matrix-pipe busy, VALU issue, waits, LDS)
set -e
TAG=${1:-r03}
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_step_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq" -o s -- python3 "$ROOT/bench.py" $ARGS > "$OUT/sq.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/grbm" -o g -- python3 "$ROOT/bench.py" $ARGS > "$OUT/grbm.log" 2>&1
python3 - "$OUT" "$ROOT/gpurun_out/${TAG}_step_budget.txt" <<'PY'
import csv, glob, sys, collections
out, dst = sys.argv[1], sys.argv[2]
def load(sub):
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))   # dispatch id -> counter -> value
    meta = {}
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            d = r["Dispatch_Id"]
            cnt[d][r["Counter_Name"]] += float(r["Counter_Value"])
            meta[d] = (r["Kernel_Name"], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
    dur = {}
    for f in glob.glob(f"{out}/{sub}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return cnt, meta, dur
def short(n):
    return n.split("(")[0].replace("void smbv::", "").replace("smbv::", "")
sq, meta, dur = load("sq")
gr, gmeta, gdur = load("grbm")
clk = collections.defaultdict(list)
for d, c in gr.items():
    if d in gdur and gdur[d] > 0 and "GRBM_GUI_ACTIVE" in c:
        clk[(short(gmeta[d][0]), gmeta[d][1])].append(c["GRBM_GUI_ACTIVE"] / 8.0 / gdur[d])   # cycles per ns = GHz
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for d, c in sq.items():
    k = (short(meta[d][0]), meta[d][1])
    for n, v in c.items():
        tab[k][n].append(v)
    tab[k]["dur"].append(dur.get(d, 0))
STEPS = 3.0
rows = []
for k, c in tab.items():
    n = len(c["dur"])
    tot = sum(c["dur"])
    m = lambda name: sum(c[name]) / max(1, len(c[name]))
    w = m("SQ_WAVE_CYCLES") or 1.0
    g = sorted(clk.get(k, [0.0]))
    ghz = g[len(g) // 2]
    avg_ns = tot / n
    # matrix-pipe busy: SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1,024 SIMDs, in cycles (guide: = 32 x N_mfma for 32x32x16 bf16)
    mfma = m("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * avg_ns * ghz) if ghz > 0 else 0.0
    rows.append((tot / STEPS / 1e3, n / STEPS, avg_ns / 1e3, ghz, mfma, m("SQ_ACTIVE_INST_VALU") / w, m("SQ_WAIT_ANY") / w, m("SQ_WAIT_INST_ANY") / w, m("SQ_LDS_IDX_ACTIVE") / w, k))
rows.sort(reverse=True)
with open(dst, "w") as f:
    f.write("# one MIM step under rocprofv3 --pmc (dispatches serialised: each kernel alone on the chip); clock = GRBM_GUI_ACTIVE / 8 / duration (reads high below ~0.3 ms)\n")
    f.write("#   us/step calls/step   us avg  clock GHz  MFMA pipe busy  VALU/wave-cycles  WAIT_ANY  WAIT_INST  LDS_IDX  kernel (grid threads)\n")
    for r in rows[:40]:
        f.write(f"{r[0]:10.1f} {r[1]:8.1f} {r[2]:9.1f} {r[3]:9.2f} {r[4]:14.2f} {r[5]:12.2f} {r[6]:12.2f} {r[7]:9.2f} {r[8]:8.2f}  {r[9][0]} ({r[9][1]})\n")
print(open(dst).read())
PY
