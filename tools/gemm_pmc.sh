#!/bin/bash
# PMC passes on the NT GEMM kernels (separate passes per counter block, --kernel-trace only).  Run on the GPU box from the repo root.
set -e
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_gemm_r02
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/tcc" -o t -- python3 "$ROOT/tools/gemm_pmc.py" > "$OUT/tcc.log" 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/sq" -o s -- python3 "$ROOT/tools/gemm_pmc.py" > "$OUT/sq.log" 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 "$ROOT/tools/gemm_pmc.py" > "$OUT/fetch.log" 2>&1 || true
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/tcp" -o p -- python3 "$ROOT/tools/gemm_pmc.py" > "$OUT/tcp.log" 2>&1 || true
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_nt" not in k:
            continue
        key = (("ring" if "ring" in k else "128x128"), r.get("Grid_Size", ""))
        tab[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{out}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_nt" in k:
            dur[(("ring" if "ring" in k else "128x128"), r.get("Grid_Size", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for key in sorted(tab):
    med = lambda v: sorted(v)[len(v) // 2]
    c = {n: med(v) for n, v in tab[key].items()}
    d = med(dur[key]) / 1e3 if dur.get(key) else float("nan")
    line = f"{key[0]:8s} grid {key[1]:>8s}  {d:7.1f} us"
    if "TCC_HIT_sum" in c:
        line += f"  L2 hit rate {c['TCC_HIT_sum'] / max(1.0, c['TCC_HIT_sum'] + c.get('TCC_MISS_sum', 0)):.3f} ({c['TCC_HIT_sum'] + c.get('TCC_MISS_sum', 0):.3g} requests)"
    if "TCP_TCC_READ_REQ_sum" in c:
        line += f"  L1->L2 read requests {c['TCP_TCC_READ_REQ_sum']:.3g}"
    if "FETCH_SIZE" in c:
        line += f"  fabric fetch {c['FETCH_SIZE'] * 2 * 1024 / 1e6:.1f} MB"
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        line += f"  WAIT_ANY {c.get('SQ_WAIT_ANY', 0) / w:.2f} WAIT_INST {c.get('SQ_WAIT_INST_ANY', 0) / w:.2f} ACTIVE {c.get('SQ_ACTIVE_INST_ANY', 0) / w:.2f} LDS_ACTIVE {c.get('SQ_LDS_IDX_ACTIVE', 0) / w:.2f} LDS_CONFLICT {c.get('SQ_LDS_BANK_CONFLICT', 0) / w:.3f} MFMA_BUSY/WAVE_CYCLES {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / w:.2f}"
    print(line)
PY
