#!/bin/bash
# SQ counter breakdown of the attention forward kernels (bf16 hand-phased vs fp8) at the model's shapes.  GPU box, repo root.
set -e
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_attn_r02
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/sq" -o s -- python3 "$ROOT/tools/attn_fp8_bench.py" > "$OUT/sq.log" 2>&1 || true
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_fwd" not in k:
            continue
        tab[(("fp8" if "fp8" in k else "bf16"), r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(tab):
    med = lambda v: sorted(v)[len(v) // 2]
    c = {n: med(v) for n, v in tab[key].items()}
    w = c["SQ_WAVE_CYCLES"]
    print(f"{key[0]:5s} grid {key[1]:>8s}  WAVE_CYCLES {w:.3g} WAIT_ANY {c.get('SQ_WAIT_ANY', 0) / w:.2f} WAIT_INST {c.get('SQ_WAIT_INST_ANY', 0) / w:.2f} ACTIVE {c.get('SQ_ACTIVE_INST_ANY', 0) / w:.2f} "
          f"VALU {c.get('SQ_ACTIVE_INST_VALU', 0) / w:.2f} LDS {c.get('SQ_LDS_IDX_ACTIVE', 0) / w:.2f} CONFLICT {c.get('SQ_LDS_BANK_CONFLICT', 0) / w:.3f} MFMA_BUSY/WAVE_CYCLES {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / w:.2f}")
PY
