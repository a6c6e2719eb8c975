#!/bin/bash
# HBM / fabric traffic of the dominant kernels with PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
# passes (4 TCC slots), --kernel-trace only; FETCH_SIZE reports half the bytes of wide streaming reads on gfx950 (doubled below),
# both counters are in KiB-like units of 1 KB as rocprofv3 prints them.  Run on the GPU box from the repo root.
set -e
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_attn_bwd
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 "$ROOT/tools/attn_bwd_traffic.py" > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 "$ROOT/tools/attn_bwd_traffic.py" > "$OUT/write.log" 2>&1
python3 - "$OUT" "$ROOT/gpurun_out/attn_bwd_traffic.json" <<'PY'
import csv, glob, json, sys
out, dst = sys.argv[1], sys.argv[2]
def collect(sub, counter):
    vals = {}
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            for name in ("attn_dkv2_kernel", "attn_dq2_kernel", "attn_bwd_pair_kernel"):
                if name in k:
                    vals.setdefault(name, []).append(float(r["Counter_Value"]))
    return vals
fetch, write = collect("fetch", "FETCH_SIZE"), collect("write", "WRITE_SIZE")
res = {"unit_note": "rocprofv3 FETCH_SIZE / WRITE_SIZE are reported in kilobytes; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B)", "kernels": {}}
for name in ("attn_dkv2_kernel", "attn_dq2_kernel", "attn_bwd_pair_kernel"):
    f = sorted(fetch.get(name, [0]))[len(fetch.get(name, [0])) // 2] * 1024 * 2
    w = sorted(write.get(name, [0]))[len(write.get(name, [0])) // 2] * 1024
    res["kernels"][name] = {"fetch_bytes": f, "write_bytes": w, "launches_seen": len(fetch.get(name, []))}
pk = res["kernels"]["attn_bwd_pair_kernel"]
res["traffic_bytes_per_launch"] = pk["fetch_bytes"] + pk["write_bytes"]
res["what"] = "decoder attention backward (S=20480, H=6): attn_bwd_pair_kernel (the shipped single launch), fabric-side bytes per launch; the two standalone kernels beside it"
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res))
PY
