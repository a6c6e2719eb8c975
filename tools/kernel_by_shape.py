"""Per kernel / launch grid / stream table of one bench step from a rocprofv3 --kernel-trace CSV (tools/profile_step.sh writes it):
usage: kernel_by_shape.py <..._kernel_trace.csv> [steps]   (steps = number of profiled steps the totals are divided by, default 4)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
streams = {}
acc = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void smbv::", "").replace("smbv::", "")
    sid = streams.setdefault(r.get("Stream_Id", "0"), len(streams))
    grid = tuple(int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"])) for a in "XYZ")
    k = (name, grid, sid)
    d = acc.setdefault(k, [0, 0])
    d[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    d[1] += 1
print("   us/step calls/step    us avg  stream  kernel  (blocks x, y, z)")
for (name, grid, sid), (ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{ns / 1e3 / steps:10.1f} {n / steps:10.1f} {ns / 1e3 / n:9.1f}  s{sid}  {name}  {grid}")
# union (wall) time of kernel families across streams
def union(pred):
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if pred(r["Kernel_Name"]))
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot / 1e6 / steps
fam = {"attention backward (pair | dkv2 | dq2 | delta)": lambda n: "attn_bwd_pair" in n or "attn_dkv" in n or "attn_dq" in n or "attn_delta" in n, "attention forward": lambda n: "attn_fwd" in n,
       "GEMMs (nt | tn | reduce_tn)": lambda n: "gemm_" in n or "reduce_tn" in n or "reduce_slabs" in n or "reduce_xsum" in n, "NT GEMMs": lambda n: "gemm_nt" in n,
       "TN GEMMs + reductions": lambda n: "gemm_tn" in n or "reduce_tn" in n or "reduce_slabs" in n or "reduce_xsum" in n,
       "LayerNorm (fwd | bwd | reduce_cols)": lambda n: "ln_" in n or "reduce_cols" in n, "optimizer + weight preparation": lambda n: "adamw" in n or "sqnorm" in n or "prep_weights" in n,
       "everything": lambda n: True}
print("# union (wall-clock) time per step of kernel families, all streams:")
for k, f in fam.items():
    print(f"#   {k}: {union(f):.2f} ms")
