"""Timeline of ONE steady-state train step from a rocprofv3 --kernel-trace CSV (tools/profile_step.sh keeps it): the step between the last two adamw_kernel
launches; every kernel with its start offset, duration, stream and the gap since the previous kernel ended on the SAME stream.
usage: step_timeline.py <..._kernel_trace.csv> [from_us] [to_us]      (window inside the step; default: whole step, long)"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void smbv::", "").replace("smbv::", ""), r.get("Stream_Id", "0"),
               int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in csv.DictReader(open(sys.argv[1]))), key=lambda t: t[0])
ends = [e for s, e, n, _, _ in rows if n.startswith("adamw_kernel")]
if len(ends) < 2:
    raise SystemExit("need two optimizer launches in the trace")
t0, t1 = ends[-2], ends[-1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e12
streams, last_end = {}, {}
print(f"# step of {(t1 - t0) / 1e3:.1f} us (end of adamw -> end of next adamw)")
print("#  start_us    dur_us   gap_us  stream  blocks  kernel")
for s, e, n, sid, blocks in rows:
    if e <= t0 or s >= t1:
        continue
    k = streams.setdefault(sid, len(streams))
    gap = (s - last_end[sid]) / 1e3 if sid in last_end else float("nan")
    last_end[sid] = e
    off = (s - t0) / 1e3
    if lo <= off <= hi:
        print(f"{off:10.1f} {(e - s) / 1e3:9.1f} {gap:8.1f}  s{k}  {blocks:6d}  {n[:70]}")
