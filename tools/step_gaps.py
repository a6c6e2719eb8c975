"""GPU idle time inside train steps, from a rocprofv3 --kernel-trace CSV: steps are delimited by the optimizer kernel (adamw_kernel); for the last N steps prints
the wall time per step, the time at least one kernel was running (union over all streams), the idle remainder and the launch count.
usage: step_gaps.py <..._kernel_trace.csv> [N]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda t: t[0])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ends = [e for s, e, n in rows if "adamw_kernel" in n]
if len(ends) < N + 1:
    raise SystemExit(f"only {len(ends)} optimizer launches in the trace")
t0, t1 = ends[-N - 1], ends[-1]
iv = [(max(s, t0), min(e, t1), n) for s, e, n in rows if e > t0 and s < t1]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - t0
print(f"last {N} steps: {wall / N / 1e6:.3f} ms/step wall, {busy / N / 1e6:.3f} ms/step with a kernel running, {(wall - busy) / N / 1e6:.3f} ms/step idle "
      f"({100.0 * (wall - busy) / wall:.1f} %), {len(iv) / N:.0f} launches/step, mean kernel {sum(e - s for s, e, _ in iv) / len(iv) / 1e3:.1f} us")
