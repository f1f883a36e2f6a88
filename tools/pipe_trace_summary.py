"""Overlap figures from a rocprofv3 kernel trace of tools/gpu_pipe_trace.py: for the transport launches of the LAST iteration in the trace, the
time during which one / two / three of them are in flight, and per launch role (main = the long launches on the main stream; side = the
launches that start while a main launch runs) the summed durations.  usage: pipe_trace_summary.py <kernel_trace.csv>"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    if "transport" in r["Kernel_Name"]:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "0")) or 0),
                     int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0)))
rows.sort()
t_first, t_last = rows[0][0], max(r[1] for r in rows)
half = [r for r in rows if r[0] >= (t_first + t_last) // 2] or rows           # the second iteration
ev = sorted([(s, 1) for s, e, *_ in half] + [(e, -1) for s, e, *_ in half])
depth, t_prev, busy = 0, ev[0][0], {}
for t, d in ev:
    busy[depth] = busy.get(depth, 0) + (t - t_prev)
    depth += d; t_prev = t
span = (max(r[1] for r in half) - half[0][0]) / 1e6
print(f"transport launches in the second half of the trace: {len(half)}; span {span:.2f} ms")
for k in sorted(busy):
    print(f"  {k} transport launch(es) in flight: {busy[k] / 1e6:8.2f} ms ({100 * busy[k] / 1e6 / span:5.1f} %)")
by_q = {}
for s, e, name, q, wg, grid in half:
    a = by_q.setdefault((q, name), [0, 0.0, 0])
    a[0] += 1; a[1] += (e - s) / 1e6; a[2] = max(a[2], grid // max(wg, 1))
print("per queue and kernel: launches, summed duration, largest grid (workgroups)")
for (q, name), (n, ms, g) in sorted(by_q.items(), key=lambda kv: -kv[1][1]):
    print(f"  queue {q:>3} {name:34s} {n:4d} launches {ms:9.2f} ms   <= {g} workgroups")
