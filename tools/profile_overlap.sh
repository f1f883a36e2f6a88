#!/bin/bash
# Kernel trace of the bench line with the overlapped leg: how much of the time two K1 launches are resident at once.
# usage (GPU box): tools/profile_overlap.sh <tag>  -> gpurun_out/<tag>/overlap_summary.txt
set -e
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 4 > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for p in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(p)):
        if "transport" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
out = open("$OUT/overlap_summary.txt", "w")
def P(*a):
    print(*a); print(*a, file=out)
P(f"== rocprofv3 --kernel-trace of: python bench.py --no-cpu --steps 4   ({len(rows)} K1 launches on {len(set(r[2] for r in rows))} streams / queues)")
# sweep line: time with >= 1 and with >= 2 K1 launches resident, separately for the sequential part (one stream) and the overlapped leg
ev = sorted([(s, 1) for s, e, q in rows] + [(e, -1) for s, e, q in rows])
busy1 = busy2 = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
P(f"== time with at least one K1 launch in flight: {busy1 / 1e6:.1f} ms; with two or more: {busy2 / 1e6:.1f} ms")
for l in open("$OUT/trace.log"):
    if l.startswith("{"): P("== bench line:", l.strip()[:300], "...", l.strip()[-330:])
PY
