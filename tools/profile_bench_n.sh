#!/bin/bash
# Same passes as profile_bench.sh for another population size: tools/profile_bench_n.sh <tag> <particles>
set -e
TAG=$1; NP=$2; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --particles $NP --steps 1 --warmup 0 --no-cpu"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
timeout -k 10 500 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/busy -- $CMD > $OUT/busy.log 2>&1
python3 - <<PY
import csv, glob, collections
out = open("$OUT/summary.txt", "w")
def P(*a):
    print(*a); print(*a, file=out)
for p in glob.glob("$OUT/stats/*/*kernel_stats.csv"):
    P("== kernel stats (rocprofv3 --kernel-trace --stats) of: $CMD")
    P(open(p).read())
for name in ("fetch", "write", "busy"):
    for p in glob.glob("$OUT/%s/*/*counter_collection.csv" % name):
        agg = collections.defaultdict(float); n = collections.defaultdict(set)
        for r in csv.DictReader(open(p)):
            if "transport" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
        for k, v in agg.items():
            P(f"== pmc {k}: sum over {len(n[k])} mcs_k_transport launches = {v:.6g}; per launch = {v/len(n[k]):.6g}")
for name in ("stats", "fetch", "write", "busy"):
    for l in open("$OUT/%s.log" % name):
        if l.startswith("{"): P("== bench line under the %s pass:" % name, l.strip())
PY
