#!/bin/bash
# Instruction-cache behaviour of K1 over an iteration (rocprofv3 --pmc, one pass): requests, hits, misses, fetch stalls.
# usage (GPU box): tools/icache_pmc.sh <N> <npcuts> -> gpurun_out/ic/icache.txt
N=${1:-1000000}; NPC=${2:-30}
OUT=$GRAFT_REPO_ROOT/gpurun_out/ic; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for P in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_SALU"; do
rm -rf /tmp/pmc_ic; timeout -k 10 250 rocprofv3 --pmc $P --output-format csv -d /tmp/pmc_ic -- python $GRAFT_REPO_ROOT/tools/gpu_run.py $N $NPC > /tmp/pmc_ic.log 2>&1
python3 - <<PY >> $OUT/icache.txt
import csv, glob, collections
for p in glob.glob("/tmp/pmc_ic/*/*counter_collection.csv"):
    agg = collections.defaultdict(float); per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(p)):
        if "transport" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    print("sum over all K1 launches:", {k: "%.4g" % v for k, v in agg.items()})
    big = sorted(per, key=lambda d: -max(per[d].values()))[:6]
    for d in sorted(big): print("  dispatch", d, {k: "%.4g" % v for k, v in per[d].items()})
PY
done
tail -3 /tmp/pmc_ic.log >> $OUT/icache.txt
