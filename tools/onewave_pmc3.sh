#!/bin/bash
# stall-side counters per wave-step for the one-wave probe (N lanes live)
N=${1:-1}
cd /tmp; export TMPDIR=/tmp
for P in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INST_LEVEL_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SMEM"; do
rm -rf /tmp/pmc_x; rocprofv3 --pmc $P --output-format csv -d /tmp/pmc_x -- python $GRAFT_REPO_ROOT/tools/gpu_onewave.py $N > /tmp/pmc_x.log 2>&1
W=$(grep "wave-steps" /tmp/pmc_x.log | tail -1 | sed "s/.*max helix)=\([0-9]*\).*/\1/")
python3 - <<PY
import csv, glob
W=float("$W")
for p in glob.glob("/tmp/pmc_x/*/*counter_collection.csv"):
    rows=[r for r in csv.DictReader(open(p)) if "transport" in r["Kernel_Name"]]
    last=max(int(r["Dispatch_Id"]) for r in rows)
    agg={}
    for r in rows:
        if int(r["Dispatch_Id"])==last: agg[r["Counter_Name"]]=agg.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    print("N=$N W=%d per wave-step:"%W, {k.replace("SQ_",""): round(v/W,1) for k,v in agg.items()})
PY
done
