#!/bin/bash
# instruction-class mix per wave-step for the one-wave probe
N=${1:-1}
cd /tmp; export TMPDIR=/tmp
for P in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
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
    print("N=$N W=%d per wave-step:"%W, {k.replace("SQ_INSTS_","").replace("SQ_",""): round(v/W,1) for k,v in agg.items()})
PY
done
