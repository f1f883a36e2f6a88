#!/bin/bash
# usage: tools/onewave_pmc.sh <lib.so> [N]  -- one-wave latency probe + instruction counters per wave-step
LIB=$1; N=${2:-64}
cd /tmp; export TMPDIR=/tmp
export MCS_HIP_LIB=$LIB
python $GRAFT_REPO_ROOT/tools/gpu_onewave.py $N | tail -1
rm -rf /tmp/pmc_one
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY --output-format csv -d /tmp/pmc_one -- python $GRAFT_REPO_ROOT/tools/gpu_onewave.py $N > /tmp/pmc_one.log 2>&1
W=$(grep "wave-steps" /tmp/pmc_one.log | tail -1 | sed "s/.*max helix)=\([0-9]*\).*/\1/")
python3 - <<PY
import csv, glob
W=float("$W")
for p in glob.glob("/tmp/pmc_one/*/*counter_collection.csv"):
    rows=[r for r in csv.DictReader(open(p)) if "transport" in r["Kernel_Name"]]
    last=max(int(r["Dispatch_Id"]) for r in rows)
    agg={}
    for r in rows:
        if int(r["Dispatch_Id"])==last: agg[r["Counter_Name"]]=agg.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    print("$LIB per wave-step:", {k.replace("SQ_",""): round(v/W,1) for k,v in agg.items() if k!="SQ_WAVES"})
PY
