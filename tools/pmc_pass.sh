#!/bin/bash
# usage: tools/pmc_pass.sh <outdir-under-gpurun_out> <N> <NPC> -- runs 3 rocprofv3 --pmc passes (counters only)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; N=$2; NPC=$3
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS"
P3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_LDS_BANK_CONFLICT"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/tools/gpu_run.py $N $NPC > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    rows = list(csv.DictReader(open(p)))
    agg = collections.OrderedDict()
    for r in rows:
        if "transport" not in r["Kernel_Name"]: continue
        k = (int(r["Dispatch_Id"]), r["Counter_Name"])
        agg[k] = agg.get(k, 0) + float(r["Counter_Value"])
    disp = sorted({k[0] for k in agg})
    names = []
    for k in agg:
        if k[1] not in names: names.append(k[1])
    print(p)
    print("dispatch," + ",".join(names))
    for d in disp:
        print(str(d) + "," + ",".join(f"{agg.get((d,n),0):.4g}" for n in names))
PY
