#!/bin/bash
# Round profile at HEAD: rocprofv3 kernel-trace stats of the bench command plus separate --pmc passes (HBM traffic, SQ
# counters), for the headline size and for 1e7 particles.  usage (on the GPU box): tools/profile_round.sh <tag> [only1e6]
# -> gpurun_out/<tag>/summary_1e6.txt, summary_1e7.txt, traffic.json (copied to profiles/ by hand)
set -e
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run_passes () {   # $1 = label, $2.. = bench arguments
  L=$1; shift
  CMD="python $GRAFT_REPO_ROOT/bench.py $* --no-cpu --overlap 1 --long-draws 0"     # (--overlap 1, --long-draws 0: without the extra overlapped and pipelined legs, whose launches have another geometry / another kernel: the kernel statistics then cover exactly the launches of the warm-up and timed iterations)
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$L -- $CMD > $OUT/stats_$L.log 2>&1
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$L -- $CMD > $OUT/fetch_$L.log 2>&1
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$L -- $CMD > $OUT/write_$L.log 2>&1
  timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $OUT/sq_$L -- $CMD > $OUT/sq_$L.log 2>&1
  timeout -k 10 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $OUT/sq2_$L -- $CMD > $OUT/sq2_$L.log 2>&1
  python3 - <<PY
import csv, glob, collections, json
L = "$L"
out = open("$OUT/summary_%s.txt" % L, "w")
def P(*a):
    print(*a); print(*a, file=out)
P("== command: $CMD   (rocprofv3, ROCm 7.2, MI355X; one pass per counter group)")
for p in glob.glob("$OUT/stats_%s/*/*kernel_stats.csv" % L):
    P("== kernel stats (rocprofv3 --kernel-trace --stats)"); P(open(p).read())
tot = {}
for name in ("fetch", "write", "sq", "sq2"):
    for p in glob.glob("$OUT/%s_%s/*/*counter_collection.csv" % (name, L)):
        agg = collections.defaultdict(float); n = collections.defaultdict(set)
        for r in csv.DictReader(open(p)):
            if "transport" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
        for k, v in agg.items():
            tot[k] = (v, len(n[k]))
            P(f"== pmc {k}: sum over {len(n[k])} K1 launches = {v:.6g}; per launch = {v/len(n[k]):.6g}")
g = lambda k: tot.get(k, (0.0, 1))[0]
if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
    # SQ_* count quad-cycles summed over all SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (guide: rocprofv3 PMC slots)
    P(f"== derived: VALUBusy = 100 * SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8) = {100 * g('SQ_ACTIVE_INST_VALU') * 4 / 1024 / (g('GRBM_GUI_ACTIVE') / 8):.1f} %")
    P(f"== derived: lanes enabled per VALU instruction = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = {100 * g('SQ_THREAD_CYCLES_VALU') / (64 * g('SQ_ACTIVE_INST_VALU')):.1f} %")
    P(f"== derived: wave time split (quad-cycles): WAVE_CYCLES {g('SQ_WAVE_CYCLES'):.4g} = ACTIVE_INST_ANY {g('SQ_ACTIVE_INST_ANY'):.4g} + WAIT_INST_ANY {g('SQ_WAIT_INST_ANY'):.4g} + WAIT_ANY {g('SQ_WAIT_ANY'):.4g}")
if g("FETCH_SIZE") and g("WRITE_SIZE"):
    nl = tot["FETCH_SIZE"][1]
    tr = {"kernel": "mcs_k_transport_plain", "fetch_kib_per_launch": g("FETCH_SIZE") / nl, "write_kib_per_launch": g("WRITE_SIZE") / tot["WRITE_SIZE"][1],
          "hbm_bytes_per_launch": 1024.0 * (g("FETCH_SIZE") / nl + g("WRITE_SIZE") / tot["WRITE_SIZE"][1]), "launches": nl,
          "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of: $CMD",
          "note": "KiB as reported; uncalibrated for this kernel's 8-B-per-lane loads and 8-B fp64 atomics (the guide's x2 for FETCH_SIZE holds for 16-B streams); WRITE_SIZE is dominated by the memory-side fp64 tally atomics"}
    json.dump(tr, open("$OUT/traffic_%s.json" % L, "w"), indent=1)
for name in ("stats", "sq"):
    for l in open("$OUT/%s_%s.log" % (name, L)):
        if l.startswith("{"): P("== bench line under the %s pass:" % name, l.strip())
PY
}
run_passes 1e6 --steps 2 --warmup 1
if [ "$2" != "only1e6" ]; then run_passes 1e7 --particles 10000000 --steps 1 --warmup 0; fi
