#!/bin/bash
# A/B sweep of the deferral threshold (MCS_DEFER_K) and the refill batch (MCS_REFILL_MIN): kernel time of a whole iteration.
# usage (GPU box): tools/sweep_defer_refill.sh <N> "<K list>" "<R list>"   -> one line per combination
N=${1:-1000000}; KS=${2:-"6 8 12"}; RS=${3:-"8 12 16"}
for K in $KS; do for R in $RS; do
  echo -n "N=$N defer_k=$K refill_min=$R : "
  MCS_DEFER_K=$K MCS_REFILL_MIN=$R timeout -k 10 120 python $GRAFT_REPO_ROOT/tools/gpu_run.py $N 45 | tail -1
done; done
