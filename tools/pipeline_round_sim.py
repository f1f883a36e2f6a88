"""Replay of the ROUND scheduler of the pipelined pcut loop (mcs_run_species_pipelined) on the model machine of
tools/prefix_pipeline_sim.py: what a practical policy keeps of the gain the ideal prefix pipeline shows.

A round launches, for every pcut that has work, ONE batch = [exported stragglers | newly released fresh particles] on its
share of the lanes (shares in proportion to the item counts).  A batch ends `budget` passes after its queue ran dry (its
unfinished particles are exported, to be resumed in the next round) or when everything finished; the round ends when all
its batches have, plus a fixed host cost.  Between rounds the host advances, per pcut, the determined prefix (first
unresolved index) and releases the children of the saved particles below it to the next pcut, provided i_mult is known:
the previous pcut complete (population size final) and floor(n_target / S) == floor(n_target / (S + U)).

usage: python tools/pipeline_round_sim.py histories.npz [budget_passes] [round_overhead_us] [lanes]
"""
import heapq
import sys

import numpy as np


def run_batch(rem, lanes, tau, budget, quantum=0.0):
    """rem: remaining steps of the batch's items in queue order.  Returns (new_rem, duration[, claimed mask]).
    quantum > 0: the batch ends at that time whatever its state (items not claimed by then stay in the queue)."""
    n = len(rem)
    if n == 0:
        return rem, 0.0
    L = max(1, min(lanes, n))
    heap = [0.0] * L
    start = np.empty(n)
    dur = rem * tau
    pop, push = heapq.heappop, heapq.heappush
    for i in range(n):
        t = pop(heap)
        start[i] = t
        push(heap, t + dur[i])
    fin = start + dur
    t_ex = start.max()
    T = min(fin.max(), t_ex + budget * tau)
    if quantum > 0:
        T = min(fin.max(), quantum)
    ran = np.clip(T - start, 0.0, dur)
    new_rem = np.where(fin <= T + 1e-15, 0.0, np.maximum(rem - np.floor(ran / tau), 1.0))
    new_rem = np.where(start >= T, rem, new_rem)
    return new_rem, T, start < T


def main():
    d = np.load(sys.argv[1])
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 1200.0
    ovh = float(sys.argv[3]) * 1e-6 if len(sys.argv) > 3 else 60e-6
    lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 1024 * 64
    tau = float(sys.argv[5]) * 1e-6 if len(sys.argv) > 5 else 0.843e-6
    quantum = float(sys.argv[6]) * 1e-6 if len(sys.argv) > 6 else 0.0
    n_pc = int(d["n_pcuts"]); n_target = int(d["n_target"])
    steps = [d[f"steps_{k}"].astype(np.float64) for k in range(n_pc)]
    saved = [np.unpackbits(d[f"saved_{k}"])[:len(steps[k])].astype(bool) for k in range(n_pc)]
    i_mult_true = d["i_mult"]

    rem = [s.copy() for s in steps]                 # remaining steps; 0 = resolved
    started = [np.zeros(len(s), bool) for s in steps]
    released = [0] * n_pc; released[0] = len(steps[0])
    size_final = [False] * n_pc; size_final[0] = True
    imult_known = [False] * n_pc
    t = 0.0
    rounds = 0
    busy_lane_time = 0.0
    while True:
        # the batches of this round
        batches = []
        for k in range(n_pc):
            idx_res = np.nonzero(started[k] & (rem[k] > 0))[0]
            idx_new = np.nonzero(~started[k][:released[k]])[0]
            if len(idx_res) + len(idx_new):
                batches.append((k, np.concatenate([idx_res, idx_new])))
        if not batches:
            break
        tot = sum(len(ix) for _, ix in batches)
        T_round = 0.0
        for k, ix in batches:
            share = max(256, int(lanes * len(ix) / tot)) if len(batches) > 1 else lanes
            new_rem, T, claimed = run_batch(rem[k][ix], share, tau, budget, quantum)
            busy_lane_time += float(np.sum(rem[k][ix] - new_rem)) * tau
            rem[k][ix] = new_rem
            started[k][ix[claimed]] = True
            T_round = max(T_round, T)
        t += T_round + ovh
        rounds += 1
        # host: prefixes, i_mult, releases
        for k in range(n_pc - 1):
            n_k = len(steps[k])
            unresolved = (rem[k] > 0) | ~started[k]
            unresolved[released[k]:] = True
            U = int(unresolved.sum())
            done_k = size_final[k] and released[k] == n_k and U == 0
            res_saved = int((saved[k] & ~unresolved).sum())
            if not imult_known[k] and size_final[k] and released[k] == n_k:
                if U == 0 or (res_saved > 0 and n_target // res_saved == n_target // (res_saved + U)):
                    imult_known[k] = True
            if imult_known[k]:
                low = int(np.argmax(unresolved)) if U else n_k
                n_par = int(saved[k][:low].sum())
                im = int(i_mult_true[k])
                released[k + 1] = max(released[k + 1], n_par * im)
            if done_k:
                size_final[k + 1] = True
    work = sum(float(s.sum()) for s in steps) * tau
    print(f"quantum {quantum*1e6:.0f} us, budget {budget:.0f} passes, round overhead {ovh*1e6:.0f} us, tau {tau*1e6:.3f} us: {rounds} rounds, makespan {t*1e3:.2f} ms "
          f"(work / lanes = {work/lanes*1e3:.2f} ms; lane utilisation {100*work/lanes/t:.1f} %)")


if __name__ == "__main__":
    main()
