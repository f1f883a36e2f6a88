"""Could pcut k+1 start on the already-determined prefix of its population while pcut k's stragglers finish?

new_pcut (src/cuts.jl:34-98) numbers the children of the saved particles in the order of their parents: child
`o = pos * i_mult + j` of the parent that is the `pos`-th saved particle in index order.  The RNG key of a particle is
(iteration, species, pcut, o) (src/particle_loop.jl:35-40), so a child can start only when its `o` is known, i.e. when
  (a) every pcut-k particle with a lower index than its parent is resolved (saved or ended), and
  (b) i_mult = n_target // n_saved is known: floor(n_target / S) == floor(n_target / (S + U)) with S saved so far and
      U still running.
This script takes the per-particle step counts and saved flags of every pcut of one iteration (tools/gpu_histories.py)
and replays the launches on a model machine: `lanes` persistent lanes, particles claimed in index order by the lane
that frees up first, a particle of h steps holding its lane for h * tau.  It reports, per pcut: when the work counter
is exhausted, how many particles are unresolved then and where the lowest of them sits (the determined prefix), when
i_mult becomes known, and the makespan of the whole iteration (i) launch after launch, as the product runs it, and
(ii) with every child released at the moment (a) and (b) hold, on the same lanes -- the best a prefix pipeline can do.

usage: python tools/prefix_pipeline_sim.py histories.npz [lanes] > profiles/r03_prefix_pipeline_sim.txt
"""
import heapq
import sys

import numpy as np


def replay(steps, release, lanes_heap, tau):
    """Greedy in index order: particle i takes the lane that frees up first, starts at max(that, release[i]).
    lanes_heap: heap of lane-free times, updated in place.  Returns (start, finish) arrays."""
    n = len(steps)
    start = np.empty(n); fin = np.empty(n)
    pop, push = heapq.heappop, heapq.heappush
    dur = steps.astype(np.float64) * tau
    for i in range(n):
        t = pop(lanes_heap)
        r = release[i]
        if r > t:
            t = r
        start[i] = t
        f = t + dur[i]
        fin[i] = f
        push(lanes_heap, f)
    return start, fin


def main():
    d = np.load(sys.argv[1])
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1024 * 64
    n_pc = int(d["n_pcuts"])
    n_target = int(d["n_target"])
    kms = d["kernel_ms"]
    steps = [d[f"steps_{k}"].astype(np.int64) for k in range(n_pc)]
    saved = [np.unpackbits(d[f"saved_{k}"])[:len(steps[k])].astype(bool) for k in range(n_pc)]
    i_mult = d["i_mult"]

    # tau: one constant per particle-step, fitted so that the launch-after-launch replay reproduces the measured kernel time
    # (the replay is linear in tau: done once with tau = 1 and scaled)
    seq = []
    for k in range(n_pc):
        h = [0.0] * min(lanes, len(steps[k]))
        seq.append(replay(steps[k], np.zeros(len(steps[k])), h, 1.0))
        print(f"# replayed pcut {k+1}", file=sys.stderr, flush=True)
    tot = sum(f.max() for _, f in seq)
    tau = kms.sum() * 1e-3 / tot if np.isfinite(kms.sum()) and kms.sum() > 0 else 0.75e-6     # (no measured time: a nominal pass)
    print(f"# {sys.argv[1]}: {n_pc} pcuts, n_target {n_target}, {lanes} lanes; measured kernels {kms.sum():.1f} ms per iteration")
    print(f"# tau = {tau*1e6:.3f} us per particle-step (fit: the launch-after-launch replay takes the measured kernel time)")
    print("# pcut     n   saved i_mult | replay ms (measured) | exhausted at | unresolved then, lowest index (prefix %) | prefix 50 % at | i_mult known at | last ends")

    # (i) launch after launch   (ii) pipelined on shared lanes
    t_seq = 0.0
    heap_p = [0.0] * lanes
    rel = np.zeros(len(steps[0]))
    end_p = 0.0
    for k in range(n_pc):
        n = len(steps[k])
        st, f = seq[k][0] * tau, seq[k][1] * tau
        T = f.max()
        t_ex = st.max()                                   # the last claim
        unres = f > t_ex
        n_un = int(unres.sum())
        low = int(np.argmax(unres)) if n_un else n
        # running maximum of the finish times in index order = when the prefix up to i is resolved
        pre = np.maximum.accumulate(f)
        t_half = pre[n // 2 - 1] if n >= 2 else T
        # when is i_mult known: floor(nt / S(t)) == floor(nt / (S(t) + U(t)))
        order = np.argsort(f, kind="stable")
        S = np.cumsum(saved[k][order])                    # saved among the first j finished
        U = n - 1 - np.arange(n)                          # still running after the j-th finish
        Sf = np.maximum(S, 1)
        known = (S > 0) & ((n_target // Sf) == (n_target // np.maximum(Sf + U, 1))) | (U == 0)
        # the first j from which it stays known
        bad = np.nonzero(~known)[0]
        j_known = 0 if len(bad) == 0 else bad[-1] + 1
        t_known = f[order[min(j_known, n - 1)]]
        print(f"  {k+1:3d} {n:8d} {int(saved[k].sum()):7d} {int(i_mult[k]):6d} | {T*1e3:7.2f} ({kms[k]:6.2f}) | {t_ex*1e3:7.2f} ms | "
              f"{n_un:6d}  {low:8d} ({100.0*low/n:5.2f} %) | {t_half*1e3:7.2f} ms | {t_known*1e3:7.2f} ms | {T*1e3:7.2f} ms")
        t_seq += T

        # pipelined: this pcut's particles are released at rel[] (absolute times), on the shared lanes
        st_p, f_p = replay(steps[k], rel, heap_p, tau)
        end_p = max(end_p, f_p.max())
        if k + 1 < n_pc and len(steps[k + 1]) > 0:
            pre_p = np.maximum.accumulate(f_p)            # absolute time at which the prefix up to i is resolved
            order_p = np.argsort(f_p, kind="stable")
            S = np.cumsum(saved[k][order_p]); U = n - 1 - np.arange(n); Sf = np.maximum(S, 1)
            known = (S > 0) & ((n_target // Sf) == (n_target // np.maximum(Sf + U, 1))) | (U == 0)
            bad = np.nonzero(~known)[0]
            j_known = 0 if len(bad) == 0 else bad[-1] + 1
            t_known_p = f_p[order_p[min(j_known, n - 1)]]
            parents = np.nonzero(saved[k])[0]             # index of the pos-th saved particle
            im = int(i_mult[k])
            n_next = len(steps[k + 1])
            assert n_next == len(parents) * im, (k, n_next, len(parents), im)
            rel = np.repeat(np.maximum(pre_p[parents], t_known_p), im)
    print(f"# makespan, launch after launch : {t_seq*1e3:8.2f} ms (measured kernels: {kms.sum():.2f} ms)")
    print(f"# makespan, prefix pipeline     : {end_p*1e3:8.2f} ms   -> attainable gain {1e3*(t_seq-end_p):.2f} ms per iteration "
          f"({100*(t_seq-end_p)/t_seq:.1f} %)")


if __name__ == "__main__":
    main()
