"""Per-pcut A/B of the pipelined pcut loop (mcs_run_pcuts_pipelined) against the per-pcut loop at BASELINE config[1]'s size: kernel time of
every pcut's (main) launch and the wall time of the species loop.  usage: gpu_pipe_ab.py [N] [long_draws ...]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from conftest import mcs, make_problem, hip_backend

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
IM = int(os.environ.get("MCS_LONG_IMULT_MAX", "8"))
Bs = [int(b) for b in sys.argv[2:]] or [8192]
prob = make_problem(N)
hb = hip_backend(prob)
rows = {}
for B in [0] + Bs:
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        r = mcs.driver.run(prob, hb, n_itrs=1, long_draws=B, long_imult_max=IM, species_tallies="light", fused_pcuts=False)
        wall = (time.perf_counter() - t0) * 1e3
        if best is None or wall < best[0]:
            best = (wall, [(s.i_pcut, s.n_pts_use, s.n_saved, s.kernel_ms) for s in r.stats], r.steps_helix + r.steps_retro)
    rows[B] = best
    print(f"long_draws {B:6d}: iteration wall {best[0]:8.2f} ms, sum of (main) kernel times {sum(x[3] for x in best[1]):8.2f} ms, steps {best[2]:.4e}", flush=True)
base = {x[0]: x for x in rows[0][1]}
for B in Bs:
    print(f"-- per pcut: per-pcut loop kernel ms | pipelined (long_draws {B}) main kernel ms")
    for x in rows[B][1]:
        b = base.get(x[0])
        print(f"pcut {x[0]:2d}  n_use {b[1] if b else -1:8d} {b[3] if b else float('nan'):7.2f} | n_use {x[1]:8d} {x[3]:7.2f}")
hb.destroy()
