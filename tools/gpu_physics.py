"""Physics check at scale: the downstream dN/dp slope of a full iteration against the Keshet & Waxman (2005) index the
reference prints (src/io.jl:146-150), for several N.  usage: python tools/gpu_physics.py [N ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import mcs, make_problem, hip_backend
from test_physics import keshet_waxman_slope, dndp_slope
for N in [int(x) for x in sys.argv[1:]] or [1_000_000]:
    prob = make_problem(N=N)
    be = hip_backend(prob)
    t0 = time.perf_counter(); res = mcs.driver.run(prob, be, n_itrs=1); dt = time.perf_counter() - t0
    P = prob.params
    want = keshet_waxman_slope(P)
    got = [dndp_slope(prob, be.layout, res.tallies_f64, z) for z in (P.i_shock + 3, P.i_shock + 10, P.i_shock + 14)]
    print(f"N = {N}: downstream dN/dp slope in zones shock+3, +10, +14: " + ", ".join(f"{g:.4f}" for g in got) +
          f"; Keshet & Waxman index {want:.4f}; {res.steps_helix + res.steps_retro:.3e} steps in {dt:.2f} s", flush=True)
    be.destroy()
