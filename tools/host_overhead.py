"""Host-side overhead of one iteration: kernel time against the wall time of the run_pcut calls and of the whole driver
loop (new_pcut, fills, the 66 MB tally read-back): python tools/host_overhead.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _mcs_loader; m = _mcs_loader.load()
from mcs_amd import hip_backend
N=1000000
cfg = m.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=2)
prob = m.inputs.build_problem(cfg)
hb = hip_backend.HipBackend(0, debug_finals=True); hb.create(prob)
t0=time.perf_counter(); res = m.driver.run(prob, hb, n_itrs=2); t1=time.perf_counter()
st=[s for s in res.stats if s.i_iter==2]
print("iteration 2: pcuts", len(st), "kernel ms", sum(s.kernel_ms for s in st), "wall(run_pcut) ms", sum(s.wall_ms for s in st), "total 2 iters ms", (t1-t0)*1e3)
