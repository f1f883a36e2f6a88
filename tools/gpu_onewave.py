"""Latency probe: ONE wave (64 particles) through pcut 5; wave-steps = max helix."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _mcs_loader; m = _mcs_loader.load()
from mcs_amd import hip_backend
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
big = 20000
cfg = m.inputs.Config(N_PTS_INJ=big, N_PTS_PCUT=big, N_PTS_PCUT_HI=big)
prob = m.inputs.build_problem(cfg)
hb = hip_backend.HipBackend(0, debug_finals=True); hb.create(prob)
hb.begin_iteration(1)
inj = m.inputs.init_pop_host(prob, 1)
hb.begin_species(1, 1, 1.0, 1.0, prob.pmax, 1.0, 1.0)
hb.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
for ip in range(1, 5):
    ns = hb.run_pcut(ip, 0); hb.new_pcut(1)
pop = hb.get_population().slice(0, N)
hb.set_population(pop)
for rep in range(3):
    hb.set_population(pop)
    ns = hb.run_pcut(5, 0)
    f = hb.finals()
    W = int(f['helix'].max()); S = int(f['helix'].astype(np.int64).sum())
    ms = hb.last_kernel_ms()
    print(f"N={N} wave-steps(max helix)={W} particle-steps={S} kernel={ms:.3f} ms -> {ms*1e3/W:.3f} us per wave-step, lane util {S/(W*64*((N+63)//64)):.2f}")
