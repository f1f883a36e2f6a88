"""The pass rate of the END of a launch: the K longest histories of pcut 5's first 4096 particles, run again ALONE (one wave, K live lanes,
queue exhausted from the start -- the state a launch is in while it waits for its last particles).  us per pass of the wave = kernel time / longest history.
usage: [MCS_TAIL_LOOP=n] python tools/gpu_tailpass.py [K ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch; torch.cuda.init()
import _mcs_loader; m = _mcs_loader.load()
from mcs_amd import hip_backend
Ks = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16]
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
pop_all = hb.get_population()
pop = pop_all.slice(0, 4096)
hb.set_population(pop)
hb.run_pcut(5, 0)
hel = hb.finals()['helix'][:4096].astype(np.int64)
order = np.argsort(-hel)
print(f"MCS_TAIL_LOOP={os.environ.get('MCS_TAIL_LOOP', '(default)')}; longest histories of the 4096: {hel[order[:8]].tolist()}")
for K in Ks:
    # particle k keeps its index (its random stream): the launch runs indices order[:K] through an index list
    sel = np.sort(order[:K])
    best = None
    for rep in range(3):
        hb.set_population(pop)
        g = torch.from_numpy(sel.astype(np.int64)).cuda()
        # (run_pcut_indexed runs local particle j with the stream of global index g[j]: select the K particles as the population)
        sub = pop.take(sel) if hasattr(pop, "take") else None
        if sub is None:
            raise SystemExit("population has no take()")
        hb.set_population(sub)
        hb.run_pcut_indexed(5, g)
        f = hb.finals()
        W = int(f['helix'][:K].max()); S = int(f['helix'][:K].astype(np.int64).sum())
        ms = hb.last_kernel_ms()
        best = ms if best is None else min(best, ms)
    ok = np.array_equal(np.sort(f['helix'][:K].astype(np.int64)), np.sort(hel[sel]))
    print(f"K={K:2d}: longest {W} passes, {S} particle-steps, kernel {best:.3f} ms -> {best*1e3/W:.3f} us per pass of the wave  (histories as in the full launch: {ok})")
hb.destroy()
