import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import _mcs_loader; m = _mcs_loader.load()
from mcs_amd import hip_backend
N = 200_000
for fp32 in (False, True):
    for it in (1, 2, 3):
        cfg = m.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, state_fp32=fp32, num_iterations=3)
        prob = m.inputs.build_problem(cfg)
        hb = hip_backend.HipBackend(0); hb.create(prob)
        r = m.driver.run(prob, hb, None, n_itrs=1, first_iter=it)
        print("fp32" if fp32 else "fp64", "iter", it, "steps", r.steps_helix + r.steps_retro, "pcuts", len(r.stats), flush=True)
        hb.destroy()
