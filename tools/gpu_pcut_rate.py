"""Per pcut of one BASELINE config[1] iteration: steps, kernel time, steps per particle and the roofline fraction of that launch alone.
usage: python tools/gpu_pcut_rate.py [N]"""
import sys, os
sys.path.insert(0, 'tests')
import numpy as np
from conftest import mcs, make_problem, hip_backend
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
prob = make_problem(N)
hb = hip_backend(prob)
ng, IC = prob.n_grid, mcs.capi.IC
cum = []
def hook(i_iter, i_ion, i_pcut):
    i = hb.read_counters()
    cum.append((i_pcut, int(i[ng + IC["STEPS_HELIX"]] + i[ng + IC["STEPS_RETRO"]])))
mcs.driver.run(prob, hb, n_itrs=1, before_pcut=hook)      # warm
cum.clear()
r = mcs.driver.run(prob, hb, n_itrs=1, first_iter=1, before_pcut=hook)
i = hb.read_counters(); cum.append((99, int(i[ng + IC["STEPS_HELIX"]] + i[ng + IC["STEPS_RETRO"]])))
tot_ms = 0
for (p, c0), (_, c1), s in zip(cum[:-1], cum[1:], r.stats):
    steps = c1 - c0
    frac = steps * 400 / (s.kernel_ms * 1e-3) / 78.6e12 if s.kernel_ms > 0 else 0
    print(f"pcut {p:2d} n_use {s.n_pts_use:8d} n_saved {s.n_saved:8d} steps {steps:.3e} kernel {s.kernel_ms:6.2f} ms  steps/particle {steps/max(s.n_pts_use,1):8.1f}  frac {frac:.3f}")
hb.destroy()
