"""Kernel time per species of a p + e- run (radiative losses, energy transfer): python tools/gpu_species.py N"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _mcs_loader; m = _mcs_loader.load()
from mcs_amd import hip_backend
N = int(sys.argv[1])
ME_MP = m.constants.ME / m.constants.MP
cfg = m.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, radiation_losses=True, energy_transfer_frac=0.1,
                      species=[m.inputs.Species(1.0, 1.0, 1e6, 1.0), m.inputs.Species(ME_MP, -1.0, 1e6, 1.0)])
prob = m.inputs.build_problem(cfg)
hb = hip_backend.HipBackend(0, debug_finals=True); hb.create(prob)
res = m.driver.run(prob, hb, n_itrs=1)
ng = prob.n_grid; IC = m.capi.IC
prev = 0
for (it, ion, Gf, Gi) in res.per_species:
    st = int(Gi[ng + IC["STEPS_HELIX"]] + Gi[ng + IC["STEPS_RETRO"]]) - prev; prev += st
    ms = sum(s.kernel_ms for s in res.stats if s.i_ion == ion)
    npc = sum(1 for s in res.stats if s.i_ion == ion)
    print(f"species {ion}: {st} steps in {ms:.1f} ms of kernels over {npc} pcuts -> {st/(ms*1e-3):.3e} steps/s")
