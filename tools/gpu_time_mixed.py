"""Kernel time of one iteration of the three-species case (p + He + e-, radiative losses, ion -> electron energy transfer):
usage: python tools/gpu_time_mixed.py [N per species] [fp32]"""
import sys, time
sys.path.insert(0, "tests")
from conftest import mcs
from mcs_amd import hip_backend
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
FP32 = "fp32" in sys.argv[2:]
me_mp = mcs.constants.ME / mcs.constants.MP
cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=1,
                        species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1), mcs.inputs.Species(me_mp, -1.0, 1e6, 1.2)],
                        energy_transfer_frac=0.1, radiation_losses=True, state_fp32=FP32)
prob = mcs.inputs.build_problem(cfg)
be = hip_backend.HipBackend(0); be.create(prob)
res = mcs.driver.run(prob, be, None, n_itrs=1)
ng, IC = prob.n_grid, mcs.capi.IC
prev = 0
for (it, ion, Gf, Gi) in res.per_species:
    st = [s for s in res.stats if s.i_ion == ion]
    steps = int(Gi[ng + IC["STEPS_HELIX"]] + Gi[ng + IC["STEPS_RETRO"]]) - prev; prev += steps
    ms = sum(s.kernel_ms for s in st)
    print(f"species {ion}: {len(st)} pcuts, {steps} steps, kernels {ms:.1f} ms -> {steps / (ms * 1e-3):.3e} steps/s; slowest pcuts "
          + ", ".join(f"{s.i_pcut}:{s.kernel_ms:.1f}ms/n={s.n_pts_use}" for s in sorted(st, key=lambda s: -s.kernel_ms)[:4]))
kms = sum(s.kernel_ms for s in res.stats)
print(f"TOTAL steps={res.steps_helix + res.steps_retro} kernel_ms={kms:.1f} rate={(res.steps_helix + res.steps_retro) / (kms * 1e-3):.3e} steps/s")
