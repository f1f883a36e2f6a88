"""The common pass alone: protons placed at -75 r_g0 in the upstream zone [-83.8, -49.2] r_g0 (beyond the FEB zones, whose particles visit
the rare region in every pass) are carried ~6-9 r_g0 towards the shock in the 10^4 fine passes of the helix cap and never reach an
edge, the shock or a PRP, so every lane runs common passes from load to cap -- no rare region after the first
visit, no divergence, no tail (all histories have the same length).  What the kernel does then is the ceiling of the design.
usage: python tools/gpu_pass_ceiling.py [N [log10 p_lo  log10 p_hi [x / r_g0]]]"""
import sys
sys.path.insert(0, "tests")
import numpy as np
from conftest import mcs, hip_backend, make_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
prob = make_problem(N)
rng = np.random.default_rng(7)
pop = mcs.capi.Population(N)
mc = mcs.constants.MP * mcs.constants.C
LO, HI = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (-1.0, 1.0)
pop.ptot_pf[:] = mc * 10 ** rng.uniform(LO, HI, N)
pop.pb_pf[:] = pop.ptot_pf * rng.uniform(-1, 1, N)
pop.weight[:] = 1.0 / N
X_RG = float(sys.argv[4]) if len(sys.argv) > 4 else -75.0          # (e.g. -8e6: zone 1, one of the upstream FEB zones)
pop.x_PT_cm[:] = X_RG * prob.rg0
pop.grid[:] = np.searchsorted(prob.x_grid_cm, pop.x_PT_cm, side="right") - 1
pop.downstream[:] = 0; pop.inj[:] = 0
pop.xn_per[:] = prob.params.xn_per_fine
pop.prp_x_cm[:] = prob.params.x_grid_stop
pop.phi_rad[:] = rng.uniform(0, 2 * np.pi, N)
pop.tcut[:] = 1
hb = hip_backend(prob)
sp = prob.cfg.species[0]
inj = mcs.inputs.init_pop_host(prob, 1)
for rep in range(2):
    hb.begin_iteration(1)
    hb.begin_species(1, 1, sp.aa, abs(sp.zz), prob.pmax, sp.density, 1.0)
    hb.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
    hb.set_population(pop)
    ns = hb.run_pcut(45, 0)          # (the last pcut: nobody is saved)
    ms = hb.last_kernel_ms()
    T, I = hb.read_tallies()
    ng, IC = prob.n_grid, mcs.capi.IC
    steps = int(I[ng + IC["STEPS_HELIX"]]) - (0 if rep == 0 else steps0)
    steps0 = int(I[ng + IC["STEPS_HELIX"]])
    print(f"N = {N}, x = {X_RG} r_g0, log10(p / m_p c) in [{LO}, {HI}]: {steps} steps ({steps / N:.1f} per particle, HELIX_CAP exits so far {int(I[ng + IC['HELIX_CAP']])}), kernel {hb.last_kernel()} {ms:.2f} ms -> "
          f"{steps / (ms * 1e-3):.3e} steps/s = {steps * 400 / (ms * 1e-3) / 78.6e12:.3f} of the fp64 VALU peak by the 400-flop weight", flush=True)
