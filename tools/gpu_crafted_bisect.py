"""Bisect the crafted electron population for particles whose zone-crossing tallies differ between the HIP path and the oracle.
usage: python tools/gpu_crafted_bisect.py N"""
import sys
sys.path.insert(0, "tests")
import numpy as np
from conftest import mcs, hip_backend, oracle_backend
from golden_common import make_golden
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spec = make_golden.CASES["electrons_crafted_n64"]
kw = dict(spec["cfg"]); kw["species"] = [mcs.inputs.Species(**sp) for sp in kw["species"]]
prob = mcs.inputs.build_problem(mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, **kw))
pop = make_golden.crafted_population("electrons", prob, N)
hb = hip_backend(prob); ob = oracle_backend(prob, nthreads=16)
sp = prob.cfg.species[0]
inj = mcs.inputs.init_pop_host(prob, 1)
pmax = mcs.inputs.get_pmax_cutoff(prob.Emax_keV, prob.Emax_per_aa_keV, prob.pmax, sp.aa)
ng = prob.n_grid
def run(be, lo, hi):
    be.begin_iteration(1)
    be.begin_species(1, 1, sp.aa, abs(sp.zz), pmax, sp.density, 1.0)
    be.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
    be.set_population(pop.slice(lo, hi))
    be.run_pcut(1, lo)
    T, I = be.read_tallies()
    return T.copy(), I.copy(), be.finals()
bad = []
def rec(lo, hi):
    (Ta, Ia, fa), (Tb, Ib, fb) = run(hb, lo, hi), run(ob, lo, hi)
    if np.array_equal(Ia[:ng], Ib[:ng]): return
    if hi - lo == 1:
        d = np.nonzero(Ia[:ng] != Ib[:ng])[0]
        bad.append(lo)
        print(f"particle {lo}: ptot/mec={pop.ptot_pf[lo] / (mcs.constants.ME * mcs.constants.C):.4g} mu={pop.pb_pf[lo] / pop.ptot_pf[lo]:.3f} x/rg0={pop.x_PT_cm[lo] / prob.rg0:.4g} "
              f"grid={pop.grid[lo]} down={pop.downstream[lo]} inj={pop.inj[lo]} | reason {fa['reason'][0]} helix {fa['helix'][0]} retro {fa['retro'][0]} x_end/rg0 {fa['x'][0] / prob.rg0:.4g} | "
              f"crossings differ in zones {d.tolist()}: gpu {Ia[d].tolist()} oracle {Ib[d].tolist()}", flush=True)
        return
    if len(bad) >= 6: return
    mid = (lo + hi) // 2
    rec(lo, mid); rec(mid, hi)
rec(0, N)
print("bad particles:", bad)
