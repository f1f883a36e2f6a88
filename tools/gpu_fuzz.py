"""Random caller-provided populations (every combination of the state bits, positions all over the grid, momenta over six
decades, PRP on either side of the particle) through the HIP path and the oracle: which results differ?
usage: python tools/gpu_fuzz.py [N] [seed] [kind: protons|general|electrons|oblique]"""
import sys
sys.path.insert(0, "tests")
import numpy as np
from conftest import mcs, hip_backend, oracle_backend, bits, fuzz_population, fuzz_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
KIND = sys.argv[3] if len(sys.argv) > 3 else "protons"


prob, aa = fuzz_problem(KIND, N)
pop = fuzz_population(prob, N, SEED, aa)
sp = prob.cfg.species[0]
inj = mcs.inputs.init_pop_host(prob, 1)
pmax = mcs.inputs.get_pmax_cutoff(prob.Emax_keV, prob.Emax_per_aa_keV, prob.pmax, sp.aa)
ng = prob.n_grid
names = {ng + v: k for k, v in mcs.capi.IC.items()}


def run(be, n_pc=3):
    be.begin_iteration(1)
    be.begin_species(1, 1, sp.aa, abs(sp.zz), pmax, sp.density, 1.0)
    be.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
    be.set_population(pop)
    out = []
    for ip in range(2, 2 + n_pc):
        n = be.pop_size()
        ns = be.run_pcut(ip, 0)
        out.append((be.finals(), be.get_saved(), be.read_tallies()))
        if ns == 0: break
        be.new_pcut(max(n // ns, 1))
    return out


hb = hip_backend(prob); ob = oracle_backend(prob, nthreads=16)
A, B = run(hb), run(ob)
print(f"kind {KIND}, N {N}, seed {SEED}: kernel {hb.last_kernel()}")
L = hb.layout
for ip, ((fa, (sa, la), (Ta, Ia)), (fb, (sb, lb), (Tb, Ib))) in enumerate(zip(A, B), 2):
    if any(fa[k].shape != fb[k].shape for k in fa):
        print(f"pcut {ip}: shapes differ: " + ", ".join(f"{k} {fa[k].shape} {fa[k].dtype} / {fb[k].shape} {fb[k].dtype}" for k in fa)); break
    reasons = np.bincount(fa["reason"], minlength=5).tolist()
    def neq(k):
        return (fa[k].view(np.uint64) != fb[k].view(np.uint64)) if fa[k].dtype == np.float64 else (fa[k] != fb[k])
    bad = [k for k in fa if neq(k).any()]
    mask = np.any([neq(k) for k in fa], axis=0)
    nbad = int(mask.sum())
    di = np.nonzero(Ia != Ib)[0]
    rel = np.abs(Ta - Tb) / np.maximum(np.abs(Tb), 1e-300)
    sig = (np.abs(Tb) > 1e-90) | (np.abs(Ta) > 1e-90)
    worst = float(rel[sig].max()) if sig.any() else 0.0
    print(f"pcut {ip}: n={len(fa['reason'])} reasons {reasons}; finals differ in {bad} for {nbad} particles; l_save equal {np.array_equal(la, lb)}; "
          f"{len(di)} integer tallies differ {[(names.get(int(i), f'num_crossings[{int(i)}]' if i < ng else '?'), int(Ia[i]), int(Ib[i])) for i in di[:8]]}; "
          f"worst relative tally difference {worst:.3e}")
    if nbad:
        idx = np.nonzero(mask)[0][:5]
        for i in idx:
            print(f"   particle {i}: gpu reason {fa['reason'][i]} helix {fa['helix'][i]} retro {fa['retro'][i]} | oracle reason {fb['reason'][i]} helix {fb['helix'][i]} retro {fb['retro'][i]}"
                  + (f" | in: x/rg0 {pop.x_PT_cm[i] / prob.rg0:.3g} p/mc {pop.ptot_pf[i] / (aa * mcs.constants.MP * mcs.constants.C):.3g} down {pop.downstream[i]} inj {pop.inj[i]} tcut {pop.tcut[i]} prp/rg0 {pop.prp_x_cm[i] / prob.rg0:.3g}" if ip == 2 else ""))
