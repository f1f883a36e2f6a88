"""Which integer tallies differ between the HIP path and the oracle on the crafted electron population (N particles)?
usage: python tools/gpu_crafted_diff.py N"""
import sys
sys.path.insert(0, "tests")
import numpy as np
from conftest import mcs, hip_backend, oracle_backend
from golden_common import make_golden
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spec = make_golden.CASES["electrons_crafted_n64"]
kw = dict(spec["cfg"]); kw["species"] = [mcs.inputs.Species(**sp) for sp in kw["species"]]
def build():
    return mcs.inputs.build_problem(mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, **kw))
def run(backend, prob):
    cfg = prob.cfg; sp = cfg.species[0]
    backend.begin_iteration(1)
    inj = mcs.inputs.init_pop_host(prob, 1)
    pmax = mcs.inputs.get_pmax_cutoff(prob.Emax_keV, prob.Emax_per_aa_keV, prob.pmax, sp.aa)
    backend.begin_species(1, 1, sp.aa, abs(sp.zz), pmax, sp.density, 1.0 / cfg.species[-1].density)
    backend.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
    backend.set_population(make_golden.crafted_population("electrons", prob, N))
    per = []
    for ip in range(1, 5):
        n = backend.pop_size()
        ns = backend.run_pcut(ip, 0)
        per.append(backend.read_tallies()[1].copy())
        if ns == 0: break
        backend.new_pcut(max(n // ns, 1))
    return per
prob = build(); hb = hip_backend(prob); a = run(hb, prob)
prob = build(); ob = oracle_backend(prob, nthreads=16); b = run(ob, prob)
ng = prob.n_grid
names = {ng + v: k for k, v in mcs.capi.IC.items()}
for ip, (x, y) in enumerate(zip(a, b), 1):
    d = np.nonzero(x != y)[0]
    print(f"pcut {ip}: {len(d)} integer tallies differ:", [(int(i), names.get(int(i), f"num_crossings[{int(i)}]" if i < ng else "?"), int(x[i]), int(y[i])) for i in d[:20]])
