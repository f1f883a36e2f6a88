#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/*.npz with the CPU oracle
(det math, single thread => deterministic tally order).

The reference ships NO golden vectors for this path and cannot be run here (Julia 1.12,
not installed), so these fixtures are produced by the build's own restatement
(oracle/mcs_oracle.cpp) -- "parity unpinned" with respect to the reference, pinned with
respect to the oracle: any later change of the oracle or of the HIP kernels that alters
a single bit of a particle history, or a tally beyond summation order, fails the tests.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _mcs_loader
mcs = _mcs_loader.load()
import orc

ME_MP = mcs.constants.ME / mcs.constants.MP

CASES = {
    # config[1]-shaped: protons, single unmodified shock, stock pcuts
    "protons_n128": dict(N=128, n_pcuts=9, cfg=dict()),
    # every optional branch: two species (p + e-), radiation losses, ion->electron energy
    # transfer, x_spec detectors, injection probability < 1, amplified downstream field
    "mixed_n96": dict(N=96, n_pcuts=8, cfg=dict(
        species=[dict(aa=1.0, zz=1.0, temperature=1e6, density=1.0), dict(aa=ME_MP, zz=-1.0, temperature=1e6, density=1.0)],
        energy_transfer_frac=0.1, radiation_losses=True, XSPEC_rg=[-0.5, 0.05, 2.0], INJFR=[0.7, 1.0],
        b_field_turbulence=1.0, shock_speed=3.0)),
    # no-scatter / no-DSA plumbing run (the stock mc_in.toml flags)
    "stock_flags_n64": dict(N=64, n_pcuts=4, cfg=dict(no_scatter=True, no_DSA=True)),
    # hand-placed relativistic electrons around the shock in a strong field: radiative
    # losses in the helix loop and in retro_time, the electron PRP shortening
    # (prob_return.jl:155-164), constant-mfp branch below p_e,crit, zero-energy exits
    "electrons_crafted_n64": dict(N=64, n_pcuts=3, crafted="electrons", cfg=dict(
        species=[dict(aa=ME_MP, zz=-1.0, temperature=1e6, density=1.0)], radiation_losses=True, B_mag_upstream=3.0,
        b_field_turbulence=1.0, momentum_cutoffs=(1.0, 30.0, 1e3, 1e5), maximum_energy=(0.0, 0.0, 1e6),
        electron_energy_mfp_threshold=1e4, B_CMBz=1e-3)),
}


def crafted_population(kind, prob, N):
    """Deterministic hand-made populations for branches a thermal injection never reaches."""
    rng = np.random.default_rng(12345)
    pop = mcs.capi.Population(N)
    if kind == "electrons":
        mec = mcs.constants.ME * mcs.constants.C
        pop.ptot_pf[:] = mec * 10 ** rng.uniform(0.5, 4.5, N)
        mu = rng.uniform(-1, 1, N)
        pop.pb_pf[:] = pop.ptot_pf * mu
        pop.weight[:] = 1.0 / N
        x_rg = np.where(np.arange(N) % 3 == 0, -10 ** rng.uniform(-6, -3, N), 10 ** rng.uniform(-6, 1.2, N))
        pop.x_PT_cm[:] = x_rg * prob.rg0
        pop.grid[:] = np.searchsorted(prob.x_grid_cm, pop.x_PT_cm, side="right") - 1
        pop.downstream[:] = (x_rg > 0) | (np.arange(N) % 2 == 0)
        pop.inj[:] = pop.downstream & (np.arange(N) % 4 == 0)
        pop.xn_per[:] = prob.params.xn_per_fine
        pop.prp_x_cm[:] = prob.params.x_grid_stop
        pop.acctime_sec[:] = 10 ** rng.uniform(0, 6, N)
        pop.phi_rad[:] = rng.uniform(0, 2 * np.pi, N)
        pop.tcut[:] = 1
    else:
        raise ValueError(kind)
    return pop


def build_case(name):
    spec = CASES[name]
    kw = dict(spec["cfg"])
    if "species" in kw:
        kw["species"] = [mcs.inputs.Species(**s) for s in kw["species"]]
    xs_rg = kw.pop("XSPEC_rg", None)
    N = spec["N"]
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, **kw)
    if xs_rg is not None:
        prob0 = mcs.inputs.build_problem(cfg)
        cfg.XSPEC = tuple(x * prob0.rg0 for x in xs_rg)
    return mcs.inputs.build_problem(cfg), spec


def run_case(backend, prob, spec, record):
    """Run all species of iteration 1 through the first n_pcuts pcuts, recording everything."""
    from conftest import start_species
    N = spec["N"]
    out = {}
    for i_ion in range(1, len(prob.cfg.species) + 1):
        if i_ion == 1:
            start_species(backend, prob, 1, 1)
            if spec.get("crafted"):
                backend.set_population(crafted_population(spec["crafted"], prob, N))
        else:   # later species keep the iteration's pools: no begin_iteration
            cfg = prob.cfg; sp = cfg.species[i_ion - 1]
            inj = mcs.inputs.init_pop_host(prob, i_ion)
            pmax = mcs.inputs.get_pmax_cutoff(prob.Emax_keV, prob.Emax_per_aa_keV, prob.pmax, sp.aa)
            backend.begin_species(1, i_ion, sp.aa, abs(sp.zz), pmax, sp.density, 1.0 / cfg.species[-1].density)
            backend.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
            backend.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
        pop = backend.get_population()
        for f in pop.fields():
            out[f"ion{i_ion}_init_{f}"] = getattr(pop, f).copy()
        for ip in range(1, spec["n_pcuts"] + 1):
            ns = backend.run_pcut(ip, 0)
            fin = backend.finals()
            saved, l_save = backend.get_saved()
            for k, v in fin.items():
                out[f"ion{i_ion}_pcut{ip}_final_{k}"] = v.copy()
            for f in saved.fields():
                out[f"ion{i_ion}_pcut{ip}_saved_{f}"] = getattr(saved, f).copy()
            out[f"ion{i_ion}_pcut{ip}_l_save"] = l_save.copy()
            if ns == 0:
                break
            backend.new_pcut(max(N // ns, 1))
        T, I = backend.read_tallies()
        out[f"ion{i_ion}_tallies_i64"] = I.copy()
        nz = np.nonzero((T != 0.0) & (T != 1e-99))[0]
        out[f"ion{i_ion}_tallies_idx"] = nz.astype(np.int64)
        out[f"ion{i_ion}_tallies_val"] = T[nz].copy()
        out[f"ion{i_ion}_tallies_floor_count"] = np.array([int((T == 1e-99).sum())])
    return out


def make_consumers(N=600):
    """consumers_n600.npz: dN/dp (3 frames, normalised) and thermo_calcs outputs of the oracle
    (oracle/mcs_consumers.cpp) on the tallies of a full single-threaded N=600 iteration."""
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N)
    prob = mcs.inputs.build_problem(cfg)
    be = orc.OracleBackend(mcs.capi, "det", nthreads=1)
    be.create(prob)
    res = mcs.driver.run(prob, be, n_itrs=1)
    be.write_tallies(res.tallies_f64, res.tallies_i64)
    fin = mcs.consumers.ion_finalize(prob, be, 1)
    idx = np.nonzero(fin.dNdp_cr > 1e-90)
    path = os.path.join(HERE, f"consumers_n{N}.npz")
    np.savez_compressed(path, dndp_idx=np.asarray(idx, dtype=np.int32), dndp_val=fin.dNdp_cr[idx], diag=fin.diag,
                        P_par=fin.P_psd_par, P_perp=fin.P_psd_perp, e_dens=fin.energy_density_psd, zone_pop=fin.zone_pop)
    print(f"consumers_n{N}: {len(idx[0])} non-empty dN/dp entries, {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    orc.build()
    make_consumers()
    if "--consumers-only" in sys.argv:
        return
    for name in CASES:
        prob, spec = build_case(name)
        be = orc.OracleBackend(mcs.capi, "det", nthreads=1)
        be.create(prob)
        out = run_case(be, prob, spec, True)
        out["meta"] = np.array(json.dumps(dict(case=name, N=spec["N"], n_pcuts=spec["n_pcuts"])))
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        steps = sum(int(v.astype(np.int64).sum()) for k, v in out.items() if k.endswith("final_helix"))
        print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB, helix steps {steps}")


if __name__ == "__main__":
    main()
