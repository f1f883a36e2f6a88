"""Shared fixtures.  `-m "not gpu"`: oracle vs golden vectors, host logic, ABI
surface (no compute on a GPU).  `-m gpu`: parity of the HIP path with the oracle
through the C ABI (run on the MI355X box)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import _mcs_loader  # noqa: E402

mcs = _mcs_loader.load()
import orc  # noqa: E402  (test infrastructure: the CPU oracle)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    orc.build()


@pytest.fixture(scope="session")
def m():
    return mcs


def make_problem(N=512, **kw):
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, **kw)
    return mcs.inputs.build_problem(cfg)


def oracle_backend(prob, math="det", nthreads=1):
    be = orc.OracleBackend(mcs.capi, math=math, nthreads=nthreads)
    be.create(prob)
    return be


def hip_backend(prob):
    from mcs_amd import hip_backend as hb
    be = hb.HipBackend(0, debug_finals=True)
    be.create(prob)
    return be


def start_species(be, prob, i_iter=1, i_ion=1, shard=None):
    """begin_iteration + begin_species + init_pop for one species; returns Injection."""
    cfg = prob.cfg
    sp = cfg.species[i_ion - 1]
    be.begin_iteration(i_iter)
    inj = mcs.inputs.init_pop_host(prob, i_ion)
    pmax = mcs.inputs.get_pmax_cutoff(prob.Emax_keV, prob.Emax_per_aa_keV, prob.pmax, sp.aa)
    ewf = 1.0 / cfg.species[-1].density if cfg.species[-1].density else float("inf")
    be.begin_species(i_iter, i_ion, sp.aa, abs(sp.zz), pmax, sp.density, ewf)
    be.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
    lo, hi = shard if shard else (0, inj.n_pts_use)
    be.init_pop(inj, lo, hi - lo, inj.n_pts_use)
    return inj


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def assert_pop_equal(a, b, what=""):
    assert a.n == b.n, f"{what}: population sizes {a.n} != {b.n}"
    for f in a.fields():
        x, y = getattr(a, f), getattr(b, f)
        assert np.array_equal(bits(x), bits(y)), f"{what}: field {f} differs in {(x != y).sum()} entries"


def assert_tallies_close(L, Ta, Tb, rtol=1e-11):
    """fp64 tallies agree up to the order of the atomic adds: |a-b| <= rtol * max|b| per array."""
    for name in L.offsets:
        a, b = L.view(Ta, name), L.view(Tb, name)
        scale = float(np.max(np.abs(b)))
        if scale == 0.0:
            assert not np.any(a), f"tally {name}: expected all zero"
            continue
        err = float(np.max(np.abs(a - b))) / scale
        assert err <= rtol, f"tally {name}: max|diff|/max|ref| = {err:.3e} > {rtol}"


def fuzz_population(prob, N, seed, aa):
    """A caller's own population: every combination of the state bits (downstream, inj), positions all over the grid -- far
    upstream beyond the FEB, within 1e-7 r_g of the shock on either side, downstream of x_grid_stop --, momenta over six
    decades, the PRP on either side of the particle, ages around age_max, every time-cut index, both step sizes.  States the
    path itself never produces (e.g. downstream-flagged, not injected, at x < 0) are legal inputs of mcs_pop_upload."""
    rng = np.random.default_rng(seed)
    pop = mcs.capi.Population(N)
    mc = aa * mcs.constants.MP * mcs.constants.C
    pop.ptot_pf[:] = mc * 10 ** rng.uniform(-3.0, 3.5, N)
    pop.pb_pf[:] = pop.ptot_pf * rng.uniform(-1, 1, N)
    pop.weight[:] = rng.uniform(0.5, 1.5, N) / N
    xg = prob.x_grid_cm
    far = rng.random(N) < 0.5
    x_rg = np.where(far, rng.uniform(-150, 50, N), rng.choice([-1.0, 1.0], N) * 10 ** rng.uniform(-7, 0, N))
    x = np.maximum(x_rg * prob.rg0, xg[1] * 0.999)          # (downstream of x_grid_stop is a valid place: the last zone)
    pop.x_PT_cm[:] = x
    pop.grid[:] = np.searchsorted(xg, x, side="right") - 1
    pop.downstream[:] = rng.random(N) < 0.6
    pop.inj[:] = rng.random(N) < 0.4
    pop.xn_per[:] = np.where(rng.random(N) < 0.5, prob.params.xn_per_fine, prob.params.xn_per_coarse)
    pop.prp_x_cm[:] = np.where(rng.random(N) < 0.5, prob.params.x_grid_stop, np.abs(x) * 10 ** rng.uniform(-1, 1, N))
    pop.acctime_sec[:] = np.where(rng.random(N) < 0.3, 0.0, 10 ** rng.uniform(0, 9, N))
    old = rng.random(N) < 0.1
    pop.acctime_sec[old] = 10 ** rng.uniform(11, 12.5, int(old.sum()))          # around age_max
    pop.phi_rad[:] = rng.uniform(0, 2 * np.pi, N)
    nt = len(prob.tcuts)
    pop.tcut[:] = rng.integers(1, max(nt, 1) + 2, N) if nt else 1
    return pop


def fuzz_problem(kind, N):
    """The four configurations of the fuzzed-population runs: (problem, aa of its first species)."""
    me_mp = mcs.constants.ME / mcs.constants.MP
    kw, aa = {}, 1.0
    if kind == "general":
        kw = dict(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1)], energy_transfer_frac=0.1,
                  FEB_downstream=(30.0, 0.0), INJFR=[0.7, 1.0], maximum_energy=(0.0, 0.0, 1e3))
    elif kind == "electrons":
        kw = dict(species=[mcs.inputs.Species(me_mp, -1.0, 1e6, 1.0)], radiation_losses=True, B_mag_upstream=3.0, b_field_turbulence=1.0,
                  electron_energy_mfp_threshold=1e4, B_CMBz=1e-3)
        aa = me_mp
    prob = make_problem(N, **kw)
    if kind == "oblique":
        xg, up = prob.x_grid_cm, prob.x_grid_cm < 0
        ux = prob.ux.copy(); ux[up] = prob.ux[1] * (1 - 0.3 * np.exp(xg[up] / (50.0 * prob.rg0))); prob.ux = ux      # a precursor
        prob.theta = np.where(xg < 0, 0.35, 0.8); prob.uz = 0.05 * prob.ux; prob.utot = np.hypot(prob.ux, prob.uz)
        prob.gam_sf = 1 / np.sqrt(1 - (prob.utot / mcs.constants.C) ** 2)
    return prob, aa
