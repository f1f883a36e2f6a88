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
