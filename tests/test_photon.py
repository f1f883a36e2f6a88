"""the synchrotron fold of the reference's photon post-processing (src/synch_emission.jl through
src/photon_synch.jl and src/photon_calcs.jl -- dead code there, followed as specification).  The first synchrotron function
comes from a third-party package that is not in the reference tree (SynchrotronKernel.jl 0.2.2): restated as Chebyshev fits
(include/mcs_synch.h) and pinned here against scipy's Bessel functions; the fold against a direct numpy evaluation; the
device kernel (K5) against the CPU twin."""
import math

import numpy as np
import pytest
from scipy import integrate, special

from conftest import mcs, orc, make_problem, oracle_backend

C, ME, QCGS = mcs.constants.C, mcs.constants.ME, mcs.constants.QCGS
HBAR, MEV = 1.054571817e-27, 1.602176634e-6


def F_ref(x):
    """x int_x^inf K_{5/3} = x (2 K_{2/3}(x) - int_x^inf K_{1/3})."""
    k13 = lambda t: special.kv(1.0 / 3.0, t)
    if x < 1:
        v = integrate.quad(k13, x, 1.0, epsabs=0, epsrel=1e-13, limit=400)[0] + integrate.quad(k13, 1.0, np.inf, epsabs=0, epsrel=1e-13)[0]
    else:
        v = integrate.quad(k13, x, np.inf, epsabs=0, epsrel=1e-13, limit=400)[0]
    return x * (2 * special.kv(2.0 / 3.0, x) - v)


def test_first_synchrotron_function():
    lib = orc.load("det", mcs.capi)
    worst = 0.0
    for x in 10 ** np.linspace(-3, math.log10(29.99), 240):
        worst = max(worst, abs(lib.orc_synch_F(float(x)) / F_ref(float(x)) - 1))
    assert worst < 2e-10, worst
    # known values (Rybicki & Lightman fig. 6.6 / any table of F): the maximum 0.918 at x = 0.29, F(1) = 0.6514
    assert abs(lib.orc_synch_F(0.29) - 0.9179849599452145) < 1e-13 and abs(lib.orc_synch_F(1.0) - 0.651422815355364) < 1e-13
    # x -> 0: F = 4 pi / (sqrt(3) Gamma(1/3)) (x / 2)^(1/3) (1 - ...)
    F0 = 4 * math.pi / (math.sqrt(3) * math.gamma(1 / 3)) * 2 ** (-1 / 3)
    for x in (1e-15, 1e-12, 1e-9):
        assert abs(lib.orc_synch_F(x) / (F0 * x ** (1 / 3)) - 1) < 1e-5
    # continuity at the seam of the two fits
    assert abs(lib.orc_synch_F(4.0) / lib.orc_synch_F(math.nextafter(4.0, 5.0)) - 1) < 1e-12


def _synthetic(prob, seed=0):
    """Power-law electrons in a few zones, empty zones, a zone below the 1e-99 floor."""
    P = prob.params
    ng, NM = P.n_grid, P.num_psd_mom_bins + 2
    tabs = mcs.consumers.consumer_tables(prob, 1)
    pe = tabs.mom_edge_cgs
    rng = np.random.default_rng(seed)
    d = np.full((ng, NM), 1.0e-99)
    pc = np.sqrt(pe[:-1] * pe[1:])
    for z in rng.choice(ng, size=12, replace=False):
        s = rng.uniform(1.8, 2.6)
        d[z, :-1] = 10 ** rng.uniform(30, 40) * (pc / pe[40]) ** (-s) * rng.uniform(0.5, 1.5, NM - 1)
        d[z, :20] = 1.0e-99
    return tabs, pe, d


def _direct_fold(prob, pe, d, mc, n_photon, emin_mev, bpd):
    """synch_emission.jl:27-171 with scipy's Bessel functions, no shared code with include/mcs_synch.h."""
    ng, NM = d.shape
    E = 10 ** (math.log10(emin_mev * MEV) + np.arange(n_photon) / bpd)
    out = np.full((ng, n_photon), 1.0e-99)
    for z in range(ng):
        if not np.any(d[z] > 1e-99):
            continue
        B = prob.btot[z + 1]
        p_fac = math.sqrt(3) / (2 * math.pi) * (QCGS ** 3 * B / (ME * C * C))
        for i in range(NM - 1):
            xN = 1e-99 if d[z, i] <= 1e-99 else d[z, i] * (pe[i + 1] - pe[i])
            if xN <= 1e-60:
                continue
            p = math.sqrt(pe[i] * pe[i + 1])
            if p * C < 3 * MEV:
                continue
            ge = math.hypot(p / mc, 1)
            wc = 3 * ge * ge * QCGS * B / (2 * mc)
            for j in range(n_photon):
                wg = E[j] / HBAR
                x = wg / wc
                if x >= 30.0 or x < 1e-15:
                    continue
                add = xN * wg * p_fac * F_ref(x)
                if add > 1e-55:
                    out[z, j] += add
    return E, out


def test_fold_against_direct_evaluation():
    prob = make_problem(64, species=[mcs.inputs.Species(ME / mcs.constants.MP, -1.0, 1e6, 1.0)], b_field_turbulence=1.0, B_mag_upstream=3e-3)
    tabs, pe, d = _synthetic(prob)
    ob = oracle_backend(prob)
    n_photon, emin, bpd = 40, 1e-9, 4
    keep = np.flatnonzero((d > 1e-99).any(axis=1))[:2]
    d2 = np.full_like(d, 1e-99); d2[keep] = d[keep]             # (the direct evaluation is slow: two zones)
    E, got = ob.photon_synch(d2, pe, tabs.mc, n_photon, emin, bpd)
    E_ref, want = _direct_fold(prob, pe, d2, tabs.mc, n_photon, emin, bpd)
    assert np.allclose(E, E_ref, rtol=1e-14)
    assert want[keep].max() > 1e10                                # there is emission
    assert np.allclose(got, want, rtol=1e-9, atol=0)
    assert np.all(got[np.setdiff1d(np.arange(prob.n_grid), keep)] == 1e-99)      # empty zones stay at the floor


def test_photon_synch_host_wrapper_units():
    """consumers.photon_synch: energies, fluxes at Earth (photon_synch.jl:74-108) from the emitted dP/dlnE."""
    prob = make_problem(64, species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(ME / mcs.constants.MP, -1.0, 1e6, 1.0)],
                        b_field_turbulence=1.0, B_mag_upstream=3e-3)
    tabs, pe, d = _synthetic(prob, 3)
    ob = oracle_backend(prob)
    ng, NM = d.shape
    fin = mcs.consumers.IonFinal(np.stack([d, d, d]), tabs.zone_pop, None, None, None, None)
    ph = mcs.consumers.photon_synch(prob, ob, fin, 2, jet_dist_kpc=1.0e3, redshift=0.5)
    assert len(ph.energy_MeV) == 180 and abs(ph.energy_MeV[0] / 1e-13 - 1) < 1e-12 and abs(ph.energy_MeV[10] / 1e-12 - 1) < 1e-12
    dl = 1.0e3 * 1.5 * 3.0856775814913674e21
    z = int(np.argmax(ph.emis_erg_s.max(axis=1)))
    j = int(np.argmax(ph.emis_erg_s[z]))
    assert abs(ph.energy_flux_MeV[z, j] / (ph.emis_erg_s[z, j] / (4 * math.pi * dl ** 2) / MEV) - 1) < 1e-12
    assert abs(ph.photon_flux[z, j] / (ph.energy_flux_MeV[z, j] / ph.energy_MeV[j]) - 1) < 1e-12
    assert ph.photon_flux.min() == 1e-99
    with pytest.raises(ValueError):
        mcs.consumers.photon_synch(prob, ob, fin, 1)              # protons: no synchrotron branch


@pytest.mark.gpu
def test_gpu_fold_matches_cpu_twin():
    from conftest import hip_backend
    prob = make_problem(64, species=[mcs.inputs.Species(ME / mcs.constants.MP, -1.0, 1e6, 1.0)], b_field_turbulence=1.0, B_mag_upstream=3e-3)
    tabs, pe, d = _synthetic(prob, 5)
    hb, ob = hip_backend(prob), oracle_backend(prob)
    for n_photon, emin, bpd in ((180, 1e-13, 10), (37, 1e-7, 3)):
        Eg, g = hb.photon_synch(d, pe, tabs.mc, n_photon, emin, bpd)
        Eo, o = ob.photon_synch(d, pe, tabs.mc, n_photon, emin, bpd)
        assert np.allclose(Eg, Eo, rtol=1e-15) and o.max() > 1e10
        assert np.allclose(g, o, rtol=1e-11, atol=0), float(np.max(np.abs(g / o - 1)))
    hb.destroy()
