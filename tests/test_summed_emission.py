"""The arithmetic of `get_summed_emission` (src/get_summed_emission.jl:37-413; the last piece of the dead photon stack) and
the photon shells it sums over (src/initializers.jl:305-398, src/MonteCarloScattering.jl:392-401): host numpy on the arrays the three
emission routines return (consumers.photon_shells / doppler_to_ism / summed_emission; S1-S5 list where the Julia text cannot run as
written).  No reference vectors exist: the checks are the properties the formulas imply -- the identity for a plasma at rest, photon number
conservation and the mean Doppler factor gamma for isotropic emission, the gamma^3 of beaming and time dilation, the offsets of the three
spectra on the common grid, shells that tile the zones."""
import math
import types

import numpy as np
import pytest

from conftest import mcs, make_problem

cons = mcs.consumers


def test_photon_shells_tile_the_grid():
    prob = make_problem(16)
    P = prob.params
    for nu, nd in ((3, 4), (1, 1), (6, 2)):
        ends, zones = cons.photon_shells(prob, nu, nd)
        assert len(ends) == nu + nd + 1 and np.all(np.diff(ends) > 0)
        assert ends[nu] == 0.0 and abs(ends[0] / P.feb_upstream - 1) < 1e-12 and abs(ends[-1] / P.x_grid_stop - 1) < 1e-12
        # equal steps in log10|x / rg0| outside the innermost shells, which start at the shock and end where the others would start
        up = np.log10(-ends[:nu] / prob.rg0)
        assert np.allclose(np.diff(up), up[1] - up[0] if nu > 1 else 0) if nu > 2 else True
        assert abs(math.log10(ends[nu + 1] / prob.rg0) - (-1 + (math.log10(P.x_grid_stop / prob.rg0) + 1) / nd)) < 1e-12
        x = np.asarray(prob.x_grid_cm)
        found = zones[zones > 0]
        assert np.all(np.diff(found) > 0)
        for k, z in enumerate(zones):
            if z > 0:
                assert x[z] <= ends[k] < x[z + 1]
    with pytest.raises(ValueError):
        cons.photon_shells(prob, 0, 3)


def test_doppler_shift_properties():
    n, bpd = 120, 10
    E = 10.0 ** (np.arange(n) / bpd)                     # MeV
    rng = np.random.default_rng(3)
    flux = np.full((4, n), 1e-99)
    flux[0, 30:60] = 10.0 ** rng.uniform(-8, -4, 30)
    flux[1, 40] = 2.5e-6
    flux[2, 20:50] = 1e-7
    # a plasma at rest: every bin keeps its photons (number per bin = flux per d(log10 E) x the bin width)
    out = cons.doppler_to_ism(flux, E, np.ones(4), np.zeros(4), bpd)
    assert np.allclose(out[:3], np.where(flux[:3] > 1e-90, flux[:3] / bpd, 1e-99), rtol=1e-12)
    assert np.all(out[3] == 1e-99)
    # a moving plasma: photon number conserved up to gamma^3; the mean energy grows by gamma (isotropic emission: <1 - beta cos> = 1)
    for beta in (0.3, 0.9):
        gam = 1 / math.sqrt(1 - beta * beta)
        out = cons.doppler_to_ism(flux, E, np.full(4, gam), np.full(4, beta), bpd)
        mid = E * 10.0 ** (0.5 / bpd)
        for z in range(3):
            lit_in, lit_out = flux[z] > 1e-90, out[z] > 1e-95
            n_in = (flux[z][lit_in] / bpd).sum()
            n_out = out[z][lit_out].sum() / gam ** 3
            assert abs(n_out / n_in - 1) < 1e-9
            e_in = (flux[z][lit_in] / bpd * mid[lit_in]).sum() / n_in
            e_out = (out[z][lit_out] * mid[lit_out]).sum() / out[z][lit_out].sum()
            assert abs(e_out / e_in / gam - 1) < 0.12                    # (re-binned on a grid of 10 per decade: +- one bin = 26 %)
        # a line spreads over the range gamma (1 -+ beta) of Doppler factors, and no further
        lit = np.flatnonzero(out[1] > 1e-95)
        assert E[lit[0]] <= E[40] * 10 ** (0.5 / bpd) * gam * (1 - beta) * 1.3 and E[lit[-1] + 1] >= E[40] * gam * (1 + beta) * 0.75
        assert E[lit[-1]] <= E[40] * 10 ** (1.0 / bpd) * gam * (1 + beta)
    # photons shifted beyond the grid are dropped, not folded back (S3)
    top = np.full((1, n), 1e-99); top[0, n - 2] = 1.0
    out = cons.doppler_to_ism(top, E, np.array([5.0]), np.array([math.sqrt(1 - 1 / 25)]), bpd)
    assert out[0][out[0] > 1e-95].sum() / 125 < 1.0 / bpd


def _fake_prob(n_grid, gam=None, beta=None):
    P = types.SimpleNamespace(n_grid=n_grid)
    g = np.ones(n_grid + 2) if gam is None else np.concatenate([[1.0], gam, [1.0]])
    b = np.zeros(n_grid + 2) if beta is None else np.concatenate([[0.0], beta, [0.0]])
    return types.SimpleNamespace(params=P, gam_ef=g, beta_ef=b)


def test_summed_emission_layout_and_sums():
    ng = 12
    n_pion = int(math.log10(cons.PHOTON_E_MAX_MEV / cons.PHOTON_PION_E_MIN_MEV) * 10)
    n_syn = int(math.log10(cons.PHOTON_SYNCH_E_MAX_MEV / cons.PHOTON_E_MIN_MEV) * 10)
    n_ic = int(math.log10(cons.PHOTON_E_MAX_MEV / cons.PHOTON_IC_E_MIN_MEV) * 10)
    Ep = cons.PHOTON_PION_E_MIN_MEV * 10.0 ** (np.arange(n_pion) / 10)
    Es = cons.PHOTON_E_MIN_MEV * 10.0 ** (np.arange(n_syn) / 10)
    Ei = cons.PHOTON_IC_E_MIN_MEV * 10.0 ** (np.arange(n_ic) / 10)
    fp = np.full((ng, n_pion), 1e-99); fs = np.full((ng, n_syn), 1e-99); fi = np.full((ng, n_ic), 1e-99)
    fp[2, 20] = 3e-7; fp[7, 20] = 1e-7; fp[7, 50] = 5e-9           # zones 3 and 8 (1-based)
    fs[4, 100] = 2e-5
    fi[9, 30] = 4e-8; fi[9, 60] = 4e-10
    pion = cons.PhotonPion(Ep, None, None, fp, None, 1)
    syn = cons.PhotonSynch(Es, None, None, fs)
    ic = cons.PhotonIC(Ei, None, None, fi, None)
    ends = np.array([1, 4, 9, 13])                                  # three shells: zones 1-3, 4-8, 9-12
    prob = _fake_prob(ng)                                           # a plasma at rest everywhere: the Doppler step is the identity
    se = cons.summed_emission(prob, ends, pion=pion, synch=syn, ic=ic)
    assert len(se.log_energy_MeV) == 250 and abs(se.log_energy_MeV[0] + 13) < 1e-12 and abs(se.log_energy_MeV[-1] - 11.9) < 1e-9
    lp, pp = se.per_process["pion"]
    assert pp.shape == (3, n_pion - 1) and np.allclose(lp, np.log10(Ep[:-1]))
    assert abs(10 ** pp[0, 20] / 3e-7 - 1) < 1e-12 and abs(10 ** pp[1, 20] / 1e-7 - 1) < 1e-12 and abs(10 ** pp[1, 50] / 5e-9 - 1) < 1e-12
    assert (pp > -99).sum() == 3
    ls, ps = se.per_process["synch"]
    assert abs(10 ** ps[1, 100] / 2e-5 - 1) < 1e-12 and (ps > -99).sum() == 1
    li, pi_ = se.per_process["ic"]
    assert abs(10 ** pi_[2, 30] / 4e-8 - 1) < 1e-12 and (pi_ > -99).sum() == 2
    # on the common grid every line sits at its own energy
    def at(logE):
        return int(round((logE + 13) * 10))
    sh = se.per_shell
    assert abs(10 ** sh[0, at(math.log10(Ep[20]))] / 3e-7 - 1) < 1e-12
    assert abs(10 ** sh[1, at(math.log10(Es[100]))] / 2e-5 - 1) < 1e-12
    assert abs(10 ** sh[2, at(math.log10(Ei[60]))] / 4e-10 - 1) < 1e-12
    assert (sh > -99).sum() == 6
    # where two processes meet, they add; the total is the sum over the shells
    fi2 = fi.copy(); fi2[7, int(round((math.log10(Ep[20]) + 2) * 10))] = 2e-7       # an IC line at the pion line's energy, zone 8
    se2 = cons.summed_emission(prob, ends, pion=pion, ic=cons.PhotonIC(Ei, None, None, fi2, None))
    assert abs(10 ** se2.per_shell[1, at(math.log10(Ep[20]))] / (1e-7 + 2e-7) - 1) < 1e-12
    assert abs(10 ** se2.total[at(math.log10(Ep[20]))] / (3e-7 + 1e-7 + 2e-7) - 1) < 1e-12
    # the plasma-frame spectra are boosted, the inverse-Compton one (computed in the ISM frame) is not
    gam = np.full(ng, 2.0); beta = np.full(ng, math.sqrt(0.75))
    se3 = cons.summed_emission(_fake_prob(ng, gam, beta), ends, pion=pion, ic=ic)
    assert np.array_equal(se3.per_process["ic"][1], se.per_process["ic"][1])
    n3 = (10 ** se3.per_process["pion"][1][se3.per_process["pion"][1] > -99]).sum()
    assert abs(n3 / (8 * (3e-7 + 1e-7 + 5e-9)) - 1) < 1e-9                                   # gamma^3 x the photons of all zones
    only = cons.summed_emission(prob, ends, pion=pion)
    assert set(only.per_process) == {"pion"}


def test_summed_emission_from_a_transport_run():
    """end to end on the oracle backend: a proton run, dN/dp, the pion fold, shells from the run's grid, the summed spectrum"""
    from conftest import oracle_backend
    prob = make_problem(400)
    be = oracle_backend(prob, nthreads=8)
    res = mcs.driver.run(prob, be, n_itrs=1)
    be.write_tallies(res.tallies_f64, res.tallies_i64)
    fin = cons.ion_finalize(prob, be, 1)
    ph = cons.photon_pion(prob, be, fin, 1, jet_dist_kpc=1.0e3)
    ends, zones = cons.photon_shells(prob, 3, 3)
    assert np.all(zones > 0)
    se = cons.summed_emission(prob, zones, pion=ph)
    lit = se.total > -99
    assert lit.sum() > 30 and np.all(se.log_energy_MeV[lit] >= 0.0 - 1e-9)                   # pion-decay photons: above 1 MeV
    # photon number is conserved by the Doppler step up to gamma^3 per zone: the total lies between the extremes
    f = ph.photon_flux[:, :-1]
    n_in = np.where(f > 1e-90, f / 10, 0.0).sum(axis=1)
    z_lo, z_hi = int(zones[0]), int(zones[-1]) - 1
    g3 = np.asarray(prob.gam_ef)[1:prob.params.n_grid + 1] ** 3
    tot = (10.0 ** se.total[lit] / 10).sum()
    inside = slice(z_lo - 1, z_hi)
    assert 0.8 * (n_in[inside] * g3[inside]).sum() <= tot <= 1.0001 * (n_in[inside] * g3[inside]).sum()
    be.destroy()
