"""SURVEY.md 8(f-4), second half: get_dNdp_2D (src/particle_counter.jl:343-627, live code, called at src/ion_finalize.jl:50-59) and
the inverse-Compton fold that consumes it (src/inverse_compton.jl:36-383 through src/photon_calcs.jl:116-138 -- dead code there,
followed as specification; include/mcs_ic.h lists where it cannot run as written).

CPU: the oracle-side twins (oracle/mcs_consumers.cpp: orc_dndp_2d, orc_photon_ic) against direct numpy evaluations that share no
code with them, the invariants the formulas imply (cell weights conserved by the rebin; the CMB table against the closed forms of a
black body), and the host wrapper's units.  GPU: the device kernels (K6: mcs_dndp_2d, mcs_photon_ic through the C ABI) against the
twins on IDENTICAL tallies -- 1e-11 relative to the largest entry: the only difference allowed is the order of fp64 adds and one
ulp of log / pow."""
import math

import numpy as np
import pytest

from conftest import mcs, make_problem, oracle_backend

C, ME, MP, QCGS, KB = mcs.constants.C, mcs.constants.ME, mcs.constants.MP, mcs.constants.QCGS, mcs.constants.KB
MEV = 1.602176634e-6
H = 6.62607015e-27


def relerr(a, b):
    s = float(np.max(np.abs(b)))
    return float(np.max(np.abs(a - b))) / s if s > 0 else float(np.max(np.abs(a)))


def electron_run(N=600):
    """p + e- with radiative losses through the oracle: real psd / therm_sf / num_crossings of the LAST species (quirk Q2: the
    histograms are cleared per species), which is the one get_dNdp_2D returns an ISM-frame array for (m_max = 2, :538)."""
    prob = make_problem(N, species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(ME / MP, -1.0, 1e6, 1.0)],
                        radiation_losses=True, B_mag_upstream=3e-3, JETFR=(0.0, 20.0))
    be = oracle_backend(prob, nthreads=8)
    res = mcs.driver.run(prob, be, n_itrs=1)
    be.write_tallies(res.tallies_f64, res.tallies_i64)
    return prob, be, res


def proton_run(N=600):
    """protons only: psd (injected particles) AND therm_sf (thermal crossings) are both populated, and the one species is the last
    one (m_max = 2): get_dNdp_2D's arithmetic does not care which species it rebins (E0 = m c^2 of `in`)."""
    prob = make_problem(N)
    be = oracle_backend(prob, nthreads=8)
    res = mcs.driver.run(prob, be, n_itrs=1)
    be.write_tallies(res.tallies_f64, res.tallies_i64)
    return prob, be, res


@pytest.fixture(scope="module")
def prun():
    prob, be, res = proton_run()
    yield prob, be, res
    be.destroy()


@pytest.fixture(scope="module")
def erun():
    prob, be, res = electron_run()
    yield prob, be, res
    be.destroy()


def dndp_2d_numpy(prob, t, T, I, L, gam_x, beta_x):
    """get_dNdp_2D written from the Julia text with array operations (no loop over cells)."""
    P = prob.params
    nm, nt, ng = P.num_psd_mom_bins, P.num_psd_tht_bins, P.n_grid
    psd, ths = L.view(T, "psd"), L.view(T, "therm_sf")
    dp = np.diff(t.mom_edge_cgs)                                    # [nm+1]
    K, J = np.meshgrid(np.arange(nm + 1), np.arange(nt + 1))        # [j][k]
    pt, cs = t.pt_center[K], t.cos_center[J]
    px = pt * cs
    et = np.hypot(pt * C, t.rest_energy)
    pxX = gam_x * (px - beta_x * et / C)
    ptX = np.sqrt(pt ** 2 - px ** 2 + pxX ** 2)
    kb = np.where(ptX < P.psd_mom_min, 0, np.trunc(np.log10(ptX / P.psd_mom_min) * P.psd_bins_per_dec_mom).astype(int) + 1)
    kb = np.minimum(kb, nm)
    cc = -pxX / ptX
    th = np.arccos(np.clip(cc, -1, 1))
    with np.errstate(divide="ignore"):
        jl = np.where(th < P.psd_tht_min, 0, np.trunc(np.log10(np.maximum(th, 1e-300) / P.psd_tht_min) * P.psd_bins_per_dec_tht).astype(int) + 1)
    jb = np.minimum(np.where(cc < P.psd_cos_fine, nt - np.trunc((cc + 1) / P.psd_dcos).astype(int), jl), nt)
    out = np.full((ng, nt + 2, nm + 2), 1e-99)
    weights = np.zeros((ng, 2))
    for i in range(1, ng + 1):
        sf = np.full((nt + 2, nm + 2), 1e-99)
        nc = int(I[i - 1])
        if nc != 0 and t.therm_from_hist:
            sf = sf + ths[i - 1]
        w = psd[i - 1]
        sf = sf + np.where(w > 1e-66, w, 0.0)
        sf[:, :nm + 1] = np.where(sf[:, :nm + 1] > 1e-66, sf[:, :nm + 1] / dp[None, :], sf[:, :nm + 1])
        dens = float(sf[sf > 1e-66].sum())
        if nc == 0 and dens > 0:
            dens += t.n0
        norm = t.zone_pop[i - 1] / dens if dens > 0 else 0.0
        sf = np.where((sf > 1e-99) & (norm > 0), sf * norm, 1e-99)
        c = sf[:nt + 1, :nm + 1]
        sel = c > 1e-66
        cw = c * dp[None, :]
        np.add.at(out[i - 1], (jb[sel], kb[sel]), (cw / dp[kb])[sel])
        weights[i - 1] = (cw[sel].sum(), 0.0)
    return out, weights


@pytest.mark.parametrize("which", ["protons", "electrons"])
def test_dndp_2d_matches_independent_numpy(prun, erun, which):
    prob, be, res = prun if which == "protons" else erun
    i_ion = 1 if which == "protons" else 2
    P, L = prob.params, mcs.capi.Layout(prob.params)
    t = mcs.consumers.consumer_tables(prob, i_ion)
    T, I = res.tallies_f64, res.tallies_i64
    assert L.view(T, "therm_sf").max() > 1e-60
    if which == "protons":
        assert L.view(T, "psd").max() > 1e-60          # (the thermal electrons of the other case are never injected: quirk Q5)
    got = be.dndp_2d(t, P.gam0, P.beta0)
    want, wts = dndp_2d_numpy(prob, t, T, I, L, P.gam0, P.beta0)
    assert got.shape == want.shape and want.max() > 1e-60
    assert relerr(got, want) < 1e-12
    assert np.array_equal(got > 1e-90, want > 1e-90)
    # the rebin conserves the cell weights: sum of ef * dp over a zone == sum of the normalised shock-frame cells * dp
    dp = np.diff(t.mom_edge_cgs)
    back = ((got[:, :, :P.num_psd_mom_bins + 1] - 1e-99) * dp[None, None, :]).sum(axis=(1, 2))
    ok = wts[:, 0] > 0
    assert ok.any() and np.allclose(back[ok], wts[ok, 0], rtol=1e-10)
    # ... and a zone's content is the zone population times the share of the cells that are rebinned (<= 1)
    assert np.all(back[ok] <= t.zone_pop[ok] * (1 + 1e-10))
    # the identity frame leaves every cell where its centre bins to: nothing moves to another momentum bin
    same = be.dndp_2d(t, 1.0, 0.0)
    sf_only, _ = dndp_2d_numpy(prob, t, T, I, L, 1.0, 0.0)
    assert relerr(same, sf_only) < 1e-12
    # as written (the thermal list is inert in the reference: A9 / C5) the thermal crossings are absent
    t0 = mcs.consumers.consumer_tables(prob, i_ion, therm_from_hist=False)
    got0 = be.dndp_2d(t0, P.gam0, P.beta0)
    want0, _ = dndp_2d_numpy(prob, t0, T, I, L, P.gam0, P.beta0)
    assert relerr(got0, want0) < 1e-12 or want0.max() <= 1e-99


def test_photon_field_is_the_cmb():
    """photon_field! tabulates a black body between nu_peak / 30 and 20 nu_peak: the bins must add up to (nearly) all of its
    photons and energy -- n = 16 pi zeta(3) (kT / hc)^3 = 410.7 cm^-3 and u = a T^4 = 4.17e-13 erg cm^-3 at 2.725 K."""
    for z in (0.0, 1.5):
        a, n = mcs.consumers.photon_field_cmb(z)
        T = 2.725 * (1 + z)
        n_bb = 16 * math.pi * 1.2020569031595942 * (KB * T / (H * C)) ** 3
        u_bb = 8 * math.pi ** 5 * KB ** 4 / (15 * H ** 3 * C ** 3) * T ** 4
        assert len(a) == 60 and np.all(np.diff(a) > 0)
        assert 0.97 < n.sum() / n_bb < 1.0, n.sum() / n_bb
        assert 0.99 < (a * ME * C * C * n).sum() / u_bb < 1.001
        assert abs(a[0] * ME * C * C / (H * 5.879e10 * T / 30 * 10 ** (math.log10(600) / 120)) - 1) < 1e-12     # first bin centre


def _synthetic_d2n(prob, seed=0):
    P = prob.params
    ng, NT, NM = P.n_grid, P.num_psd_tht_bins + 2, P.num_psd_mom_bins + 2
    t = mcs.consumers.consumer_tables(prob, 1)
    pe = t.mom_edge_cgs
    pc = np.sqrt(pe[:-1] * pe[1:])
    rng = np.random.default_rng(seed)
    d = np.full((ng, NT, NM), 1.0e-99)
    for zz in rng.choice(ng, size=6, replace=False):
        s = rng.uniform(1.8, 2.6)
        spec = 10 ** rng.uniform(55, 65) * (pc / pe[40]) ** (-s)
        ang = rng.uniform(0.0, 1.0, NT - 1) * (rng.uniform(0, 1, NT - 1) > 0.3)
        d[zz, :NT - 1, :NM - 1] = np.maximum(ang[:, None] * spec[None, :] * rng.uniform(0.5, 1.5, (NT - 1, NM - 1)), 1e-99)
        d[zz, :, :15] = 1.0e-99
    return t, pe, d


def ic_direct(pe, d, mc, j_max, alpha_in, n_in, n_photon, emin_mev, bpd, beam_area):
    """Jones (1968) eq. (9) over (electron bin, incoming bin, outgoing bin) by broadcasting -- no shared code with mcs_ic.h."""
    r0 = QCGS ** 2 / (ME * C * C)
    ao = 10 ** (math.log10(emin_mev * MEV / (ME * C * C)) + np.arange(n_photon) / bpd)
    dp = np.diff(pe)
    out = np.full((d.shape[0], n_photon), 1e-99)
    p1 = np.sqrt(pe[:-1] * pe[1:]) / mc
    g = np.where(p1 < 0.005, 1.0, np.hypot(p1, 1.0))
    for z in range(d.shape[0]):
        sl = np.where(d[z, :j_max + 1, :-1] <= 1e-99, 1e-99, d[z, :j_max + 1, :-1] * dp[None, :])      # [j][i]
        use = sl.max(axis=0) > 1e-99
        if not use.any():
            continue
        xn = sl.sum(axis=0)
        G, A1, AO = g[use][:, None, None], alpha_in[None, :, None], ao[None, None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            q = AO / (4 * A1 * G ** 2 * (1 - AO / G))
            val = (n_in[None, :, None] * 2 * math.pi * r0 ** 2 * C / (A1 * G ** 2)) * xn[use][:, None, None] * (
                2 * q * np.log(q) + (1 + 2 * q) * (1 - q) + 8 * (A1 * G * q) ** 2 * (1 - q) / (1 + 4 * A1 * G * q))
        val = np.where((AO < G) & (val > 1e-60), val, 0.0)
        d2 = 1e-99 + val.sum(axis=(0, 1))
        e = ao * ME * C * C
        em = d2 / beam_area / (ME * C * C) * e ** 2
        out[z] = np.where(em <= 1e-55, 1e-99, em)
    return ao * ME * C * C, out


def test_ic_fold_against_direct_evaluation():
    prob = make_problem(64, species=[mcs.inputs.Species(ME / MP, -1.0, 1e6, 1.0)])
    t, pe, d = _synthetic_d2n(prob)
    ob = oracle_backend(prob)
    ob._d2n = np.ascontiguousarray(d)
    alpha_in, n_in = mcs.consumers.photon_field_cmb(0.3)
    for j_max, n_photon, emin, bpd in ((prob.params.num_psd_tht_bins, 140, 1e-2, 10), (7, 33, 1e-5, 3)):
        area = 4 * math.pi * (3.0e24) ** 2 * 0.1
        E, got = ob.photon_ic(pe, t.mc, j_max, alpha_in, n_in, n_photon, emin, bpd, area)
        E_ref, want = ic_direct(pe, d, t.mc, j_max, alpha_in, n_in, n_photon, emin, bpd, area)
        assert np.allclose(E, E_ref, rtol=1e-13)
        assert want.max() > 1e-30 and (want > 1e-99).sum() > 100           # there is emission
        assert np.array_equal(got > 1e-99, want > 1e-99)
        assert np.allclose(got, want, rtol=1e-9, atol=0), float(np.max(np.abs(got / want - 1)))
        empty = np.flatnonzero((d <= 1e-99).all(axis=(1, 2)))
        assert np.all(got[empty] == 1e-99)
    # Thomson regime check on one electron bin: a photon of energy a1 is scattered to at most ~4 gamma^2 a1 (q = 1)
    one = np.full_like(d, 1e-99); i_el = 100
    one[0, 0, i_el] = 1e40
    g = math.hypot(math.sqrt(pe[i_el] * pe[i_el + 1]) / t.mc, 1)
    E, em = ob.__class__.photon_ic(_With(ob, one), pe, t.mc, 0, alpha_in[:1], n_in[:1], 200, 1e-12, 10, 1.0)
    top = E[em[0] > 1e-99].max() / (ME * C * C)
    assert 0.5 < top / (4 * g * g * alpha_in[0] / (1 + 4 * g * alpha_in[0])) <= 1.0 + 1e-9
    ob.destroy()


class _With:
    """an oracle backend whose stored d2N is replaced (photon_ic reads self._d2n, self.P, self.lib, self._chk)"""
    def __init__(self, ob, d2n):
        self.P, self.lib, self._chk, self._d2n = ob.P, ob.lib, ob._chk, np.ascontiguousarray(d2n)


def test_photon_ic_host_wrapper(erun):
    """consumers.photon_ic end to end on the oracle backend: get_dNdp_2D on real electron tallies, the CMB table, the cone, the
    fold, photon_IC's unit conversions (inverse_compton.jl:93-133)."""
    prob, be, res = erun
    ph = mcs.consumers.photon_ic(prob, be, 2, jet_dist_kpc=1.0e3, redshift=0.5)
    assert len(ph.energy_MeV) == 140 and abs(ph.energy_MeV[0] / 1e-2 - 1) < 1e-12 and abs(ph.energy_MeV[10] / 1e-1 - 1) < 1e-12
    assert ph.d2N_ef is not None and ph.d2N_ef.max() > 1e-60
    lit = ph.emis_erg > 1e-99
    assert np.all(ph.energy_flux_MeV[~lit] == 1e-99) and np.all(ph.photon_flux[~lit] == 1e-99)
    if lit.any():
        assert np.allclose(ph.energy_flux_MeV[lit], ph.emis_erg[lit] / MEV, rtol=1e-14)
        assert np.allclose(ph.photon_flux[lit], (ph.energy_flux_MeV / ph.energy_MeV[None, :])[lit], rtol=1e-14)
        assert np.allclose(ph.ic_photon_sum[lit], 1e-99 + (ph.emis_erg / (ph.energy_MeV[None, :] * MEV))[lit], rtol=1e-12)
    # the jet of the problem: 20 degrees -> f = (1 - cos 20deg) / 2; cone bin from the true cosines of the angle edges
    f = mcs.consumers.jet_sphere_fraction(prob.cfg)
    assert abs(f - (1 - math.cos(math.radians(20))) / 2) < 1e-15
    t = mcs.consumers.consumer_tables(prob, 2)
    j = mcs.consumers.ic_cone_last_bin(t.cos_edge, f, prob.params.num_psd_tht_bins)
    assert t.cos_edge[j] > 2 * f - 1 and (j == 0 or t.cos_edge[j - 1] <= 2 * f - 1)
    assert mcs.consumers.ic_cone_last_bin(t.cos_edge, 1.0, prob.params.num_psd_tht_bins) == prob.params.num_psd_tht_bins      # I2
    with pytest.raises(ValueError):
        mcs.consumers.photon_ic(prob, be, 1)               # protons: no inverse-Compton branch
    with pytest.raises(ValueError):
        mcs.consumers.photon_ic(prob, be, 2, jet_sph_frac=0.0)


@pytest.mark.gpu
def test_gpu_ic_matches_cpu_twin():
    """K6 on the device against the twins on identical tallies: get_dNdp_2D (real electron tallies of an oracle run, uploaded) and
    the inverse-Compton fold over it; then the fold alone over a synthetic array with bright zones."""
    from conftest import hip_backend
    # protons: injected particles and thermal crossings in one array
    prob, be, res = proton_run(2000)
    P = prob.params
    hb = hip_backend(prob)
    hb.begin_iteration(1)
    hb.write_tallies(res.tallies_f64, res.tallies_i64)
    for hist in (True, False):
        t = mcs.consumers.consumer_tables(prob, 1, therm_from_hist=hist)
        for gx, bx in ((P.gam0, P.beta0), (1.0, 0.0), (1.25, 0.6)):
            d_o = be.dndp_2d(t, gx, bx)
            d_g = hb.dndp_2d(t, gx, bx)
            assert d_o.max() > 1e-60
            assert relerr(d_g, d_o) < 1e-11
            assert np.array_equal(d_g > 1e-90, d_o > 1e-90)
    be.destroy(); hb.destroy()
    # electrons (the species the reference calls it for), and the fold over the result
    prob, be, res = electron_run(1500)
    P = prob.params
    hb = hip_backend(prob)
    hb.begin_iteration(1)
    hb.write_tallies(res.tallies_f64, res.tallies_i64)
    alpha_in, n_in = mcs.consumers.photon_field_cmb(0.0)
    for hist in (True, False):
        t = mcs.consumers.consumer_tables(prob, 2, therm_from_hist=hist)
        for gx, bx in ((P.gam0, P.beta0), (1.0, 0.0)):
            d_o = be.dndp_2d(t, gx, bx)
            d_g = hb.dndp_2d(t, gx, bx)
            assert d_o.max() > 1e-60 or not hist
            assert relerr(d_g, d_o) < 1e-11
            assert np.array_equal(d_g > 1e-90, d_o > 1e-90)
        area = 4 * math.pi * (3.0e24) ** 2 * 0.03
        for j_max, n_photon, emin, bpd in ((P.num_psd_tht_bins, 140, 1e-2, 10), (P.num_psd_tht_bins // 2, 40, 1e-6, 4)):
            Eo, o = be.photon_ic(t.mom_edge_cgs, t.mc, j_max, alpha_in, n_in, n_photon, emin, bpd, area)
            Eg, g = hb.photon_ic(t.mom_edge_cgs, t.mc, j_max, alpha_in, n_in, n_photon, emin, bpd, area)
            assert np.allclose(Eg, Eo, rtol=1e-14)
            assert np.array_equal(g > 1e-99, o > 1e-99)
            assert np.allclose(g, o, rtol=1e-11, atol=0), float(np.max(np.abs(g / o - 1)))
    # bright zones: a synthetic power-law psd for the electrons (the thermal electrons of this run are never injected, quirk Q5,
    # and radiate below the 1e-55 floor), through get_dNdp_2D and the fold on both sides
    L = mcs.capi.Layout(P)
    Tf, Ti = res.tallies_f64.copy(), res.tallies_i64.copy()
    psd = L.view(Tf, "psd")
    rng = np.random.default_rng(11)
    kk = np.arange(P.num_psd_mom_bins + 1)
    for zz in rng.choice(P.n_grid, size=8, replace=False):
        spec = 1e-6 * 10 ** (-0.02 * kk) * (kk > 30)
        ang = rng.uniform(0, 1, P.num_psd_tht_bins + 1) * (rng.uniform(0, 1, P.num_psd_tht_bins + 1) > 0.4)
        psd[zz, :P.num_psd_tht_bins + 1, :P.num_psd_mom_bins + 1] = np.maximum(ang[:, None] * spec[None, :], 1e-99)
    be.write_tallies(Tf, Ti); hb.write_tallies(Tf, Ti)
    t = mcs.consumers.consumer_tables(prob, 2)
    d_o, d_g = be.dndp_2d(t, P.gam0, P.beta0), hb.dndp_2d(t, P.gam0, P.beta0)
    assert relerr(d_g, d_o) < 1e-11
    lit = 0
    for j_max, n_photon, emin, bpd in ((P.num_psd_tht_bins, 140, 1e-2, 10), (40, 60, 1e-8, 5)):
        Eo, o = be.photon_ic(t.mom_edge_cgs, t.mc, j_max, alpha_in, n_in, n_photon, emin, bpd, area)
        Eg, g = hb.photon_ic(t.mom_edge_cgs, t.mc, j_max, alpha_in, n_in, n_photon, emin, bpd, area)
        assert np.array_equal(g > 1e-99, o > 1e-99)
        assert np.allclose(g, o, rtol=1e-11, atol=0), float(np.max(np.abs(g / o - 1)))
        lit += int((o > 1e-99).sum())
    assert lit > 200, lit
    be.write_tallies(res.tallies_f64, res.tallies_i64); hb.write_tallies(res.tallies_f64, res.tallies_i64)
    # the wrapper, device-resident (nothing but the spectra crosses PCIe)
    ph_g = mcs.consumers.photon_ic(prob, hb, 2, jet_dist_kpc=1.0e3)
    ph_o = mcs.consumers.photon_ic(prob, be, 2, jet_dist_kpc=1.0e3)
    assert ph_g.d2N_ef is None
    assert np.allclose(ph_g.emis_erg, ph_o.emis_erg, rtol=1e-11, atol=0)
    with pytest.raises(RuntimeError, match="bad arguments"):
        hb.photon_ic(t.mom_edge_cgs, t.mc, P.num_psd_tht_bins + 1, alpha_in, n_in, 10, 1e-2, 10, 1.0)
    be.destroy(); hb.destroy()
