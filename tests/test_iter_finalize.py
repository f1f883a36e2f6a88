"""BASELINE config[2]'s loop closure on the CPU: the host-side profile update
(montecarloscattering.jl_amd/iter_finalize.py: iter_finalize -> smooth_grid_par, src/iter_finalize.jl:1-110,
src/smoothers.jl:54-604) against its independent C++ twin (oracle/mcs_iter.cpp, Newton iterations where the
product uses the closed-form roots), and the multi-iteration driver loop with an evolving profile."""
import copy
import ctypes as ct
import math

import numpy as np
import pytest

from conftest import mcs, orc, make_problem, oracle_backend

itf = mcs.iter_finalize
dp = ct.POINTER(ct.c_double)


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(dp)


def _species_arrays(prob):
    sp = prob.cfg.species
    return (np.array([s.density for s in sp]), np.array([s.temperature for s in sp]), np.array([s.mass for s in sp]))


def _twin_smooth(lib, prob, st, sm, pxx, en, q_px, q_en, P_tot):
    """oracle/mcs_iter.cpp:orc_smooth_grid_par on copies of the tables -> dict of new tables"""
    P, cfg = prob.params, prob.cfg
    t = {k: np.ascontiguousarray(getattr(prob, k)).copy() for k in ("ux", "gam_sf", "utot", "beta_ef", "gam_ef", "btot")}
    n0_aa = float(sum(s.density * s.aa for s in cfg.species))
    ux_new = np.zeros(P.n_grid)
    rc = lib.orc_smooth_grid_par(P.n_grid, P.i_shock, _p(np.ascontiguousarray(prob.x_grid_rg)), _p(t["ux"]), _p(t["gam_sf"]), _p(t["utot"]),
                                 _p(t["beta_ef"]), _p(t["gam_ef"]), _p(t["btot"]), _p(np.ascontiguousarray(prob.theta)),
                                 _p(np.ascontiguousarray(st.Gamma_grid[:, 1])), _p(np.ascontiguousarray(pxx)), _p(np.ascontiguousarray(en)),
                                 _p(np.ascontiguousarray(P_tot)), q_px, q_en, st.F_px_upstream, st.F_energy_upstream, n0_aa,
                                 P.u0, P.beta0, P.gam0, P.u2, sm.SMMOE, sm.SMPFP, st.prof_weight_fac,
                                 sm.artificial_smoothing_start_rg, cfg.B_mag_upstream, cfg.b_field_turbulence, cfg.b_field_amplify, _p(ux_new))
    assert rc == 0
    return t


def _one_iteration(N=2500, **kw):
    prob = make_problem(N, **kw)
    be = oracle_backend(prob, nthreads=8)
    res = mcs.driver.run(prob, be, None, n_itrs=1, finalize=True)
    (_, fin, ion), = res.iter_finals
    return prob, be, res, fin, ion


def test_upstream_fluxes_q_esc_and_gamma_grid_match_twin():
    prob = make_problem(64, target_compression_ratio=3.2, b_field_turbulence=0.5)
    lib = orc.load("det", mcs.capi)
    n0, T0, m = _species_arrays(prob)
    P = prob.params
    out = np.zeros(3)
    lib.orc_upstream_fluxes(len(n0), _p(n0), _p(T0), _p(m), prob.cfg.B_mag_upstream, prob.cfg.theta_B0, P.u0, P.beta0, P.gam0, _p(out))
    assert np.allclose(out, itf.upstream_fluxes(prob), rtol=1e-14, atol=0)
    # F_px of a cold gamma0 = 5 proton flow ~ (gamma beta)^2 rho c^2
    assert abs(out[0] / ((P.gam0 * P.beta0) ** 2 * mcs.constants.MP * mcs.constants.C ** 2) - 1) < 1e-3
    st = itf.IterState.create(prob, itf.SmoothingConfig(), 3)
    q = np.zeros(2)
    for Gam in (4.0 / 3.0, 1.45, 5.0 / 3.0):
        lib.orc_q_esc_calcs(Gam, prob.r_comp, st.r_RH, len(n0), _p(n0), _p(T0), _p(m), P.u0, P.beta0, P.gam0, P.u2, prob.beta2, prob.gam2, _p(q))
        mine = itf.q_esc_calcs(Gam, prob.r_comp, st.r_RH, prob)
        assert np.allclose(q, mine, rtol=1e-13, atol=0) and q[0] != 0.0
    assert itf.q_esc_calcs(1.4, st.r_RH, st.r_RH, prob) == (0.0, 0.0)         # r_comp == r_RH: nothing escapes (q_esc_calcs.jl:17)
    # Γ grid: iteration 1 seeds column 1 with 5/3 upstream and Γ2_RH downstream, later iterations shift
    rng = np.random.default_rng(1)
    n = P.n_grid
    Pp, Pq, e = rng.uniform(1, 2, n), rng.uniform(1, 2, n), rng.uniform(3, 9, n)
    e[[0, 5]] = 1.0e-99
    G1, G2 = np.zeros((n, 2)), np.zeros((n, 2))
    for it in (1, 2):
        itf.set_Gamma_adiab_grid(G1, it, prob.x_grid_cm, st.Gamma2_RH, Pp, Pq, e)
        lib.orc_set_gamma_grid(_p(G2), it, n, _p(np.ascontiguousarray(prob.x_grid_cm)), st.Gamma2_RH, _p(Pp), _p(Pq), _p(e))
        assert np.array_equal(G1, G2)
    assert G1[0, 1] == 1.0e-99 and G1[1, 1] == 1 + (Pp[1] + Pq[1]) / e[1] and G1[0, 0] == 1.0e-99


@pytest.mark.parametrize("variant", ["momentum", "energy-mix", "artificial", "damped"])
def test_smooth_grid_par_matches_twin_on_real_tallies(variant):
    prob, be, res, fin, ion = _one_iteration()
    sm = {"momentum": itf.SmoothingConfig(), "energy-mix": itf.SmoothingConfig(SMMOE=0.4),
          "artificial": itf.SmoothingConfig(artificial_smoothing_start_rg=-3.0),
          "damped": itf.SmoothingConfig(old_profile_weight=2.5, increase_old_profile_weighting=True)}[variant]
    lib = be.lib
    L = be.layout
    st = itf.IterState.create(prob, sm, 3)
    itf.set_Gamma_adiab_grid(st.Gamma_grid, 1, prob.x_grid_cm, st.Gamma2_RH, ion.P_psd_par, ion.P_psd_perp, ion.energy_density_psd)
    pxx = np.round(L.view(res.tallies_f64, "pxx_flux"), 13); en = np.round(L.view(res.tallies_f64, "energy_flux"), 13)
    P_tot = ion.P_psd_par + ion.P_psd_perp
    i_iter = 2 if variant == "damped" else 1
    mine = copy.deepcopy(prob)
    st2 = copy.deepcopy(st)
    assert itf.smooth_grid_par(mine, st2, sm, i_iter, pxx, en, 0.0, 0.0, ion.P_psd_par, ion.P_psd_perp)
    twin = _twin_smooth(lib, prob, st2, sm, pxx, en, 0.0, 0.0, P_tot)        # (st2: the damped weight is updated first, smoothers.jl:95-98)
    n = prob.n_grid
    for k in ("ux", "gam_sf", "utot", "beta_ef", "gam_ef", "btot"):
        a, b = getattr(mine, k), twin[k]
        assert np.allclose(a, b, rtol=1e-11, atol=0), (k, float(np.max(np.abs(a / b - 1))))
        assert a[0] == getattr(prob, k)[0] and a[n + 1] == getattr(prob, k)[n + 1]       # entries 0 and n_grid+1 are not touched
    u = mine.ux[1:n + 1]
    assert np.all(np.diff(u) <= 1e-12 * prob.params.u0)                     # monotone: the flow only decelerates towards the shock
    assert np.allclose(u[prob.x_grid_rg[1:n + 1] >= 0], prob.params.u2, rtol=1e-14, atol=0)       # downstream stays at the R-H speed
    assert u.max() <= prob.params.u0 * (1 + 1e-12) and u.min() >= prob.params.u2 * (1 - 1e-12)
    assert not np.array_equal(mine.ux, prob.ux)                             # a precursor has formed
    if variant == "damped":
        assert st2.prof_weight_fac == 10.0                                   # max(10, 2.5 * 1.15)


def test_old_profile_weight_does_not_compound_over_iterations():
    """`prof_weight_fac` is passed by value through main_loops -> iter_finalize -> smooth_grid_par and never returned
    (src/main_loops.jl:367, src/iter_finalize.jl:65, src/smoothers.jl:63,95-98): the damping rebinds a local, so every
    iteration starts again from "old-profile-weight" -- 1.15 w0 in iterations 2..5, 1.5 w0 from 6 on, floored at 10."""
    prob, be, res, fin, ion = _one_iteration(N=600)
    sm = itf.SmoothingConfig(old_profile_weight=9.0, increase_old_profile_weighting=True)
    L = be.layout
    st = itf.IterState.create(prob, sm, 8)
    itf.set_Gamma_adiab_grid(st.Gamma_grid, 1, prob.x_grid_cm, st.Gamma2_RH, ion.P_psd_par, ion.P_psd_perp, ion.energy_density_psd)
    pxx = np.round(L.view(res.tallies_f64, "pxx_flux"), 13); en = np.round(L.view(res.tallies_f64, "energy_flux"), 13)
    used = {}
    for i_iter in (1, 3, 7, 3, 2):
        mine = copy.deepcopy(prob)
        assert itf.smooth_grid_par(mine, st, sm, i_iter, pxx, en, 0.0, 0.0, ion.P_psd_par, ion.P_psd_perp)
        used.setdefault(i_iter, []).append(st.prof_weight_fac)
    assert used[1] == [9.0] and used[2] == [10.35] and used[7] == [13.5]
    assert used[3] == [10.35, 10.35]          # the same after iteration 7 as before it: nothing is carried


def test_tcut_print_normalisation_over_iterations():
    """driver.run(tcut_print=True) replicates the in-place rewrite of src/io.jl:28-45 at the end of every iteration
    (src/main_loops.jl:383-389): spectra_coupled, which is never reset, is normalised to 1 per (tcut, ion) and floored; the
    next iteration adds raw weights on top.  Checked against the C++ twin applied to the raw per-iteration increments."""
    N, npc = 1200, 12
    lib = orc.load("det", mcs.capi)
    raw = []
    prob = make_problem(N); be = oracle_backend(prob, nthreads=8)
    L = be.layout
    mcs.driver.run(prob, be, None, n_itrs=2, max_pcuts=npc,
                   on_iteration_end=lambda it: raw.append(tuple(L.view(be.read_tallies()[0], k).copy() for k in ("weight_coupled", "spectra_coupled"))))
    (w1, s1), (w2, s2) = raw
    assert s1.sum() > 0 and (s2 - s1).sum() > 0                  # both iterations tallied coupled spectra
    nt, nm = len(prob.tcuts), prob.params.num_psd_mom_bins

    def twin(w, s):
        w, s = np.ascontiguousarray(w).copy(), np.ascontiguousarray(s).copy()
        assert lib.orc_tcut_print(_p(w.ravel()), _p(s.ravel()), w.shape[0], nt, nm) == 0
        return w, s
    e_w1, e_s1 = twin(w1, s1)
    e_w2, e_s2 = twin(w2, e_s1 + (s2 - s1))                      # weight_coupled is reset every iteration, spectra_coupled never
    prob = make_problem(N); be = oracle_backend(prob, nthreads=8)
    got = []
    res = mcs.driver.run(prob, be, None, n_itrs=2, max_pcuts=npc, tcut_print=True,
                         on_iteration_end=lambda it: got.append(tuple(L.view(be.read_tallies()[0], k).copy() for k in ("weight_coupled", "spectra_coupled"))))
    (g_w1, g_s1), (g_w2, g_s2) = got
    assert np.array_equal(g_w1, e_w1) and np.allclose(g_s1, e_s1, rtol=1e-13, atol=0)
    assert np.allclose(g_w2, e_w2, rtol=1e-13, atol=0) and np.allclose(g_s2, e_s2, rtol=1e-12, atol=0)
    assert np.array_equal(L.view(res.tallies_f64, "spectra_coupled"), g_s2)      # the result carries the rewritten arrays
    tot = g_s1[0, :nt].sum(axis=1)
    hit = s1[0, :nt].sum(axis=1) > 0
    assert hit.any() and np.allclose(tot[hit], 1.0, rtol=1e-12)                  # every spectrum that was hit sums to 1
    assert np.all(g_s1[0, :nt, :nm + 1] >= 1.0e-99) and np.all(g_w1[0, :nt] >= 1.0e-99)
    assert np.all(g_s1[0, nt:] == 0)                                             # time cuts beyond n_tcuts are not touched


def test_classical_branch_matches_twin():
    """beta0 < 0.02 takes the non-relativistic equations (smoothers.jl:460-571; S2).  build_problem cannot make such a
    problem (calc_rRH's low-beta branch is broken in the reference, quirk G2), so the shock speed of a built problem is
    replaced by hand: 1500 km/s, r = 4."""
    C = mcs.constants.C
    prob = make_problem(64)
    P = prob.params
    sm = itf.SmoothingConfig(SMMOE=0.3)
    st = itf.IterState.create(prob, sm, 2)
    u0 = 1.5e8
    P.u0, P.beta0, P.gam0 = u0, u0 / C, 1 / math.sqrt(1 - (u0 / C) ** 2)
    P.u2 = u0 / 4; prob.beta2 = P.u2 / C; prob.gam2 = 1 / math.sqrt(1 - prob.beta2 ** 2)
    up = prob.x_grid_cm < 0
    prob.ux = np.where(up, u0, P.u2); prob.utot = prob.ux.copy()
    prob.gam_sf = 1 / np.sqrt(1 - (prob.ux / C) ** 2)
    n = P.n_grid
    st.F_px_upstream, _, st.F_energy_upstream = itf.upstream_fluxes(prob)       # the classical forms (initializers.jl:565-571,603-610)
    st.Gamma_grid[:, 1] = 5.0 / 3.0
    rho0 = mcs.constants.MP
    # fluxes of a mildly modified flow: momentum flux slightly above the cold ram pressure towards the shock
    x = prob.x_grid_rg[1:n + 1]
    ramp = 1 + 0.2 * np.exp(-np.abs(x) / 30.0)
    pxx = rho0 * u0 * prob.ux[1:n + 1] * ramp
    en = 0.5 * rho0 * u0 ** 3 * np.ones(n)
    mine = copy.deepcopy(prob)
    assert itf.smooth_grid_par(mine, copy.deepcopy(st), sm, 1, pxx, en, 0.0, 0.0, np.zeros(n), np.zeros(n))
    twin = _twin_smooth(orc.load("det", mcs.capi), prob, st, sm, pxx, en, 0.0, 0.0, np.zeros(n))
    for k in ("ux", "gam_sf", "beta_ef", "gam_ef", "btot"):
        assert np.allclose(getattr(mine, k), twin[k], rtol=1e-10, atol=0), k
    assert not np.array_equal(mine.ux, prob.ux)


def test_multi_iteration_loop_with_evolving_profile():
    """Three iterations of config[2]'s loop on the oracle: every iteration's profile update is applied (the tables the
    backend transports through change), the precursor deepens monotonically towards the shock, the downstream state stays
    the R-H one, and stepping through the loop one iteration at a time (first_iter / iter_state) gives the same result."""
    N = 1500
    sm = itf.SmoothingConfig(smooth_shocks=True)
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=3)
    prob = mcs.inputs.build_problem(cfg)
    be = orc.OracleBackend(mcs.capi, "det", nthreads=1); be.create(prob)
    profiles = []
    res = mcs.driver.run(prob, be, None, n_itrs=3, smoothing=sm, on_iteration_end=lambda it: profiles.append(prob.ux.copy()))
    assert len(res.iter_finals) == 3 and all(f.profile_changed for _, f, _ in res.iter_finals)
    n, P = prob.n_grid, prob.params
    for u in profiles:
        assert np.all(np.diff(u[1:n + 1]) <= 1e-12 * P.u0) and u[n] == P.u2 and u[0] == P.u0
    assert not np.array_equal(profiles[0], profiles[1]) and not np.array_equal(profiles[1], profiles[2])
    assert 1.2 < res.iter_finals[-1][1].Gamma_downstream < 1.7
    # the same loop, one call per iteration
    prob2 = mcs.inputs.build_problem(cfg)
    be2 = orc.OracleBackend(mcs.capi, "det", nthreads=1); be2.create(prob2)
    state = None
    for it in (1, 2, 3):
        r = mcs.driver.run(prob2, be2, None, n_itrs=1, smoothing=sm, first_iter=it, iter_state=state)
        state = r.iter_state
        assert np.array_equal(prob2.ux, profiles[it - 1])
    assert np.array_equal(r.tallies_i64, res.tallies_i64)
    assert np.array_equal(r.tallies_f64, res.tallies_f64)
