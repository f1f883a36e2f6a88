"""Host-side input builder (montecarloscattering.jl_amd/inputs.py) against the values the
reference's initialisers produce for the stock-shaped configuration."""
import math

import numpy as np

from conftest import mcs, make_problem

C, MP = mcs.constants.C, mcs.constants.MP


def test_grid_and_profile():
    prob = make_problem(1000)
    P = prob.params
    assert P.n_grid == 99 and len(prob.x_grid_rg) == 101            # 1+27+35+5+16+16+1, initializers.jl:450-473
    assert np.all(np.diff(prob.x_grid_rg) > 0)                       # "intended" variant is monotone
    assert prob.x_grid_rg[P.i_shock] == 0.0 and P.i_shock == 65      # MonteCarloScattering.jl:478
    assert prob.x_grid_rg[0] == -1e30 and prob.x_grid_rg[-1] == 1e30
    assert prob.x_grid_rg[27] == -10.0 and prob.x_grid_rg[-2] == 10.0
    assert abs(prob.rg0 - 1.533e12) < 1e9                            # gamma0 beta0 mp c^2 / (e B0)
    assert abs(prob.r_comp - 4.0) < 1e-5                             # quirk G2: non-relativistic R-H branch
    up = prob.x_grid_cm < 0
    assert np.all(prob.ux[up] == P.u0) and np.all(prob.gam_sf[up] == P.gam0) and np.all(prob.gam_ef[up] == 1.0)
    assert np.allclose(prob.ux[~up], P.u0 / prob.r_comp)
    b2 = P.u2 / C
    assert np.allclose(prob.beta_ef[~up], (P.beta0 - b2) / (1 - P.beta0 * b2))
    assert np.all(prob.btot == 1e-5) and np.all(prob.theta == 0)
    assert prob.x_grid_cm[P.i_grid_feb] <= P.feb_upstream < prob.x_grid_cm[P.i_grid_feb + 1]
    assert prob.x_grid_rg[prob.i_fast_stop] == -1.0


def test_verbatim_grid_reproduces_quirk_G1():
    g = mcs.inputs.setup_grid(-1e7, 10.0, "verbatim")
    assert len(g) == 101
    assert g[27] < g[1] < 0                      # the "log-spaced" block runs AWAY from the shock
    assert g[83] == 1.0 and g[84] == 1.0         # duplicated edge at 1 rg0


def test_psd_bins():
    prob = make_problem(1000)
    P = prob.params
    assert P.num_psd_mom_bins == len(prob.psd_mom_bounds) - 2
    assert P.num_psd_tht_bins in (158, 159)      # 119 linear + 39/40 log
    assert abs(P.psd_cos_fine - (1 - 2 / 120)) < 1e-15
    assert abs(P.psd_dcos - (P.psd_cos_fine + 1) / 119) < 1e-15
    assert P.num_psd_mom_bins + 1 <= mcs.constants.PSD_MAX


def test_injection_distribution():
    n = 10_000
    ptot, w, n_use = mcs.inputs.set_inj_dist(True, n, 1, 1e6, MP, 1.0)
    assert abs(n_use - n) < 0.01 * n and len(ptot) == n_use and np.all(ptot > 0)   # no zero-momentum particle (G6)
    assert np.allclose(w, 1.0 / n_use) and abs(w.sum() - 1.0) < 1e-12
    assert np.all(np.diff(ptot) >= 0)
    kT = mcs.constants.KB * 1e6
    assert 0.5 < np.mean(ptot ** 2 / (2 * MP)) / (1.5 * kT) < 1.2       # <E> ~ 3/2 kT
    ptot2, w2, n2 = mcs.inputs.set_inj_dist(False, n, 1, 1e6, MP, 2.0)
    assert n2 == (n // 150) * 150 and abs(w2.sum() - 2.0) < 1e-9


def test_toml_config(tmp_path):
    p = tmp_path / "mc_in.toml"
    p.write_text('shock-speed = 3.0\nshock-speed-unit = "gamma"\nnum-iterations = 2\nAA_ION = [1.0, nan]\n'
                 'ZZ_ION = [1.0, -1.0]\nTZ_ION = [1e6, 1e6]\nDENZ_ION = [1.0, 1.0]\nno-scatter = false\n'
                 'momentum-cutoffs = [0.01, 1.0, 100.0]\nN_PTS_INJ = 300\nN_PTS_PCUT = 300\nN_PTS_PCUT_HI = 300\n')
    cfg = mcs.inputs.config_from_toml(str(p))
    assert cfg.shock_speed == 3.0 and cfg.num_iterations == 2 and len(cfg.species) == 2
    assert abs(cfg.species[1].aa - mcs.constants.ME / MP) < 1e-18 and cfg.species[1].zz == -1.0
    prob = mcs.inputs.build_problem(cfg)
    assert prob.params.n_ions == 2 and len(prob.pcuts) == 3 and prob.params.n_pts_max == 300


def test_shard_ranges():
    for n, w in ((10, 3), (1000003, 8), (5, 8), (0, 2)):
        r = [mcs.driver.shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1


def test_custom_epsB_profile():
    """`set_custom_εB!` (src/initializers.jl:868-951) and the field it implies (:833-845), restated in build_problem: epsilon_B against
    the distance from the shock in plasma skin depths, B = sqrt(|8 pi epsilon_B e(x)|)."""
    import math
    me_mp = mcs.constants.ME / mcs.constants.MP
    C, MP = mcs.constants.C, mcs.constants.MP
    sp = [mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(me_mp, -1.0, 1e6, 1.0)]
    plain = make_problem(16, species=sp)
    prob = make_problem(16, species=sp, use_custom_epsB=True)
    assert prob.params.use_custom_epsB == 1 and plain.params.use_custom_epsB == 0
    P, cfg = prob.params, prob.cfg
    F_px, _, F_en = mcs.iter_finalize.upstream_fluxes(prob)
    n0 = sum(s.density * s.mass for s in sp) / MP
    e0 = n0 * MP * C * C
    eps0 = cfg.B_mag_upstream ** 2 / (8 * math.pi * e0)
    rg2sd = P.beta0 / math.sqrt(2 * eps0 / P.gam0 * n0 / sp[-1].density)
    x_sd = np.asarray(prob.x_grid_rg) * rg2sd
    e_x = (F_en + P.gam0 * P.u0 * e0) / np.asarray(prob.ux) - F_px
    eps = np.asarray(prob.btot) ** 2 / (8 * math.pi * np.abs(e_x))
    far_up, near, down = x_sd < -50, np.abs(x_sd) < 50, x_sd >= 50
    assert far_up.any() and near.any() and down.any()
    assert np.allclose(eps[far_up], np.maximum(1.04e-5 / np.abs(x_sd[far_up]) ** 0.6, eps0), rtol=1e-12)
    assert np.allclose(eps[near], 1.0e-4, rtol=1e-12)
    assert np.allclose(eps[down], 5.0e-3 / x_sd[down], rtol=1e-12)          # E1: the decay never ends (comp_fac is 0 at the call)
    # far upstream epsilon_B falls to its floor B0^2 / (8 pi e0) and the field is the upstream field again (cold plasma: e(x) ~ e0)
    assert abs(prob.btot[0] / cfg.B_mag_upstream - 1) < 1e-3
    # everything but the field is the unmodified profile
    for name in ("ux", "gam_sf", "gam_ef", "beta_ef", "theta"):
        assert np.array_equal(getattr(prob, name), getattr(plain, name))
    assert not np.array_equal(prob.btot, plain.btot)
