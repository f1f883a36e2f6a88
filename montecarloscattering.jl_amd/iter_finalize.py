"""Host side of the iteration loop that BASELINE config[2] names: `iter_finalize`
(src/iter_finalize.jl:1-110) and the profile update `smooth_grid_par` (src/smoothers.jl:54-348)
with `new_velocity_profile` (relativistic :351-458, classical :460-571) and `smooth_profile!`
(:583-604), plus the pieces they need: `upstream_fluxes` (src/initializers.jl:513-615),
`q_esc_calcs` (src/q_esc_calcs.jl), `set_Γ_adiab_grid!` (src/iter_finalize.jl:128-146) and
`populate_ε_target!` (src/iter_init.jl:1-15).

O(n_grid) arithmetic on the fluxes the transport kernel tallied and the pressures K4
reduced on the device; numpy, no particle data.  The result is a new set of grid tables,
handed to the device with `mcs_set_grid` / `mcs_set_cuts` before the next iteration.

Index convention: grid tables have n_grid + 2 entries, python index == Julia OffsetVector
index 0:n_grid+1; per-zone tallies (fluxes, pressures, Γ_grid) are Julia 1:n_grid -> python
0:n_grid-1.  Only entries 1..n_grid of the tables are updated (src/smoothers.jl:311-345).

Deviations from the reference as written (it cannot run there):
  S1  `Roots.find_zero(p, x0, Roots.Newton())` is given a scalar function without a derivative
      (src/smoothers.jl:412,423,517,525): Roots.Newton needs (f, f') and throws.  Both equations
      have closed-form roots -- the momentum equation is linear in γβ, the energy equation is
      γβ·√(1+γβ²) = K -- which is what Newton's iteration converges to; the closed forms are used
      (the CPU twin oracle/mcs_iter.cpp iterates Newton with the analytic derivative instead).
  S2  the classical branch reads an undefined `uₓ_guess` and `zero(n_grid)` (an Int) as an array
      (src/smoothers.jl:481,512); it is restated with u = β c as the unknown, as its comments say.
  S3  `mc_grid.dat` / plot output (src/smoothers.jl:100,233-270) is reporting: not produced.
  S4  quirk Q2 stands: the fluxes and pressures of the LAST species reach iter_finalize.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

from .constants import C, KB, MP

BETA_REL_FL = 0.02          # src/parameters.jl:30


@dataclasses.dataclass
class SmoothingConfig:
    """The mc_in.toml keys of the smoothing step (src/MonteCarloScattering.jl:121-146,186)."""
    smooth_shocks: bool = True                # "smooth-shocks"
    old_profile_weight: float = 1.0           # "old-profile-weight"
    increase_old_profile_weighting: bool = False
    SMMOE: float = 0.0                        # smooth_mom_energy_fac
    SMPFP: float = 0.0                        # smooth_pressure_flux_psd_fac (the reference refuses > 0)
    artificial_smoothing_start_rg: float = 0.0


@dataclasses.dataclass
class IterState:
    """Arrays that live across iterations (src/MonteCarloScattering.jl:267,440,589-593)."""
    Gamma_grid: np.ndarray            # [n_grid, 2]
    px_esc_flux_upstream: np.ndarray  # [n_itrs]
    energy_esc_flux_upstream: np.ndarray
    q_esc_cal_px: np.ndarray
    q_esc_cal_energy: np.ndarray
    Gamma_downstream: np.ndarray
    prof_weight_fac: float
    F_px_upstream: float
    F_energy_upstream: float
    Gamma2_RH: float
    r_RH: float

    @staticmethod
    def create(prob, sm: SmoothingConfig, n_itrs: int) -> "IterState":
        F_px, _, F_en = upstream_fluxes(prob)
        z = lambda: np.zeros(n_itrs)
        from .inputs import calc_rRH
        return IterState(np.zeros((prob.n_grid, 2)), z(), z(), z(), z(), z(), sm.old_profile_weight, F_px, F_en,
                         5.0 / 3.0, calc_rRH(prob.params.beta0, prob.cfg.species))
        # Γ₂_RH: calc_rRH takes its non-relativistic branch for every β₀ >= 0.02 (quirk G2) and pegs Γ₂ to 5/3
        # (src/initializers.jl:77-87,107)


def upstream_fluxes(prob):
    """src/initializers.jl:513-615 -> (F_px_upstream, F_pz_upstream, F_energy_upstream)."""
    P, cfg = prob.params, prob.cfg
    n0 = np.array([s.density for s in cfg.species]); T0 = np.array([s.temperature for s in cfg.species])
    m = np.array([s.mass for s in cfg.species])
    P0 = float(np.dot(n0, T0)) * KB
    rho0 = float(np.dot(n0, m))
    gam_sph = 5.0 / 3.0
    e0 = rho0 * C ** 2 + 1 / (gam_sph - 1) * P0
    B0 = cfg.B_mag_upstream
    Bx, Bz = B0 * math.cos(math.radians(cfg.theta_B0)), B0 * math.sin(math.radians(cfg.theta_B0))
    u0, b0, g0 = P.u0, P.beta0, P.gam0
    if b0 >= BETA_REL_FL:
        F_px = (g0 * b0) ** 2 * (e0 + P0) + P0 + g0 ** 2 * ((b0 * B0) ** 2 + Bz ** 2 - Bx ** 2) / (8 * math.pi)
        F_pz = -g0 * Bx * Bz / (4 * math.pi)
        F_en = C * (g0 ** 2 * b0 * (e0 + P0) + g0 ** 2 * b0 * Bz ** 2 / (4 * math.pi)) - g0 * u0 * rho0 * C ** 2
    else:
        xi = gam_sph / (gam_sph - 1)
        F_px = rho0 * u0 ** 2 * (1 + b0 ** 2) + P0 * (1 + xi * b0 ** 2) + Bz ** 2 / (8 * math.pi)
        F_pz = -Bx * Bz / (4 * math.pi)
        F_en = rho0 * u0 ** 3 * (1 + 1.25 * b0 ** 2) / 2 + P0 * u0 * xi * (1 + b0 ** 2) + u0 * Bz ** 2 / (4 * math.pi)
    return F_px, F_pz, F_en


def q_esc_calcs(Gamma, r_comp, r_RH, prob):
    """src/q_esc_calcs.jl:11-125 -> (first, second) in the reference's RETURN order, which the caller binds to
    (q_esc_cal_pₓ, q_esc_cal_energy) (src/iter_finalize.jl:37): both branches return (energy, pₓ), so the two
    names are swapped at the call site as written; replicated (zero for r_comp == r_RH either way)."""
    if r_comp == r_RH:
        return 0.0, 0.0
    P, cfg = prob.params, prob.cfg
    u0, b0, g0, u2, b2, g2 = P.u0, P.beta0, P.gam0, P.u2, prob.beta2, prob.gam2
    n0 = np.array([s.density for s in cfg.species]); T0 = np.array([s.temperature for s in cfg.species])
    m = np.array([s.mass for s in cfg.species])
    Gf = Gamma / (Gamma - 1)
    P0 = float(np.dot(n0, T0)) * KB
    rho0 = float(np.dot(n0, m))
    if b0 >= BETA_REL_FL:
        q_fac = C * math.sqrt((1 + b0) / 2)
        F_px = g0 ** 2 * b0 ** 2 * (rho0 * C ** 2 + 2.5 * P0) + P0
        F_en = g0 ** 2 * u0 * (rho0 * C ** 2 + 2.5 * P0)
        aux = g2 ** 2 * (q_fac * b2 ** 2 - u2)
        rho2 = rho0 * g0 * b0 / (g2 * b2)
        P2 = (q_fac * F_px - F_en - aux * rho2 * C ** 2) / (q_fac + Gf * aux)
        Q_px = F_px - (g2 * b2) ** 2 * (rho2 * C ** 2 + Gf * P2) - P2
        Q_en = Q_px * q_fac
        return Q_en / (F_en - g0 * u0 * rho0 * C ** 2), Q_px / F_px
    F_px = rho0 * u0 ** 2 + P0
    F_en = rho0 * u0 ** 3 / 2 + 2.5 * P0 * u0
    rho2 = rho0 * g0 * b0 / (g2 * b2)
    P2 = F_px - rho2 * u2 ** 2
    Q_en = F_en - rho0 * u0 * u2 ** 2 / 2 - P2 * u2 * Gf
    return Q_en / F_en, 0.0


def set_Gamma_adiab_grid(Gamma_grid, i_iter, x_grid_cm, Gamma2_RH, P_par, P_perp, e_dens):
    """src/iter_finalize.jl:128-146; Γ_grid rows are zones 1..n_grid."""
    n = Gamma_grid.shape[0]
    if i_iter == 1:
        up = x_grid_cm[1:n + 1] <= 0          # x_grid_cm[axes(Γ_grid, 1)] = entries 1..n_grid
        Gamma_grid[up, 0] = 5.0 / 3.0
        Gamma_grid[~up, 0] = Gamma2_RH
    else:
        Gamma_grid[:, 0] = Gamma_grid[:, 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        Gamma_grid[:, 1] = 1 + (P_par + P_perp) / e_dens
    Gamma_grid[e_dens == 1.0e-99, 1] = 1.0e-99
    return Gamma_grid


def smooth_profile(y):
    """`smooth_profile!` (src/smoothers.jl:583-604), y = zones 1..n_grid (python 0..n-1), in place."""
    n = len(y)
    for i in range(n - 1, 0, -1):              # Julia i = n_grid .. 2
        if y[i - 1] < y[i]:
            y[i - 1] = y[i]
    d = y.copy()
    d[1] = (2 * y[0] + y[1] + y[2]) / 4
    for i in range(2, n - 2):                  # Julia 3 .. n_grid-2
        d[i] = (y[i - 1] + y[i] + y[i + 1]) / 3
    d[n - 2] = (y[n - 3] + y[n - 2] + 2 * y[n - 1]) / 4
    y[1:n - 1] = d[1:n - 1]
    return y


def new_velocity_profile(prob, st: IterState, sm: SmoothingConfig, pxx_flux, energy_flux, q_px, q_en, P_tot_MC):
    """`new_velocity_profile` (src/smoothers.jl:351-571) -> uₓ_new for zones 1..n_grid."""
    P = prob.params
    n = P.n_grid
    u0, b0, g0, u2 = P.u0, P.beta0, P.gam0, P.u2
    n0 = float(sum(s.density * s.aa for s in prob.cfg.species))          # dot(n₀_ion, aa_ion), smoothers.jl:83
    relativistic = b0 >= BETA_REL_FL
    Qpx = q_px * pxx_flux[0] if relativistic else 0.0
    Qen = q_en * energy_flux[0]
    ux_px, ux_en = np.zeros(n), np.zeros(n)
    avg_px = avg_en = 0.0
    w = sm.SMPFP
    F_px, F_en = st.F_px_upstream, st.F_energy_upstream
    for i in range(1, n + 1):
        ux = prob.ux[i]; bux = ux / C; g = prob.gam_sf[i]
        g2_, gb = g * g, g * bux
        G_post = st.Gamma_grid[i - 1, 1]
        Xi = G_post / (G_post - 1)
        B = prob.btot[i]
        Bx, Bz = B * math.cos(prob.theta[i]), B * math.sin(prob.theta[i])
        pxx_EM = gb ** 2 / (8 * math.pi) * B ** 2 + g2_ / (8 * math.pi) * (Bz ** 2 - Bx ** 2)
        en_EM = g2_ / (4 * math.pi) * bux * Bz ** 2
        if relativistic:
            dens = g0 * b0 / (g * bux) * n0
            p_px = (pxx_flux[i - 1] - gb ** 2 * dens * MP * C ** 2) / (1 + gb ** 2 * Xi)
            p_loc = (1 - w) * p_px + w * P_tot_MC[i - 1]
            # momentum: F_px - Qpx - pxx_EM - γ₀β₀n₀·γβ·(m_p c² + P Ξ / n_loc) - P = 0, linear in γβ   (S1)
            gb_new = (F_px - Qpx - pxx_EM - p_loc) / (g0 * b0 * n0 * (MP * C ** 2 + p_loc * Xi / dens))
            ux_px[i - 1] = gb_new / math.sqrt(1 + gb_new ** 2) * C
            # energy: γβ·√(1+γβ²)·c·(n_loc m_p c² + Ξ P) = F_en - Qen - en_EM                            (S1)
            K = (F_en - Qen - en_EM) / (C * (dens * MP * C ** 2 + Xi * p_loc))
            y = (-1 + math.sqrt(1 + 4 * K * K)) / 2
            gb_new = math.copysign(math.sqrt(y), K)
            ux_en[i - 1] = gb_new / math.sqrt(1 + gb_new ** 2) * C
        else:
            rho0 = n0 * MP
            p_px = (pxx_flux[i - 1] - rho0 * u0 * ux * (1 + bux ** 2)) / (1 + bux ** 2 * Xi)
            p_loc = (1 - w) * p_px + w * P_tot_MC[i - 1]
            ux_px[i - 1] = _newton(lambda b: (F_px - Qpx - pxx_EM - rho0 * u0 * (b * C) * (1 + b * b) - (1 + b * b * Xi) * p_loc,
                                              -(rho0 * u0 * C * (1 + 3 * b * b) + 2 * b * Xi * p_loc)), u0 / C * 1.0e-4) * C   # (S2)
            ux_en[i - 1] = _newton(lambda u: (F_en - Qen - en_EM - 0.5 * rho0 * u0 * u * u * (1 + 1.25 * (u / C) ** 2)
                                              - Xi * p_loc * u * (1 + (u / C) ** 2),
                                              -(rho0 * u0 * u * (1 + 2.5 * (u / C) ** 2) + Xi * p_loc * (1 + 3 * (u / C) ** 2))), u0 * 1.0e-4)
        if i > n - 10:
            avg_px += ux_px[i - 1]; avg_en += ux_en[i - 1]
    if relativistic:                       # smooth, then rescale (smoothers.jl:436-457)
        smooth_profile(ux_px); smooth_profile(ux_en)
    avg_px /= 10; avg_en /= 10
    down = prob.x_grid_rg[1:n + 1] >= 0
    for arr, avg in ((ux_px, avg_px), (ux_en, avg_en)):
        sc = (u0 - u2) / (arr[0] - avg)
        arr[:] = sc * (arr - avg) + u2
        arr[down] = u2
    if not relativistic:                   # rescale, then smooth (smoothers.jl:541-566)
        smooth_profile(ux_px); smooth_profile(ux_en)
    return (1 - sm.SMMOE) * ux_px + sm.SMMOE * ux_en


def _newton(f_df, x0, tol=1.0e-14, maxit=10_000):
    x = x0
    for _ in range(maxit):
        f, df = f_df(x)
        dx = f / df
        x -= dx
        if abs(dx) <= tol * abs(x):
            break
    return x


def smooth_grid_par(prob, st: IterState, sm: SmoothingConfig, i_iter, pxx_flux, energy_flux, q_px, q_en, P_par, P_perp):
    """The part of `smooth_grid_par` (src/smoothers.jl:54-348) that changes state: the new profile tables.
    Returns True if the tables of `prob` were changed."""
    P, cfg = prob.params, prob.cfg
    n = P.n_grid
    # smoothers.jl:95-98.  `prof_weight_fac` is a by-value Real all the way down (main_loops.jl:367 -> iter_finalize.jl:65 ->
    # smoothers.jl:63) and nothing is returned: the damping rebinds a LOCAL, so every iteration starts again from the
    # configured "old-profile-weight" -- w0 * 1.15 (iterations 2-5) or w0 * 1.5 (6 on), floored at 10; it does not compound.
    w = sm.old_profile_weight
    if sm.increase_old_profile_weighting and i_iter != 1:
        w = max(10.0, w * (1.15 if i_iter < 6 else 1.5))
    st.prof_weight_fac = w                                       # (what this iteration used: reporting only)
    if not sm.smooth_shocks:
        return False
    ux_new = new_velocity_profile(prob, st, sm, pxx_flux, energy_flux, q_px, q_en, P_par + P_perp)
    xs = sm.artificial_smoothing_start_rg
    if xs < 0:                                                   # smoothers.jl:293-299
        i_trans = int(np.argmax(prob.x_grid_rg > xs)) - 1
        sc = -(ux_new[i_trans - 1] - ux_new[n - 1]) / math.atan(prob.x_grid_rg[i_trans])
        for i in range(i_trans, P.i_shock + 1):
            ux_new[i - 1] = -math.atan(prob.x_grid_rg[i]) * sc + ux_new[n - 1]
    ux_new = (ux_new + w * prob.ux[1:n + 1]) / (1 + w)           # smoothers.jl:305-307
    u0, g0 = P.u0, P.gam0
    n0 = float(sum(s.density * s.aa for s in cfg.species))
    e0 = n0 * MP * C ** 2
    for i in range(1, n + 1):                                    # smoothers.jl:311-345
        u = ux_new[i - 1]
        prob.ux[i] = u
        prob.gam_sf[i] = 1 / math.sqrt(1 - (u / C) ** 2)
        prob.utot[i] = u
        prob.beta_ef[i] = (u0 - u) / (C - u0 * u / C)
        prob.gam_ef[i] = 1 / math.sqrt(1 - prob.beta_ef[i] ** 2)
        z = (g0 * u0) / (prob.gam_sf[i] * u)
        comp = 1 + (math.sqrt(1 / 3 + 2 / 3 * z ** 2) - 1) * cfg.b_field_turbulence
        amp = 1 + (comp - 1) * cfg.b_field_amplify
        prob.btot[i] = cfg.B_mag_upstream * amp
        if cfg.use_custom_epsB:
            raise NotImplementedError("custom ε_B profile (src/smoothers.jl:338-344): εB_grid is a profile initialiser outside the path")
    return True


def populate_eps_target(prob):
    """`populate_ε_target!` (src/iter_init.jl:1-15), called at the top of every iteration (src/main_loops.jl:76-81)."""
    P = prob.params
    z_max = P.gam0 * P.beta0 / (prob.gam2 * prob.beta2)
    prefac = P.energy_transfer_frac / (z_max - 1)
    for i in range(1, P.n_grid + 1):
        if prob.ux[i] != P.u0:
            prob.eps_target[i - 1] = prefac * (P.gam0 * P.u0 / (prob.gam_sf[i] * prob.ux[i]) - 1)
    return prob.eps_target


def tcut_print(weight_coupled, spectra_coupled, n_tcuts: int, num_psd_mom_bins: int):
    """The side effect of `tcut_print` (src/io.jl:28-45), which main_loops calls at the end of EVERY iteration when time-cut
    tracking is on (src/main_loops.jl:383-389) -- the printing itself is commented out there, the in-place rewrite is not:
      weight_coupled[tcut, ion] < 1e-60            -> 1e-99
      sum(spectra_coupled[:, tcut, ion]) > 1e-99   -> the spectrum is divided by that sum (over all 0:psd_max entries)
      spectra_coupled[0:num_psd_mom_bins, tcut, ion] < 1e-60 -> 1e-99
    for tcut = 1..n_tcuts and every ion.  spectra_coupled is never reset, so from the second iteration on the
    kernel adds raw weights on top of a spectrum normalised to 1: replicated as written.
    weight_coupled: [n_ions][100], spectra_coupled: [n_ions][100][201] (C order; python index = Julia index - 1 for tcut and
    ion, equal for the momentum bin), modified in place."""
    n_ions = weight_coupled.shape[0]
    for ion in range(n_ions):
        for tc in range(n_tcuts):
            if weight_coupled[ion, tc] < 1.0e-60:
                weight_coupled[ion, tc] = 1.0e-99
            sp = spectra_coupled[ion, tc]
            tot = 0.0
            for v in sp:                     # left to right.  Julia's `sum` below its pairwise block size (1024) is an @simd loop that
                tot += v                     # LLVM may reassociate: agreement with the reference's normaliser to ~1 ulp, not order identity
            if tot > 1.0e-99:
                sp /= tot
            head = sp[:num_psd_mom_bins + 1]
            head[head < 1.0e-60] = 1.0e-99
    return weight_coupled, spectra_coupled


@dataclasses.dataclass
class IterFinal:
    q_esc_cal_px: float
    q_esc_cal_energy: float
    Gamma_downstream: float
    px_esc_flux_upstream: float
    energy_esc_flux_upstream: float
    profile_changed: bool


def iter_finalize(prob, st: IterState, sm: SmoothingConfig, i_iter: int, tallies_f64, layout, P_par, P_perp, e_dens) -> IterFinal:
    """`iter_finalize` (src/iter_finalize.jl:1-110) on the merged tallies of the iteration's last species and the
    K4 pressures; updates `st` and, when smoothing is on, the grid tables of `prob` in place."""
    L = layout
    sc = L.view(tallies_f64, "scalars")          # ΣP_downstream, ΣKE_downstream, pₓ_esc_upstream, energy_esc_upstream
    k = i_iter - 1
    st.px_esc_flux_upstream[k] = sc[2] / st.F_px_upstream
    st.energy_esc_flux_upstream[k] = sc[3] / st.F_energy_upstream
    set_Gamma_adiab_grid(st.Gamma_grid, i_iter, prob.x_grid_cm, st.Gamma2_RH, P_par, P_perp, e_dens)
    st.Gamma_downstream[k] = 1 + sc[0] / sc[1]
    st.q_esc_cal_px[k], st.q_esc_cal_energy[k] = q_esc_calcs(st.Gamma_downstream[k], prob.r_comp, st.r_RH, prob)
    n_avg = min(i_iter, 4)
    q_px_avg = float(np.mean(st.q_esc_cal_px[i_iter - n_avg:i_iter]))
    q_en_avg = float(np.mean(st.q_esc_cal_energy[i_iter - n_avg:i_iter]))
    # the reference rounds the fluxes to 13 decimal places to hide the summation-order noise of its OpenMP
    # ancestor (src/iter_finalize.jl:46-54); the atomic tallies here have the same noise
    pxx = np.round(L.view(tallies_f64, "pxx_flux"), 13)
    en = np.round(L.view(tallies_f64, "energy_flux"), 13)
    changed = smooth_grid_par(prob, st, sm, i_iter, pxx, en, q_px_avg, q_en_avg, P_par, P_perp)
    st.energy_esc_flux_upstream[k] = max(st.energy_esc_flux_upstream[k], 1.0e-99)
    st.px_esc_flux_upstream[k] = max(st.px_esc_flux_upstream[k], 1.0e-99)
    return IterFinal(st.q_esc_cal_px[k], st.q_esc_cal_energy[k], st.Gamma_downstream[k], st.px_esc_flux_upstream[k],
                     st.energy_esc_flux_upstream[k], changed)
