"""Host-side input builder for the transport path.

Restates, in numpy, exactly the pieces of the reference's driver that produce
the *inputs* of the per-particle loop: config parsing
(`src/data_input.jl`, `src/MonteCarloScattering.jl:68-260`), Rankine-Hugoniot
compression as coded (`src/initializers.jl:73-117`), the grid
(`src/initializers.jl:403-476`), the unmodified shock profile
(`src/initializers.jl:774-850`), PSD bin parameters
(`src/initializers.jl:216-285`, `src/MonteCarloScattering.jl:276-338`), the
Maxwell-Boltzmann injection distribution (`src/initializers.jl:1251-1453`), the
fast-push fluxes (`src/initializers.jl:1156-1222`) and `populate_ε_target!`
(`src/iter_init.jl:1-15`).  O(n_grid)/O(bins) host work; nothing here is on the
hot path.
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Optional, Sequence

import numpy as np

from . import capi
from .constants import MP, ME, C, QCGS, KB, B_CMB0, E_REL_PT, BETA_REL_FL, NUM_THERM_BINS, KEV

# src/mc_in.toml:84-130 -- the 45 stock momentum cutoffs [aa * m_p c]
STOCK_PCUTS = [
    0.01, 0.6, 1.6, 2.0, 4.5, 9.0, 30.0, 50.0, 200.0, 300.0, 500.0, 1000.0, 2000.0, 5000.0,
    1.000e4, 3.162e4, 1.000e5, 3.162e5, 1.000e6, 3.162e6, 1.000e7, 1.778e7, 3.162e7, 5.623e7,
    1.000e8, 1.778e8, 3.162e8, 5.623e8, 1.000e9, 1.778e9, 3.162e9, 5.623e9, 1.00e10, 1.778e10,
    3.162e10, 5.623e10, 1.000e11, 1.778e11, 3.162e11, 5.623e11, 1.000e12, 1.778e12, 3.162e12,
    5.623e12, 1.000e13,
]
STOCK_TCUTS = [1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 3e13]  # mc_in.toml:171

# src/initializers.jl:403-419
FIRST_ZONE = [-9.0, -8.0, -7.0, -6.0, -5.0, -4.5, -4.0, -3.5, -3.0, -2.5, -2.0, -1.8, -1.6, -1.4, -1.2,
              -1.0, -0.9, -0.8, -0.7, -0.6, -0.5, -0.4, -0.3, -0.2, -0.15, -0.1, -0.07, -0.05, -0.04,
              -0.03, -0.02, -0.015, -0.01, -3.0e-3, -1.0e-3]
EXTREMELY_FINE_SPACING = [-1.0e-4, -1.0e-7, 0.0, 1.0e-7, 1.0e-4]
DOWNSTREAM_SPACING = [1.0e-3, 1.0e-2, 2.0e-2, 3.0e-2, 5.0e-2, 7.0e-2, 0.1, 0.15, 0.2, 0.25, 0.3, 0.4,
                      0.5, 0.6, 0.8, 1.0]


@dataclasses.dataclass
class Species:
    """`Species` of src/utils.jl:72; aa = m/m_p, zz = charge number."""
    aa: float
    zz: float
    temperature: float   # K
    density: float       # cm^-3

    @property
    def mass(self) -> float:
        return self.aa * MP


@dataclasses.dataclass
class Config:
    """The mc_in.toml keys the transport path depends on (same key meaning;
    defaults are the measured configuration of SURVEY.md section 8d, i.e. the stock
    file with scattering and DSA enabled, protons only, uniform B)."""
    shock_speed: float = 5.0
    shock_speed_unit: str = "gamma"
    num_iterations: int = 1
    coarse_scattering_Ng: float = 100.0
    fine_scattering_Ng: float = 2000.0
    species: List[Species] = dataclasses.field(default_factory=lambda: [Species(1.0, 1.0, 1e6, 1.0)])
    input_distribution: int = 1
    injection_energy: float = 1e3
    injection_weights: bool = True
    maximum_energy: Sequence[float] = (0.0, 0.0, 1e10)
    gyrofactor: float = 1.0
    B_mag_upstream: float = 1e-5
    theta_B0: float = 0.0
    x_grid_limits: Sequence[float] = (-1e7, 1e1)
    FEB_upstream: Optional[Sequence[float]] = (-1e2, 0.0)
    FEB_downstream: Optional[Sequence[float]] = (0.0, 0.0)
    XSPEC: Sequence[float] = ()
    N_PTS_INJ: int = 10000
    N_PTS_PCUT: int = 10000
    N_PTS_PCUT_HI: int = 10000
    EN_PCUT_HI: float = 1_000_000.0
    momentum_cutoffs: Sequence[float] = tuple(STOCK_PCUTS)
    no_scatter: bool = False
    no_DSA: bool = False
    target_compression_ratio: float = -1.0
    maximum_age: float = 3.15e11
    TCUTS: Optional[Sequence[float]] = tuple(STOCK_TCUTS)
    use_retro: Optional[bool] = None
    fast_upstream_transport: bool = True
    proton_fast_transport_stop: float = -1.0
    electron_energy_mfp_threshold: Optional[float] = 1e4
    radiation_losses: bool = True
    redshift: float = 0.0
    B_CMBz: Optional[float] = None      # cosmo_calc.get_redshift is out of scope: give z or B_CMBz
    energy_transfer_frac: float = 0.0
    b_field_turbulence: float = 0.0
    b_field_amplify: float = 1.0
    use_custom_epsB: bool = False
    num_psd_bins_per_decade: Sequence[int] = (10, 10)
    psd_linear_cosine_bins: int = 119
    psd_log_theta_decs: int = 4
    use_custom_frg: bool = False
    EMNFC: float = 0.01
    INJFR: Optional[Sequence[float]] = None
    jet_shock_radius: float = 0.438     # pc   (mc_in.toml:192)
    JETFR: Sequence[float] = (0.0, 5.0)  # (sphere fraction, opening angle [deg]); first non-zero used
    # build-specific switches
    grid_variant: str = "intended"      # "intended" | "verbatim" (reference quirk G1)
    track_thermal: bool = True
    abs_charge: bool = True             # quirk Q11: pass |Z| (the reference's signed electron charge makes t_step < 0)
    state_fp32: bool = False            # the fp32-state variant of the transport kernel (BASELINE config[4])


_TOML_KEYS = {
    "shock-speed": "shock_speed", "shock-speed-unit": "shock_speed_unit", "num-iterations": "num_iterations",
    "coarse-scattering-Ng": "coarse_scattering_Ng", "fine-scattering-Ng": "fine_scattering_Ng",
    "input-distribution": "input_distribution", "injection-energy": "injection_energy",
    "injection-weights": "injection_weights", "maximum-energy": "maximum_energy", "gyrofactor": "gyrofactor",
    "B-mag-upstream": "B_mag_upstream", "theta-B0": "theta_B0", "x_grid_limits": "x_grid_limits",
    "FEB-upstream": "FEB_upstream", "FEB-downstream": "FEB_downstream", "XSPEC": "XSPEC",
    "N_PTS_INJ": "N_PTS_INJ", "N_PTS_PCUT": "N_PTS_PCUT", "N_PTS_PCUT_HI": "N_PTS_PCUT_HI",
    "EN_PCUT_HI": "EN_PCUT_HI", "momentum-cutoffs": "momentum_cutoffs", "no-scatter": "no_scatter",
    "no-DSA": "no_DSA", "target-compression-ratio": "target_compression_ratio", "maximum-age": "maximum_age",
    "TCUTS": "TCUTS", "use-retro": "use_retro", "fast-upstream-transport": "fast_upstream_transport",
    "proton-fast-transport-stop": "proton_fast_transport_stop",
    "electron-energy-mfp-threshold": "electron_energy_mfp_threshold", "radiation-losses": "radiation_losses",
    "redshift": "redshift", "energy-transfer-frac": "energy_transfer_frac",
    "b-field-turbulence": "b_field_turbulence", "b-field-amplify": "b_field_amplify",
    "use-custom-epsB": "use_custom_epsB", "num-psd-bins-per-decade": "num_psd_bins_per_decade",
    "psd-linear-cosine-bins": "psd_linear_cosine_bins", "psd-log-theta-decs": "psd_log_theta_decs",
    "use-custom-frg": "use_custom_frg", "EMNFC": "EMNFC", "INJFR": "INJFR",
    "jet-shock-radius": "jet_shock_radius", "JETFR": "JETFR",
}


def config_from_toml(path: str, **overrides) -> Config:
    """Read an mc_in.toml-shaped file (keys as in /root/reference/mc_in.toml)."""
    import tomli

    with open(path, "rb") as f:
        d = tomli.load(f)
    kw = {}
    for k, v in d.items():
        if k in _TOML_KEYS:
            kw[_TOML_KEYS[k]] = v
    if "AA_ION" in d:  # src/data_input.jl:165-184 parse_species
        sp = []
        for aa, zz, t, n in zip(d["AA_ION"], d["ZZ_ION"], d["TZ_ION"], d["DENZ_ION"]):
            if isinstance(aa, float) and math.isnan(aa):
                aa, zz = ME / MP, -1.0
            sp.append(Species(float(aa), float(zz), float(t), float(n)))
        kw["species"] = sp
    kw.update(overrides)
    return Config(**kw)


# --------------------------------------------------------------------------
def parse_shock_speed(v: float, unit: str):
    """src/data_input.jl:2-27"""
    if unit in ("gamma", "γ"):
        gam = v
        beta = math.sqrt(1 - 1 / gam ** 2)
        u = beta * C
    elif unit == "km/s":
        u = v * 1.0e5
        beta = u / C
        gam = 1 / math.sqrt(1 - beta ** 2)
    elif unit == "c":
        beta = v
        u = beta * C
        gam = 1 / math.sqrt(1 - beta ** 2)
    else:
        raise ValueError("shock-speed: unknown units provided with shock-speed-unit")
    return u, beta, gam


def calc_rRH(beta0: float, species: Sequence[Species]) -> float:
    """src/initializers.jl:73-117 AS CODED (quirk G2): the flag named `relativistic`
    is `beta0 < 0.02` and is negated, so every beta0 >= 0.02 takes the
    non-relativistic branch r = 8 / (2 + 6/M^2)."""
    relativistic = beta0 < BETA_REL_FL
    P0 = sum(s.density * s.temperature for s in species) * KB
    rho0 = sum(s.density * s.mass for s in species)
    if not relativistic:
        cs = math.sqrt(5.0 / 3.0 * P0 / rho0)
        M = beta0 * C / cs
        return 8 / (2 + 6 / M ** 2)
    raise NotImplementedError("calc_rRH_relativistic is called with a wrong arity in the reference (G2)")


def setup_grid(x_start_rg: float, x_stop_rg: float, variant: str = "intended") -> np.ndarray:
    """src/initializers.jl:436-476 -> x_grid_rg (index 0..n_grid+1).

    variant "verbatim" reproduces the committed arithmetic (quirk G1: the upstream
    step parses as ((log10(-x0) - 1)/27) - 1, giving a non-monotone block, and the
    downstream log block repeats the 1.0 edge); "intended" uses the evidently
    intended log spacing from x_start to -10 (27 edges) and from 1 to x_stop."""
    n_log_up, n_log_dn = 27, 16
    x = [-1.0e30]
    if variant == "verbatim":
        dlog = (math.log10(-x_start_rg) - 1) / n_log_up - 1
        x += [-(10.0 ** (math.log10(-x_start_rg) + k * (-dlog))) for k in range(n_log_up)]
    else:
        dlog = (math.log10(-x_start_rg) - 1) / (n_log_up - 1)   # "run from x_grid_start_rg to -10rg0"
        x += [-(10.0 ** (math.log10(-x_start_rg) - k * dlog)) for k in range(n_log_up)]
        x[n_log_up] = -10.0
    x += FIRST_ZONE + EXTREMELY_FINE_SPACING + DOWNSTREAM_SPACING
    x_end_man = x[-1]
    dlog = (math.log10(x_stop_rg) - math.log10(x_end_man)) / n_log_dn
    if variant == "verbatim":
        x += [10.0 ** (math.log10(x_end_man) + k * dlog) for k in range(n_log_dn)]
    else:
        x += [10.0 ** (math.log10(x_end_man) + (k + 1) * dlog) for k in range(n_log_dn)]
        x[-1] = float(x_stop_rg)
    x.append(1.0e30)
    return np.asarray(x, dtype=np.float64)


def set_psd_mom_bins(psd_mom_min: float, psd_mom_max: float, bpd: int):
    """src/initializers.jl:216-237 -> (num_psd_mom_bins, log10 bounds/(m_p c), index 0..n+1)"""
    n = int(math.trunc(math.log10(psd_mom_max / psd_mom_min) * bpd)) + 2
    log_p_min = math.log10(psd_mom_min / (MP * C))
    bounds = np.concatenate([[-99.0], log_p_min + np.arange(n + 1) / bpd])
    return n, bounds


def set_psd_angle_bins(bpd: int, lin_cos_bins: int, cos_fine: float, tht_min: float):
    """src/initializers.jl:265-285 -> (dcos, bounds[0..n+1])"""
    tht_fine = math.acos(cos_fine)
    ten_root = 10.0 ** (1 / bpd)
    log_bins = int(math.trunc(math.log10(tht_fine / tht_min) * bpd))
    b = [1.0e-99] + [tht_min * ten_root ** k for k in range(log_bins)]
    dcos = (cos_fine + 1) / lin_cos_bins
    b += [cos_fine - dcos * k for k in range(lin_cos_bins + 1)]
    return dcos, np.sort(np.asarray(b))


def get_pmax_cutoff(Emax_keV: float, Emax_per_aa_keV: float, pmax: float, aa: float) -> float:
    """src/ion_init.jl:55-72"""
    m = aa * MP
    E0 = m * C * C
    if Emax_keV > 0:
        g = 1 + Emax_keV * KEV / E0
        return m * C * math.sqrt(g * g - 1)
    if Emax_per_aa_keV > 0:
        g = 1 + Emax_per_aa_keV * KEV / E0
        return m * C * math.sqrt(g * g - 1)
    if pmax > 0:
        return pmax
    raise ValueError("Max CR energy not set in data_input, so can't set pmax_cutoff.")


def pcut_hi(energy_pcut_hi_keV: float, m: float) -> float:
    """src/ion_init.jl:74-82 (as coded: the non-relativistic branch returns a bare number)."""
    e = energy_pcut_hi_keV * KEV / (MP * C * C)
    if e < E_REL_PT:
        return math.sqrt(2 * e)
    return m * C * math.sqrt((e + 1) ** 2 - 1)


def create_inj_dist_momentum_range(m: float, T: float, nbins: int) -> np.ndarray:
    """src/initializers.jl:1389-1415"""
    E0 = m * C * C
    kT = KB * T
    kT_min, kT_max = 2.0e-3 * kT, 10 * kT
    if (kT / E0) < E_REL_PT:
        p_min, p_max = math.sqrt(2 * m * kT_min), math.sqrt(2 * m * kT_max)
    else:
        p_min = math.sqrt((kT_min + E0) ** 2 - E0 ** 2) / C
        p_max = math.sqrt((kT_max + E0) ** 2 - E0 ** 2) / C
    dp = (p_max - p_min) / nbins
    return p_min + dp * np.arange(nbins + 1)


def set_inj_dist_bins(inj_weight: bool, n_pts_inj: int, inp_distr: int, T_or_E: float, m: float, n0: float):
    """src/initializers.jl:1251-1328 in binned form -> (bin_ptot[b], bin_weight[b], bin_count[b]):
    every particle of bin b gets ptot = bin_ptot[b] and weight = bin_weight[b]; the per-particle
    arrays of the reference are np.repeat of these (set_inj_dist below).  O(bins) work.

    Deviation (quirk G6): the equal-weight loop starts its counter at 0, not 1, so
    there is no zero-momentum first particle (with scattering enabled that particle
    is 0/0 = NaN and the reference throws)."""
    if inp_distr not in (1, 2):
        raise ValueError("Code can only do inp_distr = 1 or 2.")
    if inp_distr == 2:
        E_inj = T_or_E * KEV
        E0 = m * C * C
        p = math.sqrt(2 * m * E_inj) if E_inj / E0 < E_REL_PT else math.sqrt(E_inj ** 2 - E0 ** 2) / C
        # as coded: weight = n0 / n_pts_tot with n_pts_tot from the (discarded) M-B pass
        _, _, cnt = set_inj_dist_bins(inj_weight, n_pts_inj, 1, 1e6, m, n0)
        n_tot = int(cnt.sum())
        return np.array([p]), np.array([n0 / n_tot]), np.array([n_pts_inj], dtype=np.int64)
    p_range = create_inj_dist_momentum_range(m, T_or_E, NUM_THERM_BINS)
    E0 = m * C * C
    kT = KB * T_or_E
    if (kT / E0) < E_REL_PT:
        E_range = p_range ** 2 / (2 * m * kT)
    else:
        E_range = np.hypot(p_range * C, E0) / kT
    f = np.exp(2 * np.log(p_range) - E_range)           # calc_MB_area_single_bin, :1368-1376
    bin_area = (p_range[1:] - p_range[:-1]) * (f[:-1] + f[1:]) / 2
    area_tot = float(np.sum(bin_area))
    centers = np.sqrt(p_range[:-1] * p_range[1:])
    if inj_weight:
        area_per_pt = area_tot / n_pts_inj
        counts = np.rint(bin_area / area_per_pt).astype(np.int64)   # round(Int, .) = ties-to-even
        n = int(counts.sum())
        return centers, np.full(len(centers), n0 / n), counts
    n_per_bin = n_pts_inj // NUM_THERM_BINS
    if n_per_bin < 5:
        raise ValueError("too few particles per bin; increase n_pts_inj")
    return centers, bin_area / area_tot / n_per_bin * n0, np.full(len(centers), n_per_bin, dtype=np.int64)


def set_inj_dist(inj_weight: bool, n_pts_inj: int, inp_distr: int, T_or_E: float, m: float, n0: float):
    """src/initializers.jl:1251-1328 -> (ptot[n], weight[n], n): the per-particle arrays."""
    bp, bw, cnt = set_inj_dist_bins(inj_weight, n_pts_inj, inp_distr, T_or_E, m, n0)
    return np.repeat(bp, cnt), np.repeat(bw, cnt), int(cnt.sum())


@dataclasses.dataclass
class Problem:
    """Everything the kernel boundary needs for one run."""
    cfg: Config
    params: capi.McsParams
    x_grid_rg: np.ndarray
    x_grid_cm: np.ndarray
    ux: np.ndarray
    uz: np.ndarray
    utot: np.ndarray
    gam_sf: np.ndarray
    gam_ef: np.ndarray
    beta_ef: np.ndarray
    btot: np.ndarray
    theta: np.ndarray
    pcuts: np.ndarray
    tcuts: np.ndarray
    x_spec: np.ndarray
    inj_fracs: np.ndarray
    eps_target: np.ndarray
    rg0: float
    r_comp: float
    beta2: float
    gam2: float
    psd_mom_bounds: np.ndarray
    psd_tht_bounds: np.ndarray
    Emax_keV: float
    Emax_per_aa_keV: float
    pmax: float
    x_fast_stop_rg: float
    i_fast_stop: int

    @property
    def n_grid(self) -> int:
        return int(self.params.n_grid)

    def grid_tables(self):
        return (self.x_grid_cm, self.ux, self.uz, self.utot, self.gam_sf, self.gam_ef, self.beta_ef, self.btot, self.theta)


def build_problem(cfg: Config) -> Problem:
    """The part of `main` (src/MonteCarloScattering.jl:60-493) that feeds particle_loop."""
    u0, beta0, gam0 = parse_shock_speed(cfg.shock_speed, cfg.shock_speed_unit)
    species = list(cfg.species)
    n_ions = len(species)
    if cfg.theta_B0 != 0:
        raise ValueError("program cannot currently handle oblique shocks. Adjust theta-B0.")  # data_input.jl:71-78

    # maximum energy (data_input.jl:29-50)
    em = list(cfg.maximum_energy)
    Emax = Emax_aa = pmax = 0.0
    if em[0] > 0:
        Emax = em[0]
    elif em[1] > 0:
        Emax_aa = em[1]
    elif em[2] > 0:
        pmax = em[2] * MP * C
    else:
        raise ValueError("ENMAX: at least one choice must be non-zero.")

    eta = cfg.gyrofactor
    B0 = cfg.B_mag_upstream
    rg0 = (gam0 * MP * C * C * beta0) / (QCGS * B0)           # MonteCarloScattering.jl:86

    xs, xe = cfg.x_grid_limits
    if xs >= 0 or xe <= 0:
        raise ValueError("x_grid_limits: start must be negative, stop positive")
    # get_feb (data_input.jl:127-150)
    if cfg.FEB_upstream is None:
        feb_up = xs * rg0
    else:
        if cfg.FEB_upstream[0] < 0:
            feb_up = cfg.FEB_upstream[0] * rg0
        elif cfg.FEB_upstream[1] < 0:
            feb_up = cfg.FEB_upstream[1] * 3.0856775814913674e18
        else:
            raise ValueError("FEB-upstream: at least one choice must be negative.")
        if feb_up / rg0 < xs:
            raise ValueError("FEB-upstream: upstream FEB must be within x_grid_start")
    use_prp = False
    if cfg.FEB_downstream is None:
        feb_dn = -1.0
    elif cfg.FEB_downstream[0] > 0:
        feb_dn = cfg.FEB_downstream[0] * rg0
    elif cfg.FEB_downstream[1] > 0:
        feb_dn = cfg.FEB_downstream[1] * 3.0856775814913674e18
    else:
        feb_dn, use_prp = 0.0, True

    pcuts = np.asarray(cfg.momentum_cutoffs, dtype=np.float64) * (MP * C)

    r_RH = calc_rRH(beta0, species)
    r_comp = r_RH if cfg.target_compression_ratio == -1 else cfg.target_compression_ratio
    beta2 = beta0 / r_comp                                       # calc_downstream, initializers.jl:43-50
    gam2 = 1 / math.sqrt(1 - beta2 ** 2)
    u2 = beta2 * C

    age_max = cfg.maximum_age if cfg.maximum_age >= 0 else -1.0
    do_retro = cfg.use_retro if cfg.use_retro is not None else (age_max > 0)
    x_fast_stop_rg = cfg.proton_fast_transport_stop if cfg.fast_upstream_transport else 0.0

    # parse_electron_critical_energy (data_input.jl:52-69)
    e_crit = cfg.electron_energy_mfp_threshold
    if e_crit is None or e_crit <= 0:
        pe_crit, game_crit = -ME * C, -1.0
    else:
        rm = e_crit * KEV / (ME * C * C)
        if rm < 1.0e-2:
            pe_crit, game_crit = math.sqrt(2 * ME * e_crit * KEV), 1.0
        else:
            pe_crit, game_crit = ME * C * math.sqrt((rm + 1) ** 2 - 1), rm + 1

    bpd_mom, bpd_tht = (int(v) for v in cfg.num_psd_bins_per_decade)
    do_tcuts = cfg.TCUTS is not None and len(cfg.TCUTS) > 0
    tcuts = np.asarray(cfg.TCUTS if do_tcuts else [], dtype=np.float64)
    if do_tcuts:
        if age_max < 0:
            raise ValueError("tcut tracking must be used with an accel time limit")
        if tcuts[-1] <= 10 * age_max:
            raise ValueError("TCUTS: final tcut must be much (10x) larger than age_max.")
    inj_fracs = np.asarray(cfg.INJFR if cfg.INJFR is not None else [1.0] * n_ions, dtype=np.float64)

    # grid (MonteCarloScattering.jl:263-266)
    x_grid_rg = setup_grid(xs, xe, cfg.grid_variant)
    if not use_prp:
        x_grid_stop = feb_dn
    else:
        x_grid_stop = xe * rg0
    n_grid = len(x_grid_rg) - 2
    x_grid_cm = x_grid_rg * rg0

    # PSD parameters (MonteCarloScattering.jl:276-338)
    cos_fine = 1 - 2 / (cfg.psd_linear_cosine_bins + 1)
    tht_fine = math.acos(cos_fine)
    tht_min = tht_fine / 10.0 ** cfg.psd_log_theta_decs
    if cfg.input_distribution == 1:
        Emin = min(s.temperature for s in species) * KB * cfg.EMNFC
    elif cfg.input_distribution == 2:
        Emin = cfg.injection_energy * KEV / 5
    else:
        raise ValueError("Unknown input distribution")
    m_min = min(s.mass for s in species)
    re_min = m_min * C * C
    if Emin < re_min / 1000:
        psd_mom_min = math.sqrt(2 * m_min * Emin)
    else:
        g = 1 + Emin / re_min
        psd_mom_min = m_min * C * math.sqrt(g * g - 1)
    m_max = max(s.mass for s in species)
    re_max = m_max * C * C
    if Emax > 0:
        g = 1 + Emax * KEV / re_max
        psd_mom_max = m_max * C * math.sqrt(g * g - 1)
    elif Emax_aa > 0:
        g = 1 + Emax_aa * KEV / (MP * C * C)
        psd_mom_max = m_max * C * math.sqrt(g * g - 1)
    else:
        psd_mom_max = pmax
    psd_mom_max *= 2 * gam0
    n_mom, mom_bounds = set_psd_mom_bins(psd_mom_min, psd_mom_max, bpd_mom)
    dcos, tht_bounds = set_psd_angle_bins(bpd_tht, cfg.psd_linear_cosine_bins, cos_fine, tht_min)
    n_tht = len(tht_bounds) - 2

    i_grid_feb = int(np.argmax(x_grid_cm > feb_up)) - 1           # MonteCarloScattering.jl:414
    B_CMBz = cfg.B_CMBz if cfg.B_CMBz is not None else B_CMB0 * (1 + cfg.redshift) ** 2

    # setup_profile (initializers.jl:774-850), unmodified shock
    ne = n_grid + 2
    ux = np.empty(ne); gsf = np.empty(ne); bef = np.empty(ne); gef = np.empty(ne); bt = np.empty(ne)
    for i in range(ne):
        if x_grid_cm[i] < 0:
            ux[i], gsf[i], bef[i], gef[i], bt[i] = u0, gam0, 0.0, 1.0, B0
        else:
            u = u0 / r_comp
            b = u / C
            ux[i] = u
            gsf[i] = 1 / math.sqrt(1 - b * b)
            bef[i] = (beta0 - b) / (1 - beta0 * b)
            gef[i] = 1 / math.sqrt(1 - bef[i] ** 2)
            z_comp = (gam0 * u0) / (gsf[i] * u)
            aux = math.sqrt((1 + 2 * z_comp ** 2) / 3)
            comp_fac = 1 + (aux - 1) * cfg.b_field_turbulence
            amp_fac = 1 + (comp_fac - 1) * cfg.b_field_amplify
            bt[i] = B0 * amp_fac
    uz = np.zeros(ne)
    utot = ux.copy()
    theta = np.full(ne, math.radians(cfg.theta_B0))
    eps_B = np.full(ne, 1.0e-99)                                   # initializers.jl:847
    if cfg.use_custom_epsB:
        # set_custom_εB! (initializers.jl:868-951) and the field it implies (:833-845): epsilon_B against the distance from the shock in
        # plasma skin depths -- a rising power law far upstream, a plateau of 1e-4 within 50 skin depths, a 1/x decay downstream -- and
        # B = sqrt(|8 pi epsilon_B e(x)|) with e(x) = (F_en0 + gam0 u0 e0) / u(x) - F_px0.
        #   E1  `comp_fac` is 0.0 when the function is called: the loop above assigns a `local comp_fac` (:791, :815), so epsilon_B2 = 0,
        #       the decay never ends (5e-3 / 0 = Inf) and the last branch (:945) is never taken.  As written.
        from types import SimpleNamespace
        from .iter_finalize import upstream_fluxes
        F_px, _, F_en = upstream_fluxes(SimpleNamespace(params=SimpleNamespace(u0=u0, beta0=beta0, gam0=gam0), cfg=cfg))
        n0_tot = sum(s.density * s.mass for s in species) / MP
        e0 = n0_tot * MP * C * C
        eps_B0 = B0 ** 2 / (8 * math.pi * e0)
        n0_electron = species[-1].density                          # "electron number density": the LAST species, whatever it is (:899)
        sigma = 2 * eps_B0 / gam0
        rg2sd = beta0 / math.sqrt(sigma * n0_tot / n0_electron)
        comp_fac_call = 0.0                                        # E1
        e_dens2 = (F_en + gam0 * u0 * e0) / ux[-1] - F_px
        eps_B2 = (B0 * comp_fac_call) ** 2 / (8 * math.pi * e_dens2)
        end_decay_rg = math.inf if eps_B2 == 0 else (5.0e-3 / eps_B2) / rg2sd
        for i in range(ne):
            x_sd = x_grid_rg[i] * rg2sd
            if x_sd < -50:
                eps_B[i] = max(1.04e-5 / abs(x_sd) ** 0.6, eps_B0)
            elif x_sd < 50:
                eps_B[i] = 1.0e-4
            elif x_grid_rg[i] < end_decay_rg:
                eps_B[i] = 5.0e-3 / x_sd
            else:
                eps_B[i] = eps_B2
            bt[i] = math.sqrt(abs(8 * math.pi * eps_B[i] * ((F_en + gam0 * u0 * e0) / ux[i] - F_px)))
    bmag2 = float(bt[-1])

    i_shock = int(np.nonzero(x_grid_rg <= 0)[0][-1])              # MonteCarloScattering.jl:478
    n_pts_max = max(cfg.N_PTS_PCUT, cfg.N_PTS_PCUT_HI)            # :488

    # populate_ε_target! (iter_init.jl:1-15); zones 1..n_grid
    eps = np.zeros(n_grid)
    z_max = gam0 * beta0 / (gam2 * beta2)
    prefac = cfg.energy_transfer_frac / (z_max - 1)
    for i in range(1, n_grid + 1):
        if ux[i] != u0:
            eps[i - 1] = prefac * (gam0 * u0 / (gsf[i] * ux[i]) - 1)

    i_fast_stop = int(np.argmax(x_grid_rg > x_fast_stop_rg)) - 1 if cfg.fast_upstream_transport else 0

    P = capi.McsParams()
    P.abi_version = capi.MCS_ABI_VERSION
    P.n_ions, P.n_grid, P.n_itrs = n_ions, n_grid, cfg.num_iterations
    P.n_pts_max = n_pts_max
    P.i_grid_feb, P.i_shock = i_grid_feb, i_shock
    P.num_psd_mom_bins, P.num_psd_tht_bins = n_mom, n_tht
    P.psd_bins_per_dec_mom, P.psd_bins_per_dec_tht = bpd_mom, bpd_tht
    P.psd_cos_fine, P.psd_dcos, P.psd_tht_min, P.psd_mom_min = cos_fine, dcos, tht_min, psd_mom_min
    P.gam0, P.beta0, P.u0, P.u2, P.bmag2 = gam0, beta0, u0, u2, bmag2
    P.pe_crit, P.game_crit, P.eta_mfp = pe_crit, game_crit, eta
    P.energy_transfer_frac = cfg.energy_transfer_frac
    P.feb_upstream, P.feb_downstream, P.x_grid_stop = feb_up, feb_dn, x_grid_stop
    P.B_CMBz, P.age_max = B_CMBz, age_max
    P.xn_per_fine, P.xn_per_coarse = cfg.fine_scattering_Ng, cfg.coarse_scattering_Ng
    P.use_custom_epsB = int(cfg.use_custom_epsB)
    P.do_rad_losses, P.do_retro, P.do_tcuts = int(cfg.radiation_losses), int(do_retro), int(do_tcuts)
    P.dont_DSA, P.dont_scatter, P.use_custom_frg = int(cfg.no_DSA), int(cfg.no_scatter), int(cfg.use_custom_frg)
    P.track_thermal = int(cfg.track_thermal)
    P.state_fp32 = int(cfg.state_fp32)

    return Problem(cfg=cfg, params=P, x_grid_rg=x_grid_rg, x_grid_cm=x_grid_cm, ux=ux, uz=uz, utot=utot,
                   gam_sf=gsf, gam_ef=gef, beta_ef=bef, btot=bt, theta=theta, pcuts=pcuts, tcuts=tcuts,
                   x_spec=np.asarray(cfg.XSPEC, dtype=np.float64), inj_fracs=inj_fracs, eps_target=eps,
                   rg0=rg0, r_comp=r_comp, beta2=beta2, gam2=gam2, psd_mom_bounds=mom_bounds,
                   psd_tht_bounds=tht_bounds, Emax_keV=Emax, Emax_per_aa_keV=Emax_aa, pmax=pmax,
                   x_fast_stop_rg=x_fast_stop_rg, i_fast_stop=i_fast_stop)


@dataclasses.dataclass
class Injection:
    """Host part of init_pop (src/initializers.jl:977-1134) for one species.  The momentum
    discretisation is kept in binned form (O(bins)); `ptot_pf` / `weight` expand it to the
    reference's per-particle arrays on demand (CPU oracle, tests)."""
    n_pts_use: int
    bin_ptot: np.ndarray
    bin_weight: np.ndarray
    bin_count: np.ndarray
    x_start_cm: float
    i_grid_start: int
    relativistic: bool
    fast_push: bool
    pxx_flux: np.ndarray
    pxz_flux: np.ndarray
    energy_flux: np.ndarray

    @property
    def bin_start(self) -> np.ndarray:
        return np.concatenate([[0], np.cumsum(self.bin_count)]).astype(np.int64)

    @property
    def ptot_pf(self) -> np.ndarray:
        return np.repeat(self.bin_ptot, self.bin_count)

    @property
    def weight(self) -> np.ndarray:
        return np.repeat(self.bin_weight, self.bin_count)


def init_pop_host(prob: Problem, i_ion: int) -> Injection:
    """Momentum discretisation, weights and analytic fast-push fluxes; the
    per-particle pitch/phase draws are the device kernel K3 (mcs_init_pop)."""
    cfg, P = prob.cfg, prob.params
    sp = cfg.species[i_ion - 1]
    ng = prob.n_grid
    zeros = np.zeros(ng)
    if not cfg.fast_upstream_transport:
        T_or_E = sp.temperature if cfg.input_distribution == 1 else cfg.injection_energy
        bp, bw, cnt = set_inj_dist_bins(cfg.injection_weights, cfg.N_PTS_INJ, cfg.input_distribution, T_or_E, sp.mass, sp.density)
        x0 = cfg.x_grid_limits[0] * prob.rg0 - 10 * prob.rg0 * cfg.gyrofactor
        return Injection(int(cnt.sum()), bp, bw, cnt, x0, 0, False, False, zeros, zeros.copy(), zeros.copy())
    if cfg.input_distribution > 1:
        raise ValueError("fast push will only work with thermal input distr.")
    i_stop = prob.i_fast_stop
    relativistic = P.beta0 >= BETA_REL_FL
    density_ratio = P.u0 / prob.ux[i_stop]
    if relativistic:
        density_ratio *= P.gam0 / prob.gam_sf[i_stop]
    temp_ratio = density_ratio ** (5.0 / 3.0) / density_ratio
    if KB * sp.temperature * temp_ratio > 4 * sp.mass * C * C * E_REL_PT:
        raise ValueError("Fast push cannot work because highest energy thermal particles become mildly relativistic.")
    pxx, pxz, en = zeros.copy(), zeros.copy(), zeros.copy()
    if i_ion == 1:   # F_update! (initializers.jl:1156-1222)
        P0 = sum(s.density * s.temperature for s in cfg.species) * KB
        rho0 = sum(s.density * s.mass for s in cfg.species)
        G = 5.0 / 3.0
        Xi = G / (G - 1)
        for i in range(1, i_stop + 1):
            u_c = prob.ux[i]; b_c = u_c / C; g_c = prob.gam_sf[i]; gb = g_c * b_c
            dr = (P.gam0 * P.u0) / (g_c * u_c)
            rho_c = rho0 * dr
            P_c = P0 * dr ** G
            if not relativistic:
                Fpx = rho_c * u_c ** 2 * (1 + b_c ** 2) + P_c * (1 + Xi * b_c ** 2)
                Fen = rho_c / 2 * u_c ** 3 * (1 + 1.25 * b_c ** 2) + P_c * u_c * Xi * (1 + b_c ** 2)
            else:
                e_c = rho_c * C * C
                Fpx = P_c + gb ** 2 * (e_c + Xi * P_c)
                Fen = gb * g_c * C * (e_c + Xi * P_c) - gb * C * e_c
            pxx[i - 1], en[i - 1] = Fpx, Fen
    bp, bw, cnt = set_inj_dist_bins(cfg.injection_weights, cfg.N_PTS_INJ, cfg.input_distribution,
                                    sp.temperature * temp_ratio, sp.mass, sp.density)
    return Injection(int(cnt.sum()), bp, bw, cnt, prob.x_fast_stop_rg * prob.rg0, i_stop, relativistic, True, pxx, pxz, en)
