"""Host side of the tally consumers: the part of `ion_finalize`
(src/ion_finalize.jl:26-48) that turns the PSD into spectra and pressures.

The reductions over the 22 MB histograms run on the device (`mcs_dndp_cr`, `mcs_thermo_calcs`
in include/mcs.h); this module only builds the O(bins) / O(n_grid) tables they take and
mirrors the reference's call order.  Consumer quirks C1-C6 are listed in DESIGN.md section 3b.
"""
from __future__ import annotations

import ctypes as ct
import dataclasses
import math

import numpy as np

from . import capi
from .constants import C, KB, MP

PC_CM = 3.0856775814913674e18      # UnitfulAstro.pc in cm


def angle_edges_intended(prob) -> np.ndarray:
    """theta (rad) for the log region then cosines for the linear region, in the order
    `set_psd_angle_bins` builds them before its `sort!` (src/initializers.jl:272-279; quirk C2)."""
    P, cfg = prob.params, prob.cfg
    bpd = P.psd_bins_per_dec_tht
    lin = cfg.psd_linear_cosine_bins
    log_bins = P.num_psd_tht_bins - lin                # entries 1..log_bins hold theta
    b = [1.0e-99] + [P.psd_tht_min * (10.0 ** (1 / bpd)) ** k for k in range(log_bins)]
    b += [P.psd_cos_fine - P.psd_dcos * k for k in range(lin + 1)]
    b = np.asarray(b)
    assert len(b) == P.num_psd_tht_bins + 2, (len(b), P.num_psd_tht_bins)
    return b


def find_shock_index(x_grid: np.ndarray) -> int:
    """src/particle_counter.jl:936-947 (x_grid indexed 0..n_grid+1)."""
    for i in range(len(x_grid) - 1):
        if x_grid[i] == 0 or x_grid[i] * x_grid[i + 1] < 0:
            return i
    return 0


def set_grid_volumes(prob, i_ion: int):
    """`set_grid_volumes!` (src/particle_counter.jl:1466-1524) -> (zone_pop, zone_vol), zones 1..n_grid."""
    P, cfg = prob.params, prob.cfg
    x, ux, gsf = prob.x_grid_cm, prob.ux, prob.gam_sf
    n = P.n_grid
    i_shock = find_shock_index(x)
    dx = np.diff(x)                                  # dx[i] = x[i+1]-x[i], i = 0..n_grid
    sph = jet_sphere_fraction(cfg)
    jet_rad_cm = cfg.jet_shock_radius * PC_CM
    surf = np.zeros(n + 1)
    rad_min = jet_rad_cm - x[i_shock]
    for i in range(i_shock - 1, 0, -1):
        rad_max = rad_min + dx[i] / P.gam0
        surf[i] = math.pi * (rad_max + rad_min) ** 2 * sph
        rad_min = rad_max
    rad_max = jet_rad_cm - x[i_shock]
    for i in range(i_shock, n + 1):
        rad_min = rad_max - dx[i] / P.gam0
        surf[i] = math.pi * (rad_max + rad_min) ** 2 * sph
        rad_max = rad_min
    n0 = cfg.species[i_ion - 1].density
    zone_pop = np.zeros(n)
    zone_vol = np.zeros(n)
    for i in range(1, n + 1):
        dwell = dx[i] / ux[i]
        F_up = P.gam0 * n0 * P.beta0 * C
        zone_pop[i - 1] = F_up * surf[i] * dwell
        density_pf = P.gam0 * ux[1] / (gsf[i] * ux[i])
        zone_vol[i - 1] = zone_pop[i - 1] / density_pf
    return zone_pop, zone_vol


def jet_sphere_fraction(cfg) -> float:
    """`parse_jet_frac` (src/data_input.jl:153-167)."""
    if cfg.JETFR is None:
        return 0.0
    frac, ang = cfg.JETFR
    if 0 < frac <= 1:
        return float(frac)
    if 0 < ang <= 180:
        return (1 - math.cos(math.radians(ang))) / 2
    raise ValueError("JETFR: Unphysical values entered.")


@dataclasses.dataclass
class ConsumerTables:
    mom_log_cgs: np.ndarray
    mom_edge_cgs: np.ndarray
    cos_edge: np.ndarray
    cos_center: np.ndarray
    pt_center: np.ndarray
    zone_pop: np.ndarray
    zone_vol: np.ndarray
    density_loc: np.ndarray
    cold_pressure: np.ndarray
    rest_energy: float
    mc: float
    n0: float
    gam0: float
    therm_from_hist: int = 1

    def as_struct(self) -> capi.McsConsumerIn:
        s = capi.McsConsumerIn()
        for f in ("mom_log_cgs", "mom_edge_cgs", "cos_edge", "cos_center", "pt_center", "zone_pop", "density_loc",
                  "cold_pressure"):
            a = getattr(self, f)
            assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
            setattr(s, f, a.ctypes.data_as(capi.c_double_p))
        s.rest_energy, s.mc, s.n0, s.gam0 = self.rest_energy, self.mc, self.n0, self.gam0
        s.therm_from_hist = int(self.therm_from_hist)
        return s


def consumer_tables(prob, i_ion: int, therm_from_hist: bool = True) -> ConsumerTables:
    """Tables of get_dNdp_cr (src/particle_counter.jl:46-62), transform_psd_corners
    (src/transformers.jl:646-660) and thermo_calcs (src/thermo_calcs.jl:55-80, 246, 258-266)."""
    P, cfg = prob.params, prob.cfg
    sp = cfg.species[i_ion - 1]
    nm, nt = P.num_psd_mom_bins, P.num_psd_tht_bins
    lin = cfg.psd_linear_cosine_bins
    mb = np.asarray(prob.psd_mom_bounds, dtype=np.float64)           # log10(p / m_p c), index 0..nm+1
    mom_edge = np.ascontiguousarray(10.0 ** mb * (MP * C))           # C1: cgs
    mom_log = np.ascontiguousarray(np.log10(mom_edge))
    tb = angle_edges_intended(prob)                                  # C2
    j = np.arange(nt + 2)
    cos_edge = np.ascontiguousarray(np.where(j > nt - lin, -tb, -np.cos(tb)))
    cos_center = np.ascontiguousarray(0.5 * (cos_edge[:-1] + cos_edge[1:]))            # thermo_calcs.jl:57-73
    pt_center = np.ascontiguousarray(10.0 ** (0.5 * (mb[:-1] + mb[1:])) * (MP * C))    # thermo_calcs.jl:75-80
    zone_pop, zone_vol = set_grid_volumes(prob, i_ion)
    x, ux, gsf = prob.x_grid_cm, prob.ux, prob.gam_sf
    n = P.n_grid
    with np.errstate(divide="ignore", invalid="ignore"):
        density_loc = P.gam0 * P.beta0 * sp.density / np.sqrt(gsf[1:n + 1] ** 2 - 1)
    cold_pressure = density_loc ** (5.0 / 3.0) * KB * sp.temperature
    m = sp.aa * MP
    return ConsumerTables(mom_log, mom_edge, cos_edge, cos_center, pt_center,
                          np.ascontiguousarray(zone_pop), np.ascontiguousarray(zone_vol),
                          np.ascontiguousarray(density_loc), np.ascontiguousarray(cold_pressure),
                          m * C * C, m * C, sp.density, P.gam0, int(therm_from_hist))


@dataclasses.dataclass
class IonFinal:
    """What `ion_finalize` returns for the transport consumers (src/ion_finalize.jl:78-82)."""
    dNdp_cr: np.ndarray          # [3][n_grid][nmom+2]: shock, plasma, ISM frame; normalised dN/dp
    zone_pop: np.ndarray
    P_psd_par: np.ndarray
    P_psd_perp: np.ndarray
    energy_density_psd: np.ndarray
    diag: np.ndarray


def ion_finalize(prob, backend, i_ion: int, therm_from_hist: bool = True) -> IonFinal:
    """get_normalized_dNdp (CR part) then thermo_calcs, on the backend's resident tallies."""
    tabs = consumer_tables(prob, i_ion, therm_from_hist)
    dndp, diag = backend.dndp_cr(tabs)
    ppar, pperp, edens = backend.thermo_calcs(tabs)
    return IonFinal(dndp, tabs.zone_pop, ppar, pperp, edens, diag)


# ---- photon post-processing -------------------------------------------------------------------------
# photon_calcs.jl:11-19: the common energy grid of the emission spectra
PHOTON_E_MIN_MEV = 1.0e-13
PHOTON_BINS_PER_DEC = 10
PHOTON_SYNCH_E_MAX_MEV = 1.0e5
MEV_ERG = 1.602176634e-6
KPC_CM = 3.0856775814913674e21


@dataclasses.dataclass
class PhotonSynch:
    """What `photon_synch` computes per grid zone (src/photon_synch.jl:28-133) -- it writes it to photon_synch_grid.dat."""
    energy_MeV: np.ndarray         # [n_photon]
    emis_erg_s: np.ndarray         # [n_grid][n_photon]  dP/d(ln E) emitted by the zone
    energy_flux_MeV: np.ndarray    # [n_grid][n_photon]  MeV / (cm^2 s) per d(ln E) at Earth, floor 1e-99
    photon_flux: np.ndarray        # [n_grid][n_photon]  photons / (cm^2 s) per d(ln E) at Earth, floor 1e-99


def photon_synch(prob, backend, ion_fin: IonFinal, i_ion: int, jet_dist_kpc: float = 1.0e6, redshift: float = 0.0) -> PhotonSynch:
    """The synchrotron branch of `photon_calcs` (src/photon_calcs.jl:27-113) for an electron species: per zone, the fold of
    src/synch_emission.jl over the plasma-frame dN/dp (frame 2 of get_dNdp_cr; the thermal part is empty, quirk C4), on the
    device (K5), then photon_synch's conversion to fluxes at Earth (src/photon_synch.jl:74-108) with the luminosity distance
    of photon_calcs.jl:40.  The photon stack is dead code in the reference: followed as
    specification.  Inverse Compton: `photon_ic` below; pion decay (nuclei): `photon_pion`; the sum of the three: `summed_emission`."""
    P = prob.params
    sp = prob.cfg.species[i_ion - 1]
    if sp.aa >= 1:
        raise ValueError("photon_synch: synchrotron emission is computed for the electron species (aa < 1)")
    tabs = consumer_tables(prob, i_ion)
    n_photon = int(math.log10(PHOTON_SYNCH_E_MAX_MEV / PHOTON_E_MIN_MEV) * PHOTON_BINS_PER_DEC)      # photon_calcs.jl:51
    E_erg, emis = backend.photon_synch(ion_fin.dNdp_cr[1], tabs.mom_edge_cgs, tabs.mc, n_photon, PHOTON_E_MIN_MEV, PHOTON_BINS_PER_DEC)
    dist_lum = jet_dist_kpc * (1 + redshift) * KPC_CM
    E_MeV = E_erg / MEV_ERG
    flux_erg = np.maximum(emis / (4 * math.pi * dist_lum ** 2), 1.0e-99)
    eflux_MeV = np.where(flux_erg > 1.0e-99, flux_erg / MEV_ERG, 1.0e-99)
    pflux = np.where(eflux_MeV <= 1.0e-99, 1.0e-99, eflux_MeV / E_MeV[None, :])
    return PhotonSynch(E_MeV, emis, eflux_MeV, pflux)


# ---- inverse Compton (src/photon_calcs.jl:116-138 -> src/inverse_compton.jl) ---------------------------------------------------
PHOTON_IC_E_MIN_MEV = 1.0e-2          # photon_calcs.jl:18
PHOTON_E_MAX_MEV = 1.0e12             # photon_calcs.jl:11
T_CMB0_K = 2.725                      # constants.jl:12
WIEN_B_NU = 5.879e10                  # Hz / K (inverse_compton.jl:167)
H_PLANCK = 6.62607015e-27             # erg s
ME = 9.1093837015e-28                 # g


def photon_field_cmb(redshift: float = 0.0):
    """`photon_field!` (src/inverse_compton.jl:313-383): the CMB at the source's redshift as 60 logarithmic frequency bins between
    nu_peak / 30 and 20 nu_peak -> (photon energy / m_e c^2, photons per cm^3) per bin."""
    T = T_CMB0_K * (1 + redshift)
    nu_peak = WIEN_B_NU * T
    n_nu = 60
    log_nu_min = math.log10(nu_peak / 30)
    dlog = (math.log10(nu_peak * 20) - log_nu_min) / n_nu
    con_f1 = 8 * math.pi * H_PLANCK / C ** 3
    con_f2 = H_PLANCK / (KB * T)
    alpha, dens = np.zeros(n_nu), np.zeros(n_nu)
    for j in range(1, n_nu + 1):
        log_nu1 = log_nu_min + (j - 1) * dlog
        nu1, nu2 = 10.0 ** log_nu1, 10.0 ** (log_nu1 + dlog)
        nu_avg = math.sqrt(nu1 * nu2)
        exp_fac = math.exp(min(con_f2 * nu_avg, 200.0))
        u = (nu2 - nu1) * con_f1 * nu_avg ** 3 / (exp_fac - 1)
        e = H_PLANCK * nu_avg
        alpha[j - 1] = e / (ME * C * C)
        dens[j - 1] = u / e
    return alpha, dens


def ic_cone_last_bin(cos_edge: np.ndarray, jet_sph_frac: float, num_psd_tht_bins: int) -> int:
    """`findfirst(>(2 jet_sph_frac - 1), cos_bounds)` (src/inverse_compton.jl:215): the last angle bin whose electrons can send
    photons to Earth; every bin when nothing exceeds the bound (a full sphere) or the hit lies past the last bin (mcs_ic.h, I2)."""
    hit = np.nonzero(cos_edge > 2 * jet_sph_frac - 1)[0]
    j = int(hit[0]) if len(hit) else num_psd_tht_bins
    return min(j, num_psd_tht_bins)


@dataclasses.dataclass
class PhotonIC:
    """What `photon_IC` computes per grid zone (src/inverse_compton.jl:36-188) -- it writes it to photon_IC_grid.dat."""
    energy_MeV: np.ndarray         # [n_photon]
    emis_erg: np.ndarray           # [n_grid][n_photon]  energy flux per d(ln E) at Earth, erg / (cm^2 s), floor 1e-99
    energy_flux_MeV: np.ndarray    # [n_grid][n_photon]  the same in MeV / (cm^2 s), floor 1e-99
    photon_flux: np.ndarray        # [n_grid][n_photon]  photons / (cm^2 s) per d(ln E), floor 1e-99
    ic_photon_sum: np.ndarray      # [n_grid][n_photon]  1e-99 + emis / E (the array get_summed_emission reads, :100-104)
    d2N_ef: object = None          # [n_grid][ntht+2][nmom+2] get_dNdp_2D's array, if it was downloaded


def photon_ic(prob, backend, i_ion: int, jet_dist_kpc: float = 1.0e6, redshift: float = 0.0, jet_sph_frac=None,
              download_d2n: bool = False, therm_from_hist: bool = True) -> PhotonIC:
    """The inverse-Compton branch of `photon_calcs` (src/photon_calcs.jl:116-138) for the electron species: `get_dNdp_2D`
    (src/particle_counter.jl:343-627 as called at src/ion_finalize.jl:50-59: the electrons' d2N/dp dcos in the ISM frame) and
    `photon_IC` -> `IC_emission_FCJ` on the CMB of `photon_field!` (src/inverse_compton.jl), both on the device (K6) on the
    histograms K1 left there; then photon_IC's conversions (:93-133).  Dead code in the reference, followed as specification
    (include/mcs_ic.h: I1-I3)."""
    P = prob.params
    sp = prob.cfg.species[i_ion - 1]
    if sp.aa >= 1:
        raise ValueError("photon_ic: inverse-Compton emission is computed for the electron species (aa < 1)")
    if jet_sph_frac is None:
        jet_sph_frac = jet_sphere_fraction(prob.cfg)
    if not 0 < jet_sph_frac <= 1:
        raise ValueError("photon_ic: the jet must cover a fraction 0 < f <= 1 of the sphere (JETFR)")
    tabs = consumer_tables(prob, i_ion, therm_from_hist)
    if _takes_download(backend):
        d2n = backend.dndp_2d(tabs, P.gam0, P.beta0, download=download_d2n)
    else:
        d2n = backend.dndp_2d(tabs, P.gam0, P.beta0)
    n_photon = int(math.log10(PHOTON_E_MAX_MEV / PHOTON_IC_E_MIN_MEV) * PHOTON_BINS_PER_DEC)          # photon_calcs.jl:50
    alpha_in, n_in = photon_field_cmb(redshift)
    j_max = ic_cone_last_bin(tabs.cos_edge, jet_sph_frac, P.num_psd_tht_bins)
    dist_lum = jet_dist_kpc * (1 + redshift) * KPC_CM
    beam_area = 4 * math.pi * dist_lum ** 2 * jet_sph_frac
    E_erg, emis = backend.photon_ic(tabs.mom_edge_cgs, tabs.mc, j_max, alpha_in, n_in, n_photon, PHOTON_IC_E_MIN_MEV, PHOTON_BINS_PER_DEC, beam_area)
    E_MeV = E_erg / MEV_ERG
    eflux_MeV = np.where(emis > 1.0e-99, emis / MEV_ERG, 1.0e-99)
    pflux = np.where(eflux_MeV <= 1.0e-99, 1.0e-99, eflux_MeV / E_MeV[None, :])
    ic_sum = 1.0e-99 + np.where(emis > 1.0e-99, emis / E_erg[None, :], 0.0)
    return PhotonIC(E_MeV, emis, eflux_MeV, pflux, ic_sum, d2n)


# ---- pion decay (src/photon_calcs.jl:66-88 -> src/photon_pion_decay.jl -> src/pion_kafexhiu.jl) ------------------------------------
PHOTON_PION_E_MIN_MEV = 1.0           # photon_calcs.jl:15


@dataclasses.dataclass
class PhotonPion:
    """What `photon_pion_decay` computes per grid zone (src/photon_pion_decay.jl:40-183) -- it writes it to photon_pion_decay_grid.dat."""
    energy_MeV: np.ndarray         # [n_photon]
    emis_erg_s: np.ndarray         # [n_grid][n_photon]  dP/d(ln E) emitted by the zone (pion_kafexhiu), floor 1e-99
    energy_flux: np.ndarray        # [n_grid][n_photon]  emis / (4 pi d_lum^2): erg / (cm^2 s) per d(ln E) at Earth, floor 1e-99
    photon_flux: np.ndarray        # [n_grid][n_photon]  energy_flux / E, floor 1e-99
    pion_photon_sum: np.ndarray    # [n_grid][n_photon]  this species' term of the array get_summed_emission reads (:118-125)
    n_pion_specs: int              # nuclei species of the run (:69)


def pion_scaling_factor(cfg, aa: float) -> float:
    """Baring et al. (1999) eq. 26 summed over the target nuclei (src/pion_kafexhiu.jl:60-65)."""
    n1 = cfg.species[0].density
    return float(sum((aa ** 0.375 + sp.aa ** 0.375 - 1) ** 2 * sp.density / n1 for sp in cfg.species if sp.aa >= 1))


def photon_pion(prob, backend, ion_fin: IonFinal, i_ion: int, jet_dist_kpc: float = 1.0e6, redshift: float = 0.0, i_data: int = 1) -> PhotonPion:
    """The pion-decay branch of `photon_calcs` (src/photon_calcs.jl:66-88) for a nucleus species: per zone, the fold of
    src/pion_kafexhiu.jl (Kafexhiu et al. 2014) over the plasma-frame dN/dp (frame 2 of get_dNdp_cr; the thermal histogram is empty,
    quirk C4) against the zone's thermal protons at rest, on the device (K7), then photon_pion_decay's conversion to fluxes at Earth
    (src/photon_pion_decay.jl:112-125).  Dead code in the reference, followed as specification (include/mcs_pion.h: P1-P3)."""
    P, cfg = prob.params, prob.cfg
    sp = cfg.species[i_ion - 1]
    if sp.aa < 1:
        raise ValueError("photon_pion: pion-decay emission is computed for nuclei (aa >= 1)")
    tabs = consumer_tables(prob, i_ion)
    n = P.n_grid
    with np.errstate(divide="ignore", invalid="ignore"):
        target = cfg.species[0].density * (P.gam0 * P.beta0) / np.sqrt(np.asarray(prob.gam_sf)[1:n + 1] ** 2 - 1)      # :62-63
    n_photon = int(math.log10(PHOTON_E_MAX_MEV / PHOTON_PION_E_MIN_MEV) * PHOTON_BINS_PER_DEC)                   # photon_calcs.jl:49
    E_erg, emis = backend.photon_pion(ion_fin.dNdp_cr[1], tabs.mom_edge_cgs, tabs.mc, sp.aa, np.ascontiguousarray(target),
                                      pion_scaling_factor(cfg, sp.aa), n_photon, PHOTON_PION_E_MIN_MEV, PHOTON_BINS_PER_DEC, i_data)
    dist_lum = jet_dist_kpc * (1 + redshift) * KPC_CM
    eflux = emis / (4 * math.pi * dist_lum ** 2)
    lit = eflux >= 1.0e-99
    eflux = np.where(lit, eflux, 1.0e-99)
    psum = np.where(lit, eflux / E_erg[None, :], 0.0)
    pflux = np.where(eflux <= 1.0e-99, 1.0e-99, eflux / E_erg[None, :])
    return PhotonPion(E_erg / MEV_ERG, emis, eflux, pflux, psum, sum(1 for s in cfg.species if s.aa >= 1))


# ---- get_summed_emission (src/get_summed_emission.jl:37-413): the arithmetic of it ------------------------------------------------------
# The function is 834 lines of which 700 read the photon_*_grid.dat files back in and write histograms; its calculation is restated here
# on the arrays the three emission routines above return: the plasma-frame spectra (pion decay, synchrotron) are Doppler-shifted into the
# ISM frame, the zones are summed into the photon shells, and the three processes are laid onto one energy grid.  Dead code like the rest
# of the photon stack, followed as specification; where it cannot run as written:
#   S1  `log_energy_MeV_in` is overwritten with the LINEAR energies (:136, :141) and "dlogE" is then taken as the difference of two of them
#       (:150): the logarithmic bin width 1 / bins_per_decade is used, as :258 does;
#   S2  the three factors of the Lorentz factor (:188-198) are applied after the loop over the zones with the LAST zone's value: every zone
#       gets its own;
#   S3  `findnext` returns nothing for a photon shifted beyond the energy grid (:178-179): dropped;
#   S4  `sum_synch_IC_spectra()` is not defined: synchrotron and inverse-Compton bins are ADDED to the total at their offsets, as the pion
#       decay ones are assigned (:277);
#   S6  a shell's sum over its zones adds their 1e-99 floors (:792: log10 then gives -97.5 for a shell of three dark zones): floors are not summed;
#   S5  only the Doppler-shifted spectra are converted from photons per d(log E) to photons per bin (:166-170) while all three are divided by
#       the bin width at the end (:281-312): the inverse-Compton spectrum is converted too.
N_COS_BINS_DOPPLER = 180              # get_summed_emission.jl:117


def photon_shells(prob, num_upstream_shells: int, num_downstream_shells: int):
    """`set_photon_shells` (src/initializers.jl:305-398): shell boundaries between the shock and the free-escape boundaries, equal in
    log10|x / rg0| from 0.1 rg0 outwards; and the grid zone each boundary falls in (src/MonteCarloScattering.jl:392-401).
    -> (x_shell_endpoints_cm [n+1], n_shell_endpoints [n+1], 1-based zone numbers, 0 where no zone holds the boundary)."""
    P, rg0 = prob.params, prob.rg0
    nu, nd = int(num_upstream_shells), int(num_downstream_shells)
    if nu < 1 or nd < 1:
        raise ValueError("photon_shells: at least one shell on either side of the shock")
    ends = np.zeros(nu + nd + 1)
    w = (math.log10(abs(P.feb_upstream / rg0)) + 1) / nu
    for i in range(1, nu + 1):
        start = 0.0 if i == 1 else 10.0 ** (-1 + w * (i - 1))
        n = nu + 1 - i
        ends[n - 1] = -(10.0 ** (-1 + w * i)); ends[n] = -start
    use_prp = not P.feb_downstream > 0
    limit = P.x_grid_stop / rg0 if use_prp else P.feb_downstream / rg0
    w = (math.log10(limit) + 1) / nd
    for i in range(1, nd + 1):
        ends[nu + i - 1] = 0.0 if i == 1 else 10.0 ** (-1 + w * (i - 1))
        ends[nu + i] = 10.0 ** (-1 + w * i)
    ends *= rg0
    x = np.asarray(prob.x_grid_cm)
    zones = np.zeros(nu + nd + 1, dtype=np.int64)
    k = 0
    for i in range(1, P.n_grid + 1):
        if k <= nu + nd and x[i] <= ends[k] and x[i + 1] > ends[k]:
            zones[k] = i
            k += 1
    return ends, zones


def doppler_to_ism(flux, energy_MeV, gam_ef, beta_ef, bins_per_dec=PHOTON_BINS_PER_DEC):
    """get_summed_emission.jl:103-200 for one emission process: photons per d(log10 E) emitted isotropically in the plasma frame of every zone,
    re-binned by the Doppler-shifted energy of the central ray of 180 slices in cos(theta) (boundary 0 points upstream, towards the observer:
    E' = E gamma sqrt((1 - beta c_l)(1 - beta c_l+1))) and multiplied by gamma^3.  flux [n_zone][n] (floor 1e-99) -> photons per BIN [n_zone][n]."""
    flux = np.asarray(flux, dtype=np.float64); E = np.asarray(energy_MeV, dtype=np.float64)
    nz, n = flux.shape
    dlog = 1.0 / bins_per_dec                                                 # S1
    mid = np.empty(n)
    mid[:-1] = np.sqrt(E[:-1] * E[1:]); mid[-1] = mid[-2] * 10.0 ** dlog      # :151-155
    cos_e = np.linspace(-1.0, 1.0, N_COS_BINS_DOPPLER + 1)
    out = np.full((nz, n), 1.0e-99)
    for i in range(nz):
        num = np.where(flux[i] > 1.0e-90, flux[i] * dlog, flux[i]) / N_COS_BINS_DOPPLER        # :166-170, :179
        fac = gam_ef[i] * np.sqrt((1 - beta_ef[i] * cos_e[:-1]) * (1 - beta_ef[i] * cos_e[1:]))     # [l]
        Et = mid[None, :] * fac[:, None]                                                            # [l][j]
        # findnext(>=(E'), E, 2) - 1 with the 1-based E[1..n]: the first index m >= 2 with E[m] >= E' (S3: none -> dropped)
        m = np.searchsorted(E, Et, side="left")                  # 0-based index of the first E >= E'
        m = np.maximum(m, 1)                                     # (the search starts at the second entry)
        ok = (num[None, :] > 1.0e-99) & (m < n)
        np.add.at(out[i], (m - 1)[ok], np.broadcast_to(num[None, :], Et.shape)[ok])
        lit = out[i] > 1.0e-95
        out[i] = np.where(lit, out[i] * gam_ef[i] ** 3, 1.0e-99)                                    # :188-198 (S2)
    return out


@dataclasses.dataclass
class SummedEmission:
    """What `get_summed_emission` writes to photon_*_summed.dat, photon_tot_summed.dat and photon_tot.dat."""
    log_energy_MeV: np.ndarray        # [n_pts]  the common grid: 1e-13 MeV .. 1e12 MeV, 10 per decade
    per_process: dict                 # name -> (log10 E [n], log10 photons / (cm^2 s) per d(log10 E) [n_shell][n], -99 = none)
    per_shell: np.ndarray             # [n_shell][n_pts]  all processes, same units, -99 = none
    total: np.ndarray                 # [n_pts]


def summed_emission(prob, n_shell_endpoints, pion: PhotonPion = None, synch: PhotonSynch = None, ic: PhotonIC = None) -> SummedEmission:
    """The arithmetic of `get_summed_emission` (src/get_summed_emission.jl:37-413) on the per-zone photon fluxes of `photon_pion`,
    `photon_synch` and `photon_ic`: Doppler shift of the plasma-frame spectra, sum over the zones of every photon shell, one energy grid."""
    P = prob.params
    ends = np.asarray(n_shell_endpoints, dtype=np.int64)
    n_shells = len(ends) - 1
    gam_ef, beta_ef = np.asarray(prob.gam_ef)[1:P.n_grid + 1], np.asarray(prob.beta_ef)[1:P.n_grid + 1]
    dlog = 1.0 / PHOTON_BINS_PER_DEC
    n_pts = int((math.log10(PHOTON_E_MAX_MEV) - math.log10(PHOTON_E_MIN_MEV)) * PHOTON_BINS_PER_DEC)
    grid = math.log10(PHOTON_E_MIN_MEV) + dlog * np.arange(n_pts)
    total = np.full((n_shells, n_pts + 1), 1.0e-99)
    per = {}
    for name, ph, e_min, shift in (("pion", pion, PHOTON_PION_E_MIN_MEV, True), ("synch", synch, PHOTON_E_MIN_MEV, True), ("ic", ic, PHOTON_IC_E_MIN_MEV, False)):
        if ph is None:
            continue
        E = np.asarray(ph.energy_MeV)[:-1]                       # "the subroutines don't write out the final value of each spectrum" (:457-466)
        f = np.asarray(ph.photon_flux)[:, :-1]
        if shift:
            nb = doppler_to_ism(f, E, gam_ef, beta_ef)
        else:
            nb = np.where(f > 1.0e-90, f * dlog, f)              # (inverse Compton is computed in the ISM frame: number per bin only)
        sh = np.full((n_shells, len(E)), 1.0e-99)
        for n in range(n_shells):                                # sum_spectral_regions (:788-797)
            lo, hi = int(ends[n]), int(ends[n + 1]) - 1
            if lo >= 1 and hi >= lo:
                sh[n] = np.maximum(np.where(nb[lo - 1:hi] > 1.0e-95, nb[lo - 1:hi], 0.0).sum(axis=0), 1.0e-99)     # (S6)
        start = int(math.log10(e_min / PHOTON_E_MIN_MEV) * PHOTON_BINS_PER_DEC)      # :263-265
        total[:, start + 1:start + 1 + len(E)] += np.where(sh > 1.0e-99, sh, 0.0)    # :277, S4
        per[name] = (np.log10(E), np.where(sh > 1.0e-99, np.log10(np.maximum(sh, 1e-300) / dlog), -99.0))
    tot_sh = total[:, 1:n_pts + 1]                                # (the total's index n_start + j is 1-based)
    tot_all = tot_sh.sum(axis=0)
    back = lambda a: np.where(a > 1.0e-96, np.log10(np.maximum(a, 1e-300) / dlog), -99.0)       # :297-312
    return SummedEmission(grid, per, back(tot_sh), back(tot_all))


def _takes_download(backend) -> bool:
    import inspect
    return "download" in inspect.signature(backend.dndp_2d).parameters
