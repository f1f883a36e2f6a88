"""An INDEPENDENT scalar restatement of the hot path, in plain Python -- TEST INFRASTRUCTURE ONLY.

Why it exists: everything on the GPU is compared with oracle/mcs_oracle.cpp, and nothing the reference holds pins that
oracle (its test suite is Aqua hygiene only; the package cannot run here).  A transcription error made once in the oracle
would be invisible to every GPU <-> oracle test.  This module is a second restatement, written from the Julia files --
NOT from the C++ oracle -- with glibc math (Python's `math`) and its own Philox; tests/test_twin.py runs a few hundred
protons and the crafted electrons through both and requires equal discrete outcomes and momenta / tallies within 1e-11.

What it follows, function by function (file:line in /root/reference/src):
  particle_loop           particle_loop.jl:1-508          no_DSA_loop             particle_loop.jl:510-571
  radiation_loss          particle_loop.jl:578-592        downstream_test         particle_loop.jl:595-637
  perpendicular_momentum  particle_loop.jl:639-650        do_energy_transfer      particle_loop.jl:652-723
  scattering              scattering.jl:29-101            transform_p_PS / _PSP   transformers.jl:440-476, 523-607
  all_flux! / F_stream! / calculate_x_spec_spectra!       all_flux.jl:45-259
  prob_return / retro_time                                prob_return.jl:36-344
  particle_finish!        particle_finish.jl:46-107       get_psd_bin_*           get_psd_bins.jl:16-97
  tcut_track!             cuts.jl:149-162                 the call site           main_loops.jl:228-279

What it shares with the oracle is the SPECIFICATION that the Julia source cannot give:
  * random numbers: Julia's Xoshiro(seed) cannot be reproduced offline, so draws come from Philox4x32-10 keyed by the
    reference's iseed_mod (particle_loop.jl:35-40), counter = draw index / 2, the draw index even at every call site
    (the two draws of a scatter are one block; a single draw takes the first half of a block);
  * the documented deviations where the Julia as written throws or never ends: D1 (retro_time keeps the scattered pitch),
    D2, D3, D4, D5 (+ eps_target[0] read as 0), D6, D7 (retro cap), Q1 (escaping-flux scalars kept), A9 (thermal
    crossings binned on the fly: num_crossings + therm_sf + therm_pf, the two things the consumers compute from the list,
    thermo_calcs.jl:135-160, particle_counter.jl:425-445).
Index conventions: grid tables are Python lists indexed like the Julia OffsetVectors (0..n_grid+1); per-zone tallies are
stored at [i-1] for Julia zone i; PSD bins are 0-based in both; pcut / tcut / ion / iteration numbers are 1-based.
"""
import math

import numpy as np

MP = 1.67262192369e-24
ME = 9.1093837015e-28
C = 2.99792458e10
QCGS = 4.803204712570263e-10
SIGMA_T = 6.6524587321e-25
RAD_LOSS_FAC = (4.0 / 3.0) * C * SIGMA_T / (C * C * C * ME * ME * 8.0 * math.pi)     # constants.jl:30
E_REL_PT = 0.005            # parameters.jl:32
SPIKE_AWAY = 1000.0         # all_flux.jl:4, particle_finish.jl:5
HELIX_CAP = 10_000          # particle_loop.jl:162
RETRO_CAP = 10_000_000      # deviation D7
SIN_UPPER_LIMIT = math.nextafter(1.0, 0.0)      # prevfloat(1.0), scattering.jl:3
TWOPI = 2.0 * math.pi
PSD_MAX = 200
NA_C = 100


# ---------------------------------------------------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al. 2011), written from the paper's definition
def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        hi0, lo0 = p0 >> 32, p0 & 0xFFFFFFFF
        hi1, lo1 = p1 >> 32, p1 & 0xFFFFFFFF
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


class Stream:
    """Per-particle random stream: draw j = 64-bit word (j & 1) of block j >> 1, as a double in [0, 1) from its top 53 bits."""

    def __init__(self, key):
        self.key = (key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF)
        self.n = 0

    @staticmethod
    def _unit(lo, hi):
        return float(((hi << 32) | lo) >> 11) * 2.0 ** -53

    def rand(self):                     # a single draw: index n (even), n + 1 is skipped
        o = philox4x32_10((self.n >> 1, 0, 0, 0), self.key)
        self.n += 2
        return self._unit(o[0], o[1])

    def pair(self):                     # the two draws of a scatter: indices n, n + 1
        o = philox4x32_10((self.n >> 1, 0, 0, 0), self.key)
        self.n += 2
        return self._unit(o[0], o[1]), self._unit(o[2], o[3])


def mod2pi(x):
    r = math.fmod(x, TWOPI)
    if r < 0:
        r += TWOPI
    return r


def norm3(x, y, z):
    return math.sqrt(x * x + y * y + z * z)


def copysign(a, b):
    return math.copysign(a, b)


# ---------------------------------------------------------------------------------------------------------------------
class Twin:
    """One species of one iteration: the tallies accumulate over the pcuts run through `run_pcut`."""

    COUNTERS = ("STEPS_HELIX", "STEPS_RETRO", "HELIX_CAP", "PPERP_CLAMP", "PSP_CLAMP", "MOMBIN_CLAMP", "REASON0", "REASON1", "REASON2",
                "REASON3", "REASON4", "TCUT_OVERRUN", "RNG_DRAWS", "ZONE_FAIL", "RETRO_CAP")

    def __init__(self, prob, i_iter, i_ion, aa, zz, pmax_cutoff, density, ewf, retro_cap=RETRO_CAP):
        P = prob.params
        self.P = P
        self.i_iter, self.i_ion = i_iter, i_ion
        self.aa, self.zz = aa, zz * QCGS                      # "zz already has units of charge" (particle_loop.jl:72)
        self.m = aa * MP
        self.mc = self.m * C
        self.pmax_cutoff, self.density, self.ewf = pmax_cutoff, density, ewf
        self.retro_cap = retro_cap
        f = lambda a: [float(v) for v in a]
        self.x_grid, self.ux_g, self.uz_g, self.ut_g = f(prob.x_grid_cm), f(prob.ux), f(prob.uz), f(prob.utot)
        self.gsf_g, self.gef_g, self.bt_g, self.th_g = f(prob.gam_sf), f(prob.gam_ef), f(prob.btot), f(prob.theta)
        self.pcuts, self.tcuts, self.x_spec = f(prob.pcuts), f(prob.tcuts), f(prob.x_spec)
        self.inj_fracs, self.eps_target = f(prob.inj_fracs), f(prob.eps_target)      # eps_target[i-1] = Julia eps_target[i]
        self.n_grid = int(P.n_grid)
        ng, nm, nt, pm = self.n_grid, P.num_psd_mom_bins + 2, P.num_psd_tht_bins + 2, PSD_MAX + 1
        z = np.zeros
        self.T = dict(psd=z((ng, nt, nm)), therm_sf=z((ng, nt, nm)), therm_pf=z((ng, nt, nm)), esc_psd_up=z((pm, pm)),
                      esc_psd_down=z((pm, pm)), pxx_flux=z(ng), pxz_flux=z(ng), energy_flux=z(ng), esc_flux=z(P.n_ions),
                      px_esc_feb=z((P.n_itrs, P.n_ions)), energy_esc_feb=z((P.n_itrs, P.n_ions)), esc_energy_eff=z((P.n_ions, pm)),
                      esc_num_eff=z((P.n_ions, pm)), weight_coupled=z((P.n_ions, NA_C)), spectra_coupled=z((P.n_ions, NA_C, pm)),
                      spectra_sf=z((ng, pm)), spectra_pf=z((ng, pm)), energy_transfer_pool=z(ng), energy_recv_pool=z(ng), scalars=z(4))
        self.num_crossings = np.zeros(ng, dtype=np.int64)
        self.cnt = {k: 0 for k in self.COUNTERS}

    # -- get_psd_bins.jl:16-39
    def bin_momentum(self, ptot_sk):
        P = self.P
        if ptot_sk < P.psd_mom_min:
            b = 0
        else:
            b = int(math.log10(ptot_sk / P.psd_mom_min) * P.psd_bins_per_dec_mom) + 1      # trunc(Int, ...) of a non-negative number
        if b > P.num_psd_mom_bins:
            self.cnt["MOMBIN_CLAMP"] += 1
            b = P.num_psd_mom_bins
        return b

    # -- get_psd_bins.jl:73-97
    def bin_angle(self, px_sk, ptot_sk):
        P = self.P
        if ptot_sk == 0:
            return 0
        p_cos = -px_sk / ptot_sk
        if p_cos < P.psd_cos_fine:
            b = P.num_psd_tht_bins - int((p_cos + 1) / P.psd_dcos)
        else:
            th = math.acos(p_cos)
            b = 0 if th < P.psd_tht_min else int(math.log10(th / P.psd_tht_min) * P.psd_bins_per_dec_tht) + 1
        return min(b, P.num_psd_tht_bins)

    # -- transformers.jl:440-476
    def transform_p_PS(self, pb_pf, p_perp, gam_pf, phi, ux, gsf, bcos, bsin):
        m = self.aa * MP
        mc = m * C
        phi_p = phi + math.pi / 2
        p_p_cos = p_perp * math.cos(phi_p)
        fx = pb_pf * bcos - p_p_cos * bsin
        fy = p_perp * math.sin(phi_p)
        fz = pb_pf * bsin + p_p_cos * bcos
        dpx = (gsf - 1) * fx + gsf * gam_pf * m * ux
        px, py, pz = fx + dpx, fy, fz
        ptot_sk = norm3(px, py, pz)
        gam_sk = math.hypot(ptot_sk / mc, 1)
        return ptot_sk, (px, py, pz), gam_sk

    # -- transformers.jl:523-607
    def transform_p_PSP(self, pb_pf, p_perp, gam_pf, phi, old, new):
        ux_o, uz_o, ut_o, gsf_o, bcos_o, bsin_o = old
        ux, uz, ut, gsf, bcos, bsin = new
        phi_p = phi + math.pi / 2
        m = self.aa * MP
        mc = m * C
        p_p_cos = p_perp * math.cos(phi_p)
        fx = pb_pf * bcos_o - p_p_cos * bsin_o
        fy = p_perp * math.sin(phi_p)
        fz = pb_pf * bsin_o + p_p_cos * bcos_o
        kx = ((gsf_o - 1) * (ux_o / ut_o) ** 2 + 1) * fx + (gsf_o - 1) * (ux_o * uz_o / ut_o ** 2) * fz + gsf_o * gam_pf * m * ux_o
        ky = fy
        kz = (gsf_o - 1) * (ux_o * uz_o / ut_o ** 2) * fx + ((gsf_o - 1) * (uz_o / ut_o) ** 2 + 1) * fz + gsf_o * gam_pf * m * uz_o
        ptot_sk = norm3(kx, ky, kz)
        pb_sk = kx * bcos + kz * bsin
        if ptot_sk < abs(pb_sk):
            self.cnt["PSP_CLAMP"] += 1          # (@warn in the reference; the clamped shock-frame pair is not used by the caller)
        gam_sk = math.hypot(ptot_sk / mc, 1)
        gx = ((gsf - 1) * (ux / ut) ** 2 + 1) * kx + (gsf - 1) * (ux * uz / ut ** 2) * kz - gsf * gam_sk * m * ux
        gy = ky
        gz = (gsf - 1) * (ux * uz / ut ** 2) * kx + ((gsf - 1) * (uz / ut) ** 2 + 1) * kz - gsf * gam_sk * m * uz
        ptot_pf = norm3(gx, gy, gz)
        pb = gx * bcos + gz * bsin
        if ptot_pf < abs(pb):
            pperp = 1.0e-6 * ptot_pf
            pb = copysign(math.sqrt(ptot_pf ** 2 - pperp ** 2), pb)
            self.cnt["PSP_CLAMP"] += 1
        else:
            pperp = math.sqrt(ptot_pf ** 2 - pb ** 2)
        gam = math.hypot(ptot_pf / mc, 1)
        phi_p = math.atan2(gy, -gx * bsin + gz * bcos)
        return ptot_pf, pb, pperp, gam, phi_p - math.pi / 2

    # -- particle_loop.jl:639-650 (the adjusted pb is not returned: quirk Q6)
    def perpendicular_momentum(self, ptot, pb):
        if ptot < abs(pb):
            self.cnt["PPERP_CLAMP"] += 1
            return 1.0e-6 * ptot
        return math.sqrt(ptot ** 2 - pb ** 2)

    # -- particle_loop.jl:578-592
    @staticmethod
    def radiation_loss(B2, p, dt):
        d = RAD_LOSS_FAC * B2 * p * dt
        if d > 1.0e-2:
            p /= 1 + d
        else:
            p *= 1 - d
        return p

    # -- cuts.jl:149-162
    def tcut_track(self, tcut_curr, weight, ptot_pf):
        ion = self.i_ion - 1
        self.T["weight_coupled"][ion, tcut_curr - 1] += weight
        self.T["spectra_coupled"][ion, tcut_curr - 1, self.bin_momentum(ptot_pf)] += weight

    # -- scattering.jl:29-101
    def scattering(self, rng, gyro_denom, ptot_pf, gam_pf, xn_per, pb_pf, p_perp, phi):
        P = self.P
        mc = self.aa * MP * C
        if self.aa < 1 and ptot_pf < P.pe_crit:
            grt = P.pe_crit * C * gyro_denom
            gyro_period = TWOPI * P.game_crit * mc * gyro_denom
        else:
            grt = ptot_pf * C * gyro_denom
            gyro_period = TWOPI * gam_pf * mc * gyro_denom
        vp_tg = TWOPI * grt
        lam = P.eta_mfp * grt
        cos_max = math.cos(math.sqrt(6 * vp_tg / (xn_per * lam)))
        cos_old = pb_pf / ptot_pf
        sin_old = p_perp / ptot_pf
        u1, u2 = rng.pair()
        cos_d = 1 - u1 * (1 - cos_max)
        sin_d = math.sqrt(1 - cos_d ** 2)
        phi_scat = u2 * TWOPI - math.pi
        cos_new = cos_old * cos_d + sin_old * sin_d * math.cos(phi_scat)
        sin_new = math.sqrt(max(1 - cos_new ** 2, 0.0))           # D3
        pb_pf = ptot_pf * cos_new
        p_perp = ptot_pf * sin_new
        phi_p_old = phi + math.pi / 2
        phi_p_new = phi_p_old
        if sin_new != 0:
            s = math.sin(phi_scat) * sin_d / sin_new            # get_sine_adjustment, scattering.jl:93-101
            if abs(s) > SIN_UPPER_LIMIT:
                s = copysign(SIN_UPPER_LIMIT, s)
            phi_p_new += math.asin(s)
        return gyro_period, pb_pf, p_perp, phi_p_new - math.pi / 2

    # -- all_flux.jl:45-259; returns (i_grid, i_grid_old) or None (D6: the zone search failed)
    def all_flux(self, pb_pf, p_perp, ptot_pf, gam_pf, phi, weight, i_grid, ux, gsf, bcos, bsin, x, x_old, inj):
        P = self.P
        xg = self.x_grid
        ne = self.n_grid + 2
        i_grid_old = i_grid
        found = None
        if x > x_old:                                          # findnext(>(x), x_grid, i_grid + 1) - 1
            for j in range(i_grid + 1, ne):
                if xg[j] > x:
                    found = j - 1
                    break
        else:                                                  # findprev(<=(x), x_grid, i_grid)
            for j in range(i_grid, -1, -1):
                if xg[j] <= x:
                    found = j
                    break
        if found is None:
            return None
        i_grid = found
        n_xspec = len(self.x_spec)
        if i_grid == i_grid_old and i_grid > P.i_grid_feb and n_xspec == 0:
            return i_grid, i_grid_old
        ptot_sk, (px, py, pz), gam_sk = self.transform_p_PS(pb_pf, p_perp, gam_pf, phi, ux, gsf, bcos, bsin)
        m = self.aa * MP
        if ptot_sk > abs(px * SPIKE_AWAY):
            pt_o_px_sk = SPIKE_AWAY
            abs_inv_vx = abs(SPIKE_AWAY / ux)
        else:
            pt_o_px_sk = ptot_sk / px
            abs_inv_vx = abs(gam_sk * self.aa * MP / px)
        if (gam_sk - 1) > E_REL_PT:
            e_add = (gam_sk - 1) * m * C ** 2 * weight
        else:
            e_add = ptot_sk ** 2 / (2 * m) * weight
        if n_xspec > 0:                                        # calculate_x_spec_spectra!, all_flux.jl:164-190
            r = abs(ptot_pf / pb_pf) if pb_pf != 0 else math.inf
            pt_o_px_pf = min(r, SPIKE_AWAY)
            i_pt, i_pt_pf = self.bin_momentum(ptot_sk), self.bin_momentum(ptot_pf)
            for i, xs in enumerate(self.x_spec):
                if (x_old < xs and x >= xs) or (x <= xs and x_old > xs):
                    self.T["spectra_sf"][i, i_pt] += weight * pt_o_px_sk
                    Fw = abs(pb_pf / px) * (gam_sk / gam_pf)
                    self.T["spectra_pf"][i, i_pt_pf] += weight * pt_o_px_pf * Fw
        if x > x_old:
            i_range, inj_check, sign_fac = range(i_grid_old + 1, i_grid + 1), False, 1
        else:
            i_range, inj_check, sign_fac = range(i_grid_old, i_grid, -1), True, -1
        # F_stream!, all_flux.jl:197-259
        if inj:
            i_pt, jth = self.bin_momentum(ptot_sk), self.bin_angle(px, ptot_sk)
        have_sf = False
        g0u0 = P.gam0 * P.u0
        for i in i_range:
            if inj_check and inj and i <= P.i_grid_feb:
                continue
            self.T["pxx_flux"][i - 1] += sign_fac * px * weight * g0u0
            self.T["pxz_flux"][i - 1] += abs(pz) * weight * g0u0
            self.T["energy_flux"][i - 1] += sign_fac * e_add * g0u0
            if inj:
                self.T["psd"][i - 1, jth, i_pt] += weight * abs_inv_vx
            else:
                # A9: the crossing (i, px_sk, ptot_sk, weight |1/vx|) binned as its two consumers bin it
                tw = weight * abs_inv_vx
                if P.track_thermal:
                    if not have_sf:
                        k_sf, j_sf, have_sf = self.bin_momentum(ptot_sk), self.bin_angle(px, ptot_sk), True
                    self.T["therm_sf"][i - 1, j_sf, k_sf] += tw                 # particle_counter.jl:433-444
                    gam, beta = self.gsf_g[i], self.ux_g[i] / C                # thermo_calcs.jl:135-160
                    E0 = self.m * C ** 2
                    etot = math.hypot(ptot_sk * C, E0)
                    px_X = gam * (px - beta * etot / C)
                    pt_X = math.sqrt((ptot_sk ** 2 - px ** 2) + px_X ** 2)
                    if abs(px_X) > pt_X:
                        px_X = copysign(pt_X, px_X)
                    self.T["therm_pf"][i - 1, self.bin_angle(px_X, pt_X), self.bin_momentum(pt_X)] += tw
                self.num_crossings[i - 1] += 1
        if inj and x < P.feb_upstream and x_old >= P.feb_upstream:      # Q1: kept (the reference drops these by value)
            self.T["scalars"][3] += e_add * g0u0
            self.T["scalars"][2] -= px * weight * g0u0
        return i_grid, i_grid_old

    # -- particle_loop.jl:652-723
    def do_energy_transfer(self, i_grid, i_grid_old, ptot_pf, pb_pf, p_perp, gam_pf, weight):
        P = self.P
        i_start, i_stop = i_grid_old, min(i_grid, P.i_shock)
        zones = [i for i in range(i_start + 1, i_stop + 1) if 1 <= i <= self.n_grid]
        eps = lambda i: self.eps_target[i - 1] if i >= 1 else 0.0        # (eps_target[0] would be a BoundsError: read as 0)
        E0 = self.m * C ** 2
        scale = False
        gam_f = gam_pf
        recv = self.T["energy_recv_pool"]
        if self.aa >= 1 and zones and max(eps(i) for i in zones) > 0:                   # D5: an empty range transfers nothing
            gam_i = math.hypot(1, ptot_pf / self.mc)
            gam_f = 1 + (gam_i - 1) * (1 - eps(i_stop)) / (1 - eps(i_start))
            n_split = sum(1 for i in zones if eps(i) > 0)
            inc = (gam_i - gam_f) * E0 * weight / n_split
            for i in zones:
                if eps(i) > 0:
                    self.T["energy_transfer_pool"][i - 1] += inc
            scale = True
        elif zones and max(recv[i - 1] for i in zones) > 0:
            e_tr = 0.0
            for i in zones:
                e_tr += recv[i - 1]
            e_tr *= self.ewf
            gam_i = math.hypot(1, ptot_pf / self.mc)
            gam_f = gam_i + e_tr / E0
            scale = True
        if scale:
            ptot_f = self.mc * math.sqrt(gam_f ** 2 - 1)
            sf = ptot_f / ptot_pf
            pb_pf *= sf
            p_perp *= sf
            ptot_pf = ptot_f
            gam_pf = gam_f
        return pb_pf, p_perp, ptot_pf, gam_pf

    # -- prob_return.jl:217-344, with D1 (the scattered pitch is kept) and D7 (cap)
    def retro_time(self, rng, gyro_denom, prp, ptot_pf, pb_pf, p_perp, gam_pf, acctime, weight, tcut_curr):
        P = self.P
        ng = self.n_grid
        aa = self.aa
        xn_per = 10.0
        phi_step = TWOPI / xn_per
        t_step_fac = TWOPI * aa * MP * C * gyro_denom / xn_per
        ux = -self.ux_g[ng]
        gsf, gef = self.gsf_g[ng], self.gef_g[ng]
        B = self.bt_g[ng]
        if P.use_custom_epsB:
            B *= math.sqrt(P.x_grid_stop / prp)
        bcos, bsin = math.cos(self.th_g[ng]), math.sin(self.th_g[ng])
        B_CMB_loc = P.B_CMBz * gef
        B2 = B ** 2 + B_CMB_loc ** 2
        lose_pt = capped = False
        x = prp
        phi = rng.rand() * TWOPI
        n_steps = 0
        n_tcuts = len(self.tcuts)
        while True:
            n_steps += 1
            x_old, phi_old = x, phi
            if P.use_custom_epsB:
                B = self.bt_g[ng] * math.sqrt(P.x_grid_stop / x)
                B2 = B ** 2 + B_CMB_loc ** 2
                gyro_denom = 1 / (self.zz * B)
            gyro_rad = p_perp * C * gyro_denom
            phi = mod2pi(phi_old + phi_step)
            t_step = t_step_fac * gam_pf
            x_move = pb_pf * t_step_fac / (aa * MP)
            x = x_old + gsf * (x_move * bcos - gyro_rad * bsin * (math.cos(phi) - math.cos(phi_old)) + ux * t_step)
            acctime += t_step * gef
            if P.do_tcuts:
                if tcut_curr > n_tcuts:
                    self.cnt["TCUT_OVERRUN"] += 1                       # D4
                elif acctime >= self.tcuts[tcut_curr - 1]:
                    self.tcut_track(tcut_curr, weight, ptot_pf)
                    tcut_curr += 1
            phi = TWOPI * rng.rand()
            ptot_old = ptot_pf
            pb_pf = (2 * rng.rand() - 1) * ptot_pf
            p_perp = math.sqrt(max(ptot_pf ** 2 - pb_pf ** 2, 0.0))
            cos_new, sin_new = pb_pf / ptot_old, p_perp / ptot_old        # D1: the pitch the large-angle scattering produced
            if P.do_rad_losses and aa < 1:
                ptot_pf = self.radiation_loss(B2, ptot_pf, t_step)
            if ptot_pf <= 0:
                ptot_pf, gam_pf, lose_pt = 1.0e-99, 1.0, True
                break
            pb_pf = ptot_pf * cos_new
            p_perp = ptot_pf * sin_new
            gam_pf = math.hypot(1, ptot_pf / self.mc)
            if x < prp:
                break
            if n_steps >= self.retro_cap:                               # D7
                capped = True
                self.cnt["RETRO_CAP"] += 1
                break
        return dict(lose_pt=lose_pt, capped=capped, phi=phi, tcut_curr=tcut_curr, ptot_pf=ptot_pf, pb_pf=pb_pf, p_perp=p_perp,
                    gam_pf=gam_pf, gyro_denom=gyro_denom, acctime=acctime, n_steps=n_steps)

    # -- the loop at main_loops.jl:228-279 for one particle: particle_loop (particle_loop.jl:1-508) + particle_finish!
    def run_particle(self, i_pcut, gi, st):
        """gi: global 0-based particle index (i_prt = gi + 1).  st: dict of the 12 fields of src/particle_loop.jl:48-59.
        Returns (final, saved): final = dict(reason, helix, retro, ptot, x); saved = dict of the 12 saved fields or None."""
        P = self.P
        aa, m, mc, zz = self.aa, self.m, self.mc, self.zz
        n_pcuts = len(self.pcuts)
        iseed = ((self.i_iter - 1) * P.n_pts_max * n_pcuts * P.n_ions + (self.i_ion - 1) * P.n_pts_max * n_pcuts
                 + (i_pcut - 1) * P.n_pts_max + (gi + 1))
        rng = Stream(iseed)
        pcut = self.pcuts[i_pcut - 1]
        pcut_prev = self.pcuts[i_pcut - 2] if i_pcut > 1 else 0.0
        n_tcuts = len(self.tcuts)
        helix = 0
        n_retro = 0
        weight, ptot_pf, pb_pf = st["weight"], st["ptot_pf"], st["pb_pf"]
        i_grid = int(st["grid"]); i_grid_old = i_grid
        l_down, inj = bool(st["downstream"]), bool(st["inj"])
        xn_per, prp, acctime, phi, tcut_curr = st["xn_per"], st["prp_x_cm"], st["acctime_sec"], st["phi_rad"], int(st["tcut"])
        x = st["x_PT_cm"]
        gam_pf = math.hypot(1, ptot_pf / mc)
        gyro_denom = 1 / (zz * self.bt_g[i_grid])
        if P.use_custom_epsB and x > P.x_grid_stop:
            gyro_denom *= math.sqrt(x / P.x_grid_stop)
        gyro_rad_tot = ptot_pf * C * gyro_denom
        gyro_period = TWOPI * gam_pf * m * C * gyro_denom
        ux, uz, ut, gsf, gef = self.ux_g[i_grid], self.uz_g[i_grid], self.ut_g[i_grid], self.gsf_g[i_grid], self.gef_g[i_grid]
        bmag = self.bt_g[i_grid]
        bsin, bcos = math.sin(self.th_g[i_grid]), math.cos(self.th_g[i_grid])
        i_return, i_reason, lose_pt, capped = -1, 0, False, False
        t_step = 0.0
        p_perp = self.perpendicular_momentum(ptot_pf, pb_pf)
        gyro_rad = p_perp * C * gyro_denom
        x_old = 0.0
        saved = None
        while True:
            helix += 1
            if helix > HELIX_CAP:                                   # quirk Q5
                i_reason = 1
                self.cnt["HELIX_CAP"] += 1
                break
            if i_return == 1:                                       # Code Block 1
                p_perp = self.perpendicular_momentum(ptot_pf, pb_pf)
                gyro_rad = p_perp * C * gyro_denom
            else:                                                   # Code Block 3
                old = (ux, uz, ut, gsf, bcos, bsin)
                ux, uz, ut, gsf, gef = self.ux_g[i_grid], self.uz_g[i_grid], self.ut_g[i_grid], self.gsf_g[i_grid], self.gef_g[i_grid]
                bmag = self.bt_g[i_grid]
                bsin, bcos = math.sin(self.th_g[i_grid]), math.cos(self.th_g[i_grid])
                if P.use_custom_epsB and x > P.x_grid_stop:
                    bmag = self.bt_g[self.n_grid] * math.sqrt(P.x_grid_stop / x)
                gyro_denom = 1 / (zz * bmag)
                if ux != old[0]:
                    ptot_pf, pb_pf, p_perp, gam_pf, phi = self.transform_p_PSP(pb_pf, p_perp, gam_pf, phi, old, (ux, uz, ut, gsf, bcos, bsin))
                    gyro_rad = p_perp * C * gyro_denom
                    gyro_rad_tot = ptot_pf * C * gyro_denom
                if P.energy_transfer_frac > 0 and not inj and x_old <= 0 and i_grid_old != i_grid:
                    pb_pf, p_perp, ptot_pf, gam_pf = self.do_energy_transfer(i_grid, i_grid_old, ptot_pf, pb_pf, p_perp, gam_pf, weight)
                if P.dont_scatter and x > 10 * gyro_rad:
                    i_return, i_reason = 0, 1
                    break
                if ptot_pf > self.pmax_cutoff:
                    ptot_sk, _, _ = self.transform_p_PS(pb_pf, p_perp, gam_pf, phi, ux, gsf, bcos, bsin)
                    if ptot_sk > self.pmax_cutoff:
                        i_reason = 2
                        break
                if inj and x < P.feb_upstream:
                    i_reason = 2
                    break
                if P.age_max > 0 and acctime > P.age_max:
                    i_reason = 3
                    break
                if P.do_rad_losses and aa < 1:
                    ptot_old = ptot_pf
                    B_CMB_loc = P.B_CMBz * gef
                    ptot_pf = self.radiation_loss(bmag ** 2 + B_CMB_loc ** 2, ptot_pf, t_step)
                    if ptot_pf <= 0:
                        ptot_pf = pb_pf = p_perp = 1.0e-99
                        gam_pf = 1.0
                        i_reason = 4
                        break
                    gam_pf = math.hypot(ptot_pf / mc, 1)
                    pb_pf *= ptot_pf / ptot_old
                    p_perp *= ptot_pf / ptot_old
                    gyro_rad_tot = ptot_pf * C * gyro_denom
                    gyro_rad = p_perp * C * gyro_denom
                if not P.dont_scatter:
                    gyro_period, pb_pf, p_perp, phi = self.scattering(rng, gyro_denom, ptot_pf, gam_pf, xn_per, pb_pf, p_perp, phi)
                if l_down:
                    acctime += t_step * gef
                    if P.do_tcuts:
                        if tcut_curr > n_tcuts:
                            self.cnt["TCUT_OVERRUN"] += 1               # D4 (tcuts[tcut_curr] would be a BoundsError)
                        elif acctime >= self.tcuts[tcut_curr - 1]:
                            self.tcut_track(tcut_curr, weight, ptot_pf)
                            tcut_curr += 1
                    if ptot_pf > pcut:
                        saved = dict(weight=weight, ptot_pf=ptot_pf, pb_pf=pb_pf, x_PT_cm=x, grid=i_grid, downstream=l_down, inj=inj,
                                     xn_per=xn_per, prp_x_cm=prp if x < prp else x * 1.1, acctime_sec=acctime, phi_rad=phi, tcut=tcut_curr)
                        break
                xn_per = P.xn_per_coarse if x > gyro_rad_tot else P.xn_per_fine
            # Code Block 2
            x_old = x
            phi_old = phi
            t_step = gyro_period / xn_per
            # no_DSA_loop, particle_loop.jl:510-571
            while True:
                phi = mod2pi(phi + TWOPI / xn_per)
                x_move = pb_pf * t_step / (gam_pf * m)
                dx = gsf * (x_move * bcos - gyro_rad * bsin * (math.cos(phi) - math.cos(phi_old)) + ux * t_step)
                x = x_old + dx
                inj_frac = self.inj_fracs[self.i_ion - 1]
                if x <= 0 and x_old > 0 and not inj and (P.dont_DSA or inj_frac < 1):
                    if P.dont_DSA or rng.rand() > inj_frac:
                        if pb_pf < 0:
                            pb_pf = -pb_pf
                        else:
                            phi = rng.rand() * TWOPI
                    else:
                        break
                else:
                    break
            if x_old < 0 and x >= 0:
                l_down = True
                L_diff = P.eta_mfp / 3 * gyro_rad_tot * ptot_pf / (m * gam_pf * P.u2)
                prp = max(prp, L_diff)
            if l_down and x < 0:
                inj = True
            r = self.all_flux(pb_pf, p_perp, ptot_pf, gam_pf, phi, weight, i_grid, ux, gsf, bcos, bsin, x, x_old, inj)
            if r is None:                                             # D6
                self.cnt["ZONE_FAIL"] += 1
                i_reason = 3
                break
            i_grid, i_grid_old = r
            # downstream_test, particle_loop.jl:595-637
            do_prob_ret = True
            if P.feb_downstream > 0 and x > P.feb_downstream:
                i_return, do_prob_ret = 0, False
            elif x > 1.1 * prp:
                if aa < 1 and ptot_pf < P.pe_crit:
                    gyro_fac = P.pe_crit * C * gyro_denom
                    v_fac = gyro_fac * P.pe_crit / (m * P.game_crit * P.u2)
                else:
                    v_fac = gyro_rad_tot * ptot_pf / (m * gam_pf * P.u2)
                L_diff = P.eta_mfp / 3 * v_fac
                if x > 6.91 * L_diff:
                    i_return, do_prob_ret = 0, False
            if do_prob_ret:
                # prob_return, prob_return.jl:36-173
                i_return = 2
                lose_pt = False
                if x < P.x_grid_stop:
                    pass
                elif x_old < P.x_grid_stop <= x:
                    gyro_tmp = math.sqrt(P.x_grid_stop / x) if (P.use_custom_epsB and x > P.x_grid_stop) else 1.0
                    grt = ptot_pf * C * gyro_tmp / (QCGS * P.bmag2)
                    L_diff = P.eta_mfp / 3 * grt * ptot_pf / (aa * MP * gam_pf * P.u2)
                    prp = x + 3 * L_diff
                elif x_old < prp and x >= prp:
                    vt = ptot_pf / (gam_pf * aa * MP)
                    prob_ret = ((vt - P.u2) / (vt + P.u2)) ** 2
                    if vt < P.u2 or rng.rand() > prob_ret:
                        i_return = 0
                    else:
                        i_return = 1
                        if not P.do_retro:
                            raise RuntimeError("Code not set up for analytical PRP calculations.")     # prob_return.jl:134
                        o = self.retro_time(rng, gyro_denom, prp, ptot_pf, pb_pf, p_perp, gam_pf, acctime, weight, tcut_curr)
                        lose_pt, capped, phi, tcut_curr = o["lose_pt"], o["capped"], o["phi"], o["tcut_curr"]
                        ptot_pf, pb_pf, p_perp, gam_pf = o["ptot_pf"], o["pb_pf"], o["p_perp"], o["gam_pf"]
                        gyro_denom, acctime = o["gyro_denom"], o["acctime"]
                        n_retro += o["n_steps"]
                        if lose_pt or capped:
                            i_return = 0
                        x = prp
                else:
                    if aa < 1 and ptot_pf < pcut_prev and helix % 1000 == 0:
                        grt = ptot_pf * C * gyro_denom
                        L_diff = P.eta_mfp / 3 * grt * ptot_pf / (aa * MP * gam_pf * P.u2)
                        if x > 2.0e3 * L_diff:
                            prp = 0.8 * x
                        else:
                            prp = min(prp, P.x_grid_stop + L_diff * (pcut_prev / ptot_pf) ** 5)
            if i_return == 0:
                if capped:                                            # D7: ends like an aged-out particle
                    i_reason = 3
                    break
                vel = ptot_pf / m
                if (gam_pf - 1) >= E_REL_PT:
                    vel /= gam_pf
                self.T["scalars"][0] += ptot_pf / 3 * vel * weight * self.density
                self.T["scalars"][1] += (gam_pf - 1) * m * C ** 2 * weight * self.density
                i_reason = 4 if lose_pt else 1
                break
        self.cnt["STEPS_HELIX"] += min(helix, HELIX_CAP)
        self.cnt["STEPS_RETRO"] += n_retro
        self.cnt["RNG_DRAWS"] += rng.n
        if saved is None:
            self.particle_finish(i_reason, pb_pf, p_perp, gam_pf, phi, ux, gsf, bcos, bsin, weight)
            self.cnt[f"REASON{i_reason}"] += 1
        else:
            self.cnt["REASON0"] += 1
        return dict(reason=0 if saved is not None else i_reason, helix=helix, retro=n_retro, ptot=ptot_pf, x=x), saved

    # -- particle_finish.jl:46-107 (D2: (gamma - 1) >= E_rel_pt)
    def particle_finish(self, i_reason, pb_pf, p_perp, gam_pf, phi, ux, gsf, bcos, bsin, weight):
        m = self.aa * MP
        E0 = m * C ** 2
        ion, it = self.i_ion - 1, self.i_iter - 1
        ptot_sk, (px, py, pz), gam_sk = self.transform_p_PS(pb_pf, p_perp, gam_pf, phi, ux, gsf, bcos, bsin)
        ip = self.bin_momentum(ptot_sk)
        jth = self.bin_angle(px, ptot_sk)
        if ptot_sk > abs(SPIKE_AWAY * px):
            wf = gam_sk * m * SPIKE_AWAY / ptot_sk
        else:
            wf = gam_sk * (m / abs(px))
        if i_reason == 1:
            self.T["esc_psd_down"][jth, ip] += weight * wf
        elif i_reason == 2:
            self.T["esc_flux"][ion] += weight
            self.T["esc_psd_up"][jth, ip] += weight * wf
            rel = (gam_sk - 1) >= E_REL_PT
            E_kin = (gam_sk - 1) * E0 if rel else ptot_sk ** 2 / (2 * m)
            e_add = E_kin * weight
            self.T["px_esc_feb"][it, ion] += abs(px) * weight
            self.T["energy_esc_feb"][it, ion] += e_add
            self.T["esc_energy_eff"][ion, ip] += e_add
            self.T["esc_num_eff"][ion, ip] += weight
        elif i_reason in (3, 4):
            pass
        else:
            raise RuntimeError(f"Unknown i_reason passed: {i_reason}. Can only handle 1-4")

    # -- one pcut over a population given as a dict of arrays (the 12 fields), global indices 0..n-1
    def run_pcut(self, i_pcut, pop):
        n = len(pop["weight"])
        finals = dict(reason=np.zeros(n, np.int32), helix=np.zeros(n, np.int32), retro=np.zeros(n, np.int32), ptot=np.zeros(n), x=np.zeros(n))
        saved_rows = []
        for k in range(n):
            st = {f: pop[f][k] for f in pop}
            fin, sv = self.run_particle(i_pcut, k, st)
            for f in finals:
                finals[f][k] = fin[f]
            saved_rows.append(sv)
        return finals, saved_rows


def split_population(saved_rows, i_mult):
    """new_pcut (cuts.jl:34-98): the saved particles in order, each i_mult times with weight / i_mult."""
    rows = [r for r in saved_rows if r is not None]
    fields = ("weight", "ptot_pf", "pb_pf", "x_PT_cm", "xn_per", "prp_x_cm", "acctime_sec", "phi_rad", "grid", "tcut", "downstream", "inj")
    out = {f: [] for f in fields}
    for r in rows:
        for _ in range(i_mult):
            for f in fields:
                out[f].append(r[f] / i_mult if f == "weight" else r[f])
    return {f: np.array(v) for f, v in out.items()}
