"""ctypes wrapper of the CPU oracle (oracle/mcs_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
`OracleBackend` implements the same backend protocol as the HIP backend so the
host driver (montecarloscattering.jl_amd/driver.py) can be run against either.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False) -> None:
    """Compile both oracle variants with the committed Makefile (gcc)."""
    if force:
        subprocess.check_call(["make", "-C", HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", HERE], stdout=subprocess.DEVNULL)


def load(math: str = "det", capi=None):
    path = os.path.join(HERE, f"libmcs_oracle_{math}.so")
    if not os.path.exists(path):
        build()
    lib = ct.CDLL(path)
    vp, i32, i64, dbl = ct.c_void_p, ct.c_int, ct.c_int64, ct.c_double
    dp = ct.POINTER(ct.c_double)
    i64p = ct.POINTER(ct.c_int64)
    u8p = ct.POINTER(ct.c_uint8)
    i32p = ct.POINTER(ct.c_int32)
    u32p = ct.POINTER(ct.c_uint32)
    soa_p = ct.POINTER(capi.McsSoa) if capi is not None else vp
    par_p = ct.POINTER(capi.McsParams) if capi is not None else vp
    sig = {
        "orc_last_error": (ct.c_char_p, []),
        "orc_math_mode": (ct.c_char_p, []),
        "orc_create": (vp, [par_p]),
        "orc_destroy": (None, [vp]),
        "orc_set_grid": (i32, [vp, i32] + [dp] * 9),
        "orc_set_cuts": (i32, [vp, i32, dp, i32, dp, i32, dp, dp, dp]),
        "orc_begin_iteration": (i32, [vp, i32]),
        "orc_begin_species": (i32, [vp, i32, i32, dbl, dbl, dbl, dbl, dbl]),
        "orc_set_fluxes": (i32, [vp, dp, dp, dp]),
        "orc_run_pcut": (i32, [vp, i32, i64, i64, i64, i64p, soa_p, soa_p, u8p, i64p, i32]),
        "orc_run_pcut_f32": (i32, [vp, i32, i64, i64, i64, i64p, soa_p, soa_p, u8p, i64p]),
        "orc_set_retro_cap": (i32, [vp, i64]),
        "orc_finals": (i32, [vp, i64, i32p, i32p, i32p, dp, dp]),
        "orc_read_tallies": (i32, [vp, dp, i64p]),
        "orc_write_tallies": (i32, [vp, dp, i64p]),
        "orc_new_pcut": (i64, [i64, i64, u8p, soa_p, soa_p]),
        "orc_init_pop": (i32, [vp, i64, i64, i64, i64, dp, dp, dbl, i32, i32, i32, soa_p]),
        "orc_philox_block": (None, [u32p, u32p, u32p]),
        "orc_uniform": (dbl, [ct.c_uint64, ct.c_uint32, ct.c_uint64]),
        "orc_eval_fn": (i32, [i32, i64, dp, dp, dp]),
        "orc_scattering": (None, [vp, ct.c_uint64, dbl, dbl, dbl, dbl, dbl, dp, dp, dp, dp]),
        "orc_transform_p_PS": (None, [dbl] * 11 + [dp]),
        "orc_transform_p_PSP": (None, [vp] + [dbl] * 5 + [dp, dp, dp]),
        "orc_set_long_draws": (i32, [vp, i64]),
        "orc_count_long": (i64, [vp, u8p, i64]),
        "orc_new_pcut_ordered": (i64, [vp, i64, i64, u8p, soa_p, soa_p]),
        "orc_bin_momentum": (i32, [vp, dbl]),
        "orc_bin_angle": (i32, [vp, dbl, dbl]),
        # oracle/mcs_iter.cpp: the CPU twin of montecarloscattering.jl_amd/iter_finalize.py
        "orc_upstream_fluxes": (i32, [i32, dp, dp, dp, dbl, dbl, dbl, dbl, dbl, dp]),
        "orc_q_esc_calcs": (i32, [dbl, dbl, dbl, i32, dp, dp, dp, dbl, dbl, dbl, dbl, dbl, dbl, dp]),
        "orc_set_gamma_grid": (i32, [dp, i32, i32, dp, dbl, dp, dp, dp]),
        "orc_tcut_print": (i32, [dp, dp, i32, i32, i32]),
        "orc_smooth_grid_par": (i32, [i32, i32, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp] + [dbl] * 16 + [dp]),
    }
    if capi is not None:
        cin_p = ct.POINTER(capi.McsConsumerIn)
        sig["orc_dndp_cr"] = (i32, [par_p, dp, cin_p, dp, dp, dp, i64p])
        sig["orc_thermo_calcs"] = (i32, [par_p, dp, i64p, cin_p, dp, dp, dp, dp, dp])
        sig["orc_photon_synch"] = (i32, [par_p, dp, dp, dp, dbl, i32, dbl, dbl, dp, dp])
        sig["orc_dndp_2d"] = (i32, [par_p, dp, i64p, cin_p, dbl, dbl, dp])
        sig["orc_photon_ic"] = (i32, [par_p, dp, dp, dbl, i32, i32, dp, dp, i32, dbl, dbl, dbl, dp, dp])
        sig["orc_synch_F"] = (dbl, [dbl])
        sig["orc_photon_pion"] = (i32, [par_p, dp, dp, dbl, dbl, dp, dbl, i32, i32, dbl, dbl, dp, dp])
        sig["orc_pion_sigma_pi"] = (dbl, [dbl, i32])
        sig["orc_pion_amax"] = (None, [dbl, i32, dp, dp])
        sig["orc_pion_F"] = (dbl, [dbl, dbl, i32, dbl])
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def _dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ct.POINTER(ct.c_double))


class OracleBackend:
    """CPU oracle behind the driver's backend protocol."""

    name = "oracle"

    def __init__(self, capi, math: str = "det", nthreads: int = 1):
        self.capi = capi
        self.lib = load(math, capi)
        self.math = math
        self.nthreads = nthreads
        self.h = None
        self.pop = None
        self.saved = None
        self.l_save = None

    # -- lifecycle
    def create(self, prob):
        self.prob = prob
        self.P = prob.params
        self.layout = self.capi.Layout(self.P)
        self.h = self.lib.orc_create(ct.byref(self.P))
        if not self.h:
            raise RuntimeError(self.lib.orc_last_error().decode())
        self.set_grid(prob)
        self.set_cuts(prob)

    def set_grid(self, prob):
        tabs = [np.ascontiguousarray(t) for t in prob.grid_tables()]
        self._chk(self.lib.orc_set_grid(self.h, len(tabs[0]), *[_dp(t) for t in tabs]))

    def set_cuts(self, prob):
        self._keep = [np.ascontiguousarray(a, dtype=np.float64) for a in
                      (prob.pcuts, prob.tcuts, prob.x_spec, prob.inj_fracs, prob.eps_target)]
        pc, tc, xs, inj, eps = self._keep
        self._chk(self.lib.orc_set_cuts(self.h, len(pc), _dp(pc), len(tc), _dp(tc), len(xs), _dp(xs), _dp(inj), _dp(eps)))

    def destroy(self):
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.orc_last_error().decode())

    # -- per iteration / species
    def begin_iteration(self, i_iter):
        self._chk(self.lib.orc_begin_iteration(self.h, i_iter))

    def begin_species(self, i_iter, i_ion, aa, zz, pmax_cutoff, density, ewf):
        self._chk(self.lib.orc_begin_species(self.h, i_iter, i_ion, aa, zz, pmax_cutoff, density, ewf))

    def set_fluxes(self, pxx, pxz, en):
        a, b, c = (np.ascontiguousarray(v, dtype=np.float64) for v in (pxx, pxz, en))
        self._chk(self.lib.orc_set_fluxes(self.h, _dp(a), _dp(b), _dp(c)))

    # -- population
    def init_pop(self, inj, j_offset, n_local, n_total, j_stride=1):
        pop = self.capi.Population(n_local)
        ptot = np.ascontiguousarray(inj.ptot_pf[j_offset::j_stride][:n_local])
        w = np.ascontiguousarray(inj.weight[j_offset::j_stride][:n_local])
        assert len(ptot) == n_local
        s = pop.soa()
        self._chk(self.lib.orc_init_pop(self.h, n_local, j_offset, j_stride, n_total, _dp(ptot), _dp(w), inj.x_start_cm,
                                        inj.i_grid_start, int(inj.relativistic), int(inj.fast_push), ct.byref(s)))
        self.pop = pop

    def set_population(self, pop):
        self.pop = pop

    def get_population(self):
        return self.pop

    def pop_size(self):
        return self.pop.n

    def run_pcut(self, i_pcut, i_prt_offset, i_prt_stride=1, gidx=None):
        """gidx (int64 tensor / array, one global index per local particle) overrides the affine rule."""
        n = self.pop.n
        if gidx is not None:
            g = np.ascontiguousarray(np.asarray(gidx, dtype=np.int64))
            assert len(g) == n
            self._gidx = g
            gp = g.ctypes.data_as(ct.POINTER(ct.c_int64))
        else:
            self._gidx = int(i_prt_offset) + np.arange(n, dtype=np.int64) * int(i_prt_stride)
            gp = None
        self.saved = self.capi.Population(n)
        self.l_save = np.zeros(n, dtype=np.uint8)
        ns = ct.c_int64(0)
        si, so = self.pop.soa(), self.saved.soa()
        if getattr(self, "f32_exact", False):
            # the fp32-state variant in its exact form (oracle/mcs_oracle_f32.inc; twin of MCS_F32_EXACT=1 on the device)
            self._chk(self.lib.orc_run_pcut_f32(self.h, i_pcut, n, i_prt_offset, i_prt_stride, gp, ct.byref(si), ct.byref(so),
                                                self.l_save.ctypes.data_as(ct.POINTER(ct.c_uint8)), ct.byref(ns)))
            return int(ns.value)
        self._chk(self.lib.orc_run_pcut(self.h, i_pcut, n, i_prt_offset, i_prt_stride, gp, ct.byref(si), ct.byref(so),
                                        self.l_save.ctypes.data_as(ct.POINTER(ct.c_uint8)), ct.byref(ns), self.nthreads))
        return int(ns.value)

    def run_pcut_indexed(self, i_pcut, gidx):
        return self.run_pcut(i_pcut, 0, 1, gidx=gidx)

    def saved_gidx(self):
        """Global indices of the particles the last run_pcut saved, in local order (twin of mcs_saved_gidx)."""
        import torch
        return torch.from_numpy(self._gidx[np.flatnonzero(self.l_save)].copy())

    def get_saved(self):
        return self.saved, self.l_save

    def set_retro_cap(self, cap):
        self._chk(self.lib.orc_set_retro_cap(self.h, int(cap)))

    # -- multi-rank new_pcut (the CPU twin of mcs_saved_export / mcs_split_import, for the gloo tests of the driver)
    def export_saved(self, cap):
        import torch
        F = self.capi.F64_FIELDS
        idx = np.flatnonzero(self.l_save)
        n = len(idx)
        assert cap >= n
        gidx = np.zeros(cap, np.int64); f64 = np.zeros((8, cap)); meta = np.zeros(cap, np.int32)
        gidx[:n] = self._gidx[idx]
        for r, f in enumerate(F):
            f64[r, :n] = getattr(self.saved, f)[idx]
        s = self.saved
        meta[:n] = (s.grid[idx] & 0xffff) | ((s.tcut[idx] & 0xff) << 16) | (s.downstream[idx].astype(np.int64) << 24) | \
                   (s.inj[idx].astype(np.int64) << 25)
        return torch.from_numpy(gidx), torch.from_numpy(f64), torch.from_numpy(meta)

    def import_split(self, f64, meta, n_parents, i_mult, first, stride, n_local):
        F = self.capi.F64_FIELDS
        f64 = f64.numpy(); meta = meta.numpy().astype(np.int64)
        par = (first + np.arange(n_local, dtype=np.int64) * stride) // i_mult
        assert n_local == 0 or par[-1] < n_parents
        out = self.capi.Population(n_local)
        for r, f in enumerate(F):
            getattr(out, f)[:] = f64[r, par]
        out.weight[:] = f64[0, par] / float(i_mult)
        m = meta[par]
        out.grid[:] = m & 0xffff; out.tcut[:] = (m >> 16) & 0xff
        out.downstream[:] = (m >> 24) & 1; out.inj[:] = (m >> 25) & 1
        self.pop = out

    def finals(self):
        n = self.pop.n
        reason = np.zeros(n, np.int32); helix = np.zeros(n, np.int32); retro = np.zeros(n, np.int32)
        ptot = np.zeros(n); x = np.zeros(n)
        i32p = ct.POINTER(ct.c_int32)
        self._chk(self.lib.orc_finals(self.h, n, reason.ctypes.data_as(i32p), helix.ctypes.data_as(i32p),
                                      retro.ctypes.data_as(i32p), _dp(ptot), _dp(x)))
        return dict(reason=reason, helix=helix, retro=retro, ptot=ptot, x=x)

    def set_long_draws(self, long_draws):
        """Tell long histories apart (>= long_draws random draws in a pcut): new_pcut then orders the children of the saved particles that
        are not long before those of the saved long ones -- the order mcs_run_pcuts_pipelined produces.  0: the reference's order."""
        self.long_draws = int(long_draws)
        self._chk(self.lib.orc_set_long_draws(self.h, int(long_draws)))

    def count_long_saved(self):
        return int(self.lib.orc_count_long(self.h, self.l_save.ctypes.data_as(ct.POINTER(ct.c_uint8)), len(self.l_save)))

    def new_pcut(self, i_mult):
        n_saved = int(self.l_save.sum())
        out = self.capi.Population(n_saved * i_mult)
        ss, so = self.saved.soa(), out.soa()
        if getattr(self, "long_draws", 0):
            n_new = self.lib.orc_new_pcut_ordered(self.h, self.pop.n, i_mult, self.l_save.ctypes.data_as(ct.POINTER(ct.c_uint8)),
                                                  ct.byref(ss), ct.byref(so))
            assert n_new == out.n
            self.pop = out
            return int(n_new)
        n_new = self.lib.orc_new_pcut(self.pop.n, i_mult, self.l_save.ctypes.data_as(ct.POINTER(ct.c_uint8)),
                                      ct.byref(ss), ct.byref(so))
        assert n_new == out.n
        self.pop = out
        return int(n_new)

    # -- tallies
    def read_tallies(self):
        f = np.zeros(self.layout.total)
        i = np.zeros(self.layout.n_i64, dtype=np.int64)
        self._chk(self.lib.orc_read_tallies(self.h, _dp(f), i.ctypes.data_as(ct.POINTER(ct.c_int64))))
        return f, i

    def write_tallies(self, f, i):
        f = np.ascontiguousarray(f, dtype=np.float64)
        i = np.ascontiguousarray(i, dtype=np.int64)
        self._chk(self.lib.orc_write_tallies(self.h, _dp(f), i.ctypes.data_as(ct.POINTER(ct.c_int64))))

    def write_tally(self, name, arr):
        f, i = self.read_tallies()
        self.layout.view(f, name)[...] = np.asarray(arr).reshape(self.layout.shapes[name])
        self.write_tallies(f, i)

    def photon_synch(self, dndp_pf, mom_edge_cgs, mc, n_photon, emin_mev, bins_per_dec):
        d = np.ascontiguousarray(dndp_pf, dtype=np.float64); pe = np.ascontiguousarray(mom_edge_cgs, dtype=np.float64)
        bt = np.ascontiguousarray(self.prob.btot, dtype=np.float64)
        E = np.zeros(n_photon); out = np.zeros((self.P.n_grid, n_photon))
        self._chk(self.lib.orc_photon_synch(ct.byref(self.P), _dp(d), _dp(pe), _dp(bt), float(mc), int(n_photon), float(emin_mev),
                                            float(bins_per_dec), _dp(E), _dp(out)))
        return E, out

    def photon_pion(self, dndp_pf, mom_edge_cgs, mc, aa, target_density, scaling, n_photon, emin_mev, bins_per_dec, i_data=1):
        d = np.ascontiguousarray(dndp_pf, dtype=np.float64); pe = np.ascontiguousarray(mom_edge_cgs, dtype=np.float64)
        td = np.ascontiguousarray(target_density, dtype=np.float64)
        assert d.shape == (self.P.n_grid, self.P.num_psd_mom_bins + 2) and td.shape == (self.P.n_grid,)
        E = np.zeros(n_photon); out = np.zeros((self.P.n_grid, n_photon))
        self._chk(self.lib.orc_photon_pion(ct.byref(self.P), _dp(d), _dp(pe), float(mc), float(aa), _dp(td), float(scaling), int(i_data), int(n_photon),
                                           float(emin_mev), float(bins_per_dec), _dp(E), _dp(out)))
        return E, out

    # -- consumers of the tallies (oracle/mcs_consumers.cpp)
    def dndp_cr(self, tabs, tallies=None):
        f, _ = self.read_tallies() if tallies is None else tallies
        ux, gsf = np.ascontiguousarray(self.prob.ux), np.ascontiguousarray(self.prob.gam_sf)
        P = self.P
        out = np.zeros((3, P.n_grid, P.num_psd_mom_bins + 2))
        diag = np.zeros(2, dtype=np.int64)
        s = tabs.as_struct()
        self._chk(self.lib.orc_dndp_cr(ct.byref(P), _dp(f), ct.byref(s), _dp(gsf), _dp(ux), _dp(out),
                                       diag.ctypes.data_as(ct.POINTER(ct.c_int64))))
        return out, diag

    def dndp_2d(self, tabs, gam_x, beta_x, tallies=None):
        """get_dNdp_2D (CPU twin of mcs_dndp_2d) -> [n_grid][ntht+2][nmom+2]; kept for photon_ic."""
        f, i = self.read_tallies() if tallies is None else tallies
        P = self.P
        out = np.zeros((P.n_grid, P.num_psd_tht_bins + 2, P.num_psd_mom_bins + 2))
        s = tabs.as_struct()
        self._chk(self.lib.orc_dndp_2d(ct.byref(P), _dp(f), i.ctypes.data_as(ct.POINTER(ct.c_int64)), ct.byref(s), float(gam_x), float(beta_x), _dp(out)))
        self._d2n = out
        return out

    def photon_ic(self, mom_edge_cgs, mc_e, j_max, alpha_in, n_in, n_photon, emin_mev, bins_per_dec, beam_area):
        pe = np.ascontiguousarray(mom_edge_cgs, dtype=np.float64)
        ai = np.ascontiguousarray(alpha_in, dtype=np.float64); ni = np.ascontiguousarray(n_in, dtype=np.float64)
        E = np.zeros(n_photon); out = np.zeros((self.P.n_grid, n_photon))
        self._chk(self.lib.orc_photon_ic(ct.byref(self.P), _dp(self._d2n), _dp(pe), float(mc_e), int(j_max), int(len(ai)), _dp(ai), _dp(ni),
                                         int(n_photon), float(emin_mev), float(bins_per_dec), float(beam_area), _dp(E), _dp(out)))
        return E, out

    def thermo_calcs(self, tabs, tallies=None):
        f, i = self.read_tallies() if tallies is None else tallies
        ux, gsf = np.ascontiguousarray(self.prob.ux), np.ascontiguousarray(self.prob.gam_sf)
        n = self.P.n_grid
        a, b, c = np.zeros(n), np.zeros(n), np.zeros(n)
        s = tabs.as_struct()
        self._chk(self.lib.orc_thermo_calcs(ct.byref(self.P), _dp(f), i.ctypes.data_as(ct.POINTER(ct.c_int64)), ct.byref(s),
                                            _dp(gsf), _dp(ux), _dp(a), _dp(b), _dp(c)))
        return a, b, c

    def last_kernel_ms(self):
        return float("nan")
