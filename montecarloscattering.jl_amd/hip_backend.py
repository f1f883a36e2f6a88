"""The compute backend of the transport path: libmcs_hip.so (gfx950 kernels)
through the C ABI of include/mcs.h.

There is no other backend in this package and no fallback: if the library or a
GPU is missing, construction fails with an error.  (Tests drive the same host
driver with the CPU oracle by injecting a backend object defined under oracle/;
the product never does.)
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import capi
from .capi import McsSoa, Population, c_double_p, c_int64_p, c_uint8_p, c_int32_p


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


class HipBackend:
    name = "hip"

    def __init__(self, device: int = 0, stream: int = 0, torch_tallies: bool = False, debug_finals: bool = False):
        """torch_tallies: keep the flat tally buffers in torch CUDA tensors, so that torch.distributed can
        all-reduce them in place (multi-GPU driver).  The context then works on torch's CURRENT stream
        (passed to mcs_create), so torch ops and library kernels are ordered by that stream; with
        stream=0 and torch_tallies=False the context creates its own blocking stream.
        debug_finals: record per-particle end states for finals() (tests; 24 B of stores per particle)."""
        self.lib = capi.load_library()          # raises MissingNativeLibrary
        if self.lib.mcs_abi_version() != capi.MCS_ABI_VERSION:
            raise RuntimeError("libmcs_hip.so ABI version mismatch")
        self.device = int(device)
        self.stream = int(stream)
        self.torch_tallies = torch_tallies
        self.debug_finals = debug_finals
        self.h = ct.c_void_p(None)
        self._bound = None

    # -- lifecycle
    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("libmcs_hip: " + self.lib.mcs_last_error().decode())

    def create(self, prob):
        self.prob = prob
        self.P = prob.params
        self.layout = capi.Layout(self.P)
        if self.torch_tallies:
            import torch
            torch.cuda.set_device(self.device)
            self.stream = int(torch.cuda.current_stream(self.device).cuda_stream)
        self._chk(self.lib.mcs_create(ct.byref(self.P), self.device, ct.c_void_p(self.stream or None), ct.byref(self.h)))
        if self.debug_finals:
            self._chk(self.lib.mcs_set_debug_finals(self.h, 1))
        if self.torch_tallies:
            dev = torch.device("cuda", self.device)
            t_f = torch.zeros(self.layout.total, dtype=torch.float64, device=dev)
            t_i = torch.zeros(self.layout.n_i64, dtype=torch.int64, device=dev)
            self.bind_torch_tallies(t_f, t_i)
        self.set_grid(prob)
        self.set_cuts(prob)

    def set_grid(self, prob):
        """The grid tables of src/main_loops.jl:255-260 (again after every profile update, iter_finalize.py)."""
        tabs = [np.ascontiguousarray(t, dtype=np.float64) for t in prob.grid_tables()]
        self._chk(self.lib.mcs_set_grid(self.h, len(tabs[0]), *[_dp(t) for t in tabs]))

    def set_cuts(self, prob):
        pc, tc, xs, inj, eps = (np.ascontiguousarray(a, dtype=np.float64) for a in
                                (prob.pcuts, prob.tcuts, prob.x_spec, prob.inj_fracs, prob.eps_target))
        self._chk(self.lib.mcs_set_cuts(self.h, len(pc), _dp(pc), len(tc), _dp(tc), len(xs), _dp(xs), _dp(inj), _dp(eps)))

    def destroy(self):
        if self.h:
            self.lib.mcs_destroy(self.h)
            self.h = ct.c_void_p(None)

    def bind_torch_tallies(self, t_f64, t_i64):
        """Accumulate into caller-owned torch CUDA tensors (for in-place all-reduce)."""
        assert t_f64.is_cuda and t_f64.numel() >= self.layout.total
        self._bound = (t_f64, t_i64)
        self._chk(self.lib.mcs_bind_tallies(self.h, ct.c_void_p(t_f64.data_ptr()), t_f64.numel(),
                                            ct.c_void_p(t_i64.data_ptr()), t_i64.numel()))

    def tally_tensors(self):
        """Live device tensors (f64, i64) holding the tallies, or None when the context owns them."""
        return self._bound

    def num_cus(self) -> int:
        return int(self.lib.mcs_num_cus(self.h))

    def set_launch(self, blocks: int = 0, threads: int = 0):
        self._chk(self.lib.mcs_set_launch(self.h, blocks, threads))

    def set_tail_slicing(self, budget_trips: int):
        """Sliced tail of run_pcut* (mcs_set_tail_slicing): 0 = one launch per pcut."""
        self._chk(self.lib.mcs_set_tail_slicing(self.h, int(budget_trips)))

    def last_launches(self) -> int:
        return int(self.lib.mcs_last_launches(self.h))

    def last_kernel(self) -> int:
        """0 general, 1 plain, 2 lossy, 3 fp32, 4 fp32 plain loop, 5 fp32 lossy, 6 plain with energy transfer, 7 / 8 wave-specialised, 9 fp32 exact loop,
        10 general sliced, 11 / 12 / 13 the sliced forms of 1 / 2 / 6 (mcs_last_kernel)"""
        return int(self.lib.mcs_last_kernel(self.h))

    # -- per iteration / species
    def begin_iteration(self, i_iter):
        self._chk(self.lib.mcs_begin_iteration(self.h, i_iter))

    def begin_species(self, i_iter, i_ion, aa, zz, pmax_cutoff, density, ewf):
        self._chk(self.lib.mcs_begin_species(self.h, i_iter, i_ion, aa, zz, pmax_cutoff, density, ewf))

    def set_fluxes(self, pxx, pxz, en):
        a, b, c = (np.ascontiguousarray(v, dtype=np.float64) for v in (pxx, pxz, en))
        self._chk(self.lib.mcs_set_fluxes(self.h, _dp(a), _dp(b), _dp(c)))

    # -- population
    def init_pop(self, inj, j_offset, n_local, n_total, j_stride: int = 1):
        """K3 from the binned momentum discretisation: O(bins) host work and upload, whatever N.  Local particle k is
        global particle j_offset + k * j_stride."""
        bp = np.ascontiguousarray(inj.bin_ptot, dtype=np.float64)
        bw = np.ascontiguousarray(inj.bin_weight, dtype=np.float64)
        bs = np.ascontiguousarray(inj.bin_start, dtype=np.int64)
        self._chk(self.lib.mcs_init_pop_binned_strided(self.h, n_local, j_offset, j_stride, n_total, len(bp), _dp(bp), _dp(bw),
                                                       bs.ctypes.data_as(c_int64_p), inj.x_start_cm, inj.i_grid_start,
                                                       int(inj.relativistic), int(inj.fast_push)))

    def init_pop_arrays(self, inj, j_offset, n_local, n_total):
        """K3 from per-particle arrays (the reference's own form of the call)."""
        ptot = np.ascontiguousarray(inj.ptot_pf[j_offset:j_offset + n_local], dtype=np.float64)
        w = np.ascontiguousarray(inj.weight[j_offset:j_offset + n_local], dtype=np.float64)
        self._chk(self.lib.mcs_init_pop(self.h, n_local, j_offset, n_total, _dp(ptot), _dp(w), inj.x_start_cm,
                                        inj.i_grid_start, int(inj.relativistic), int(inj.fast_push)))

    def set_population(self, pop: Population):
        s = pop.soa()
        self._chk(self.lib.mcs_pop_upload(self.h, pop.n, ct.byref(s)))

    def get_population(self) -> Population:
        n = self.pop_size()
        pop = Population(n)
        s = pop.soa()
        self._chk(self.lib.mcs_pop_download(self.h, n, ct.byref(s)))
        return pop

    def pop_size(self) -> int:
        return int(self.lib.mcs_pop_size(self.h))

    def run_pcut(self, i_pcut, i_prt_offset, i_prt_stride: int = 1) -> int:
        """Local particle k has the global 0-based index i_prt_offset + k * i_prt_stride."""
        ns = ct.c_int64(0)
        self._chk(self.lib.mcs_run_pcut_strided(self.h, i_pcut, i_prt_offset, i_prt_stride, ct.byref(ns)))
        self._n_saved_last = int(ns.value)
        return int(ns.value)

    def run_pcut_indexed(self, i_pcut, gidx) -> int:
        """Local particle k has the global 0-based index gidx[k] (int64 CUDA tensor, kept alive here until the next run)."""
        import torch
        assert gidx.is_cuda and gidx.dtype == torch.int64 and gidx.is_contiguous() and gidx.numel() == self.pop_size()
        torch.cuda.current_stream(self.device).synchronize()
        self._gidx_keep = gidx
        ns = ct.c_int64(0)
        self._chk(self.lib.mcs_run_pcut_indexed(self.h, i_pcut, ct.c_void_p(gidx.data_ptr()), ct.byref(ns)))
        self._n_saved_last = int(ns.value)
        return int(ns.value)

    def saved_gidx(self):
        """Global indices of the particles the last run_pcut* saved, in local order (int64 CUDA tensor; mcs_saved_gidx)."""
        import torch
        n = self._n_saved_last
        g = torch.zeros(max(n, 1), dtype=torch.int64, device=torch.device("cuda", self.device))
        torch.cuda.current_stream(self.device).synchronize()
        self._chk(self.lib.mcs_saved_gidx(self.h, max(n, 1), ct.c_void_p(g.data_ptr())))
        return g[:n]

    def set_retro_cap(self, cap: int):
        self._chk(self.lib.mcs_set_retro_cap(self.h, int(cap)))

    # -- multi-GPU new_pcut: saved particles out (to be all-gathered), global split slice in
    def export_saved(self, cap: int):
        """(gidx int64[cap], f64[8, cap], meta int32[cap]) torch CUDA tensors; the first n_saved columns are
        the saved particles of the last run_pcut in index order (mcs_saved_export)."""
        import torch
        dev = torch.device("cuda", self.device)
        gidx = torch.zeros(cap, dtype=torch.int64, device=dev)
        f64 = torch.zeros((8, cap), dtype=torch.float64, device=dev)
        meta = torch.zeros(cap, dtype=torch.int32, device=dev)      # the packed uint32 word, bit pattern kept
        torch.cuda.current_stream(self.device).synchronize()
        self._chk(self.lib.mcs_saved_export(self.h, cap, ct.c_void_p(gidx.data_ptr()), ct.c_void_p(f64.data_ptr()),
                                            ct.c_void_p(meta.data_ptr())))
        return gidx, f64, meta

    def import_split(self, f64, meta, n_parents: int, i_mult: int, first: int, stride: int, n_local: int):
        """New local population = elements first + k*stride of the global split of the parents (mcs_split_import)."""
        import torch
        assert f64.is_cuda and f64.is_contiguous() and meta.is_contiguous() and f64.shape[0] == 8
        torch.cuda.current_stream(self.device).synchronize()
        self._chk(self.lib.mcs_split_import(self.h, n_parents, f64.shape[1], ct.c_void_p(f64.data_ptr()),
                                            ct.c_void_p(meta.data_ptr()), i_mult, first, stride, n_local))

    def run_pcut_host(self, i_pcut, pop: Population, i_prt_offset=0):
        """The literal drop-in call (host buffers in, saved arrays out)."""
        saved = Population(pop.n)
        l_save = np.zeros(pop.n, dtype=np.uint8)
        ns = ct.c_int64(0)
        si, so = pop.soa(), saved.soa()
        self._chk(self.lib.mcs_run_pcut_host(self.h, i_pcut, pop.n, i_prt_offset, ct.byref(si), ct.byref(so),
                                             l_save.ctypes.data_as(c_uint8_p), ct.byref(ns)))
        return saved, l_save, int(ns.value)

    def get_saved(self):
        n = self.pop_size()
        saved = Population(n)
        l_save = np.zeros(n, dtype=np.uint8)
        s = saved.soa()
        self._chk(self.lib.mcs_saved_download(self.h, n, ct.byref(s), l_save.ctypes.data_as(c_uint8_p)))
        return saved, l_save

    def finals(self):
        if not self.debug_finals:
            raise RuntimeError("finals() needs HipBackend(..., debug_finals=True)")
        n = self.pop_size()
        reason = np.zeros(n, np.int32); helix = np.zeros(n, np.int32); retro = np.zeros(n, np.int32)
        ptot = np.zeros(n); x = np.zeros(n)
        self._chk(self.lib.mcs_final_download(self.h, n, reason.ctypes.data_as(c_int32_p), helix.ctypes.data_as(c_int32_p),
                                              retro.ctypes.data_as(c_int32_p), _dp(ptot), _dp(x)))
        return dict(reason=reason, helix=helix, retro=retro, ptot=ptot, x=x)

    def new_pcut(self, i_mult) -> int:
        nn = ct.c_int64(0)
        self._chk(self.lib.mcs_new_pcut(self.h, i_mult, ct.byref(nn)))
        return int(nn.value)

    # -- tallies
    def sync(self):
        """Everything launched so far is done and the bound tally tensors are complete (replicas folded in)."""
        self._chk(self.lib.mcs_sync(self.h))

    def read_tallies(self):
        f = np.zeros(self.layout.total)
        i = np.zeros(self.layout.n_i64, dtype=np.int64)
        self._chk(self.lib.mcs_read_tallies(self.h, _dp(f), i.ctypes.data_as(c_int64_p)))
        return f, i

    def read_tallies_light(self):
        """The same shapes as read_tallies, but only the part of the fp64 buffer behind the three big histograms is
        fetched (fluxes, escape spectra, coupled spectra, pools, scalars: what the host's iter_finalize reads); psd /
        therm_sf / therm_pf read as zeros -- their consumers (K4) run on the device."""
        f = np.zeros(self.layout.total)
        i = np.zeros(self.layout.n_i64, dtype=np.int64)
        first = int(self.layout.offsets["esc_psd_up"])
        tail = f[first:]
        self._chk(self.lib.mcs_read_tallies_part(self.h, first, tail.size, _dp(tail), i.ctypes.data_as(c_int64_p)))
        return f, i

    def run_pcuts_fused(self, i_pcut_first: int, i_pcut_last: int, n_target):
        """mcs_run_pcuts_fused: the pcuts first .. last of the current species queued back to back, n_saved / i_mult / the next
        population's size decided on the device -> (n_use, n_saved, i_mult, kernel_ms) per pcut.  MCS_FUSED_PCUTS=0: not offered."""
        npc = i_pcut_last - i_pcut_first + 1
        tg = np.ascontiguousarray(n_target, dtype=np.int64)
        assert tg.shape == (npc,)
        nu, ns, im = (np.zeros(npc, dtype=np.int64) for _ in range(3))
        ms = np.zeros(npc)
        self._chk(self.lib.mcs_run_pcuts_fused(self.h, int(i_pcut_first), int(i_pcut_last), tg.ctypes.data_as(c_int64_p), nu.ctypes.data_as(c_int64_p),
                                               ns.ctypes.data_as(c_int64_p), im.ctypes.data_as(c_int64_p), _dp(ms)))
        return nu, ns, im, ms

    def run_pcuts_pipelined(self, i_pcut_first: int, i_pcut_last: int, n_target, long_draws: int, long_imult_max: int = 0):
        """mcs_run_pcuts_pipelined: the pcuts first .. last of the current species with every pcut's long histories (>= long_draws random
        draws) finishing beside the next pcut on a second stream -> (n_use, n_saved, i_mult, kernel_ms, strag[npc][2]) per pcut; strag:
        particles exported, and 1 where i_mult had to wait for them.  The next population is ordered non-long before long (include/mcs.h);
        long_imult_max > 0: only in pcuts whose predecessor split by at most that factor."""
        npc = i_pcut_last - i_pcut_first + 1
        tg = np.ascontiguousarray(n_target, dtype=np.int64)
        assert tg.shape == (npc,)
        nu, ns, im = (np.zeros(npc, dtype=np.int64) for _ in range(3))
        ms = np.zeros(npc)
        sg = np.zeros((npc, 2), dtype=np.int64)
        self._chk(self.lib.mcs_run_pcuts_pipelined(self.h, int(i_pcut_first), int(i_pcut_last), tg.ctypes.data_as(c_int64_p), int(long_draws), int(long_imult_max),
                                                   nu.ctypes.data_as(c_int64_p), ns.ctypes.data_as(c_int64_p), im.ctypes.data_as(c_int64_p), _dp(ms),
                                                   sg.ctypes.data_as(c_int64_p)))
        return nu, ns, im, ms, sg

    def read_counters(self):
        """The int64 tallies alone (num_crossings + event counters: ~1 KB), no fp64 word."""
        i = np.zeros(self.layout.n_i64, dtype=np.int64)
        self._chk(self.lib.mcs_read_tallies_part(self.h, 0, 0, None, i.ctypes.data_as(c_int64_p)))
        return i

    def write_tallies(self, f, i):
        f = np.ascontiguousarray(f, dtype=np.float64)
        i = np.ascontiguousarray(i, dtype=np.int64)
        self._chk(self.lib.mcs_write_tallies(self.h, _dp(f), i.ctypes.data_as(c_int64_p)))

    def write_tally(self, name: str, arr):
        """One named array of the layout back into the device buffer (mcs_write_tallies_part)."""
        a = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        assert a.size == int(np.prod(self.layout.shapes[name]))
        self._chk(self.lib.mcs_write_tallies_part(self.h, int(self.layout.offsets[name]), a.size, _dp(a)))

    # -- consumers of the tallies (K4)
    def dndp_cr(self, tabs):
        """get_dNdp_cr + CR normalisation on the resident psd -> ([3][n_grid][nmom+2], diag[2])."""
        P = self.P
        out = np.zeros((3, P.n_grid, P.num_psd_mom_bins + 2))
        diag = np.zeros(2, dtype=np.int64)
        s = tabs.as_struct()
        self._chk(self.lib.mcs_dndp_cr(self.h, ct.byref(s), _dp(out), diag.ctypes.data_as(c_int64_p)))
        return out, diag

    def thermo_calcs(self, tabs):
        """thermo_calcs on the resident psd / therm_pf / num_crossings -> (P_par, P_perp, energy_density)."""
        n = self.P.n_grid
        a, b, c = np.zeros(n), np.zeros(n), np.zeros(n)
        s = tabs.as_struct()
        self._chk(self.lib.mcs_thermo_calcs(self.h, ct.byref(s), _dp(a), _dp(b), _dp(c)))
        return a, b, c

    def photon_synch(self, dndp_pf, mom_edge_cgs, mc, n_photon, emin_mev, bins_per_dec):
        """K5: synchrotron emission dP/d(ln E) [erg/s] per zone from the plasma-frame electron dN/dp -> (E_erg[n_photon], emis[n_grid][n_photon])."""
        d = np.ascontiguousarray(dndp_pf, dtype=np.float64); pe = np.ascontiguousarray(mom_edge_cgs, dtype=np.float64)
        assert d.shape == (self.P.n_grid, self.P.num_psd_mom_bins + 2) and pe.shape == (self.P.num_psd_mom_bins + 2,)
        E = np.zeros(n_photon); out = np.zeros((self.P.n_grid, n_photon))
        self._chk(self.lib.mcs_photon_synch(self.h, _dp(d), _dp(pe), float(mc), int(n_photon), float(emin_mev), float(bins_per_dec), _dp(E), _dp(out)))
        return E, out

    def photon_pion(self, dndp_pf, mom_edge_cgs, mc, aa, target_density, scaling, n_photon, emin_mev, bins_per_dec, i_data=1):
        """K7: pion-decay emission dP/d(ln E) [erg/s] per zone from the plasma-frame dN/dp of a nucleus species -> (E_erg[n_photon],
        emis[n_grid][n_photon])."""
        d = np.ascontiguousarray(dndp_pf, dtype=np.float64); pe = np.ascontiguousarray(mom_edge_cgs, dtype=np.float64)
        td = np.ascontiguousarray(target_density, dtype=np.float64)
        assert d.shape == (self.P.n_grid, self.P.num_psd_mom_bins + 2) and pe.shape == (self.P.num_psd_mom_bins + 2,)
        assert td.shape == (self.P.n_grid,)
        E = np.zeros(n_photon); out = np.zeros((self.P.n_grid, n_photon))
        self._chk(self.lib.mcs_photon_pion(self.h, _dp(d), _dp(pe), float(mc), float(aa), _dp(td), float(scaling), int(i_data), int(n_photon),
                                           float(emin_mev), float(bins_per_dec), _dp(E), _dp(out)))
        return E, out

    def dndp_2d(self, tabs, gam_x, beta_x, download=True):
        """K6: get_dNdp_2D on the resident histograms -> d2N/dp dcos [n_grid][ntht+2][nmom+2] in the frame (gam_x, beta_x); the array
        stays on the device for photon_ic (download=False: nothing crosses PCIe)."""
        P = self.P
        out = np.zeros((P.n_grid, P.num_psd_tht_bins + 2, P.num_psd_mom_bins + 2)) if download else None
        s = tabs.as_struct()
        self._chk(self.lib.mcs_dndp_2d(self.h, ct.byref(s), float(gam_x), float(beta_x), _dp(out) if download else None))
        return out

    def photon_ic(self, mom_edge_cgs, mc_e, j_max, alpha_in, n_in, n_photon, emin_mev, bins_per_dec, beam_area):
        """K6: inverse-Compton energy flux per d(ln E) at Earth [erg/(s cm^2)] per zone from the d2N/dp dcos of the last dndp_2d."""
        pe = np.ascontiguousarray(mom_edge_cgs, dtype=np.float64)
        ai = np.ascontiguousarray(alpha_in, dtype=np.float64); ni = np.ascontiguousarray(n_in, dtype=np.float64)
        assert pe.shape == (self.P.num_psd_mom_bins + 2,) and ai.shape == ni.shape
        E = np.zeros(n_photon); out = np.zeros((self.P.n_grid, n_photon))
        self._chk(self.lib.mcs_photon_ic(self.h, _dp(pe), float(mc_e), int(j_max), int(len(ai)), _dp(ai), _dp(ni), int(n_photon), float(emin_mev),
                                         float(bins_per_dec), float(beam_area), _dp(E), _dp(out)))
        return E, out

    def last_kernel_ms(self) -> float:
        return float(self.lib.mcs_last_kernel_ms(self.h))

    def eval_fn(self, name: str, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = a if b is None else np.ascontiguousarray(b, dtype=np.float64)
        out = np.zeros_like(a)
        self._chk(self.lib.mcs_eval_fn(self.h, capi.FN[name], len(a), _dp(a), _dp(b), _dp(out)))
        return out
