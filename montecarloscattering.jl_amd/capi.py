"""ctypes mirror of include/mcs.h (the C ABI of the transport path).

The structures here are laid out exactly as in the header; `tests/test_abi.py`
checks sizes, the tally layout and that the shared library exports every
declared symbol.
"""
from __future__ import annotations

import ctypes as ct
import os
from typing import Dict

import numpy as np

from .constants import PSD_MAX, NA_C

MCS_ABI_VERSION = 3

c_double_p = ct.POINTER(ct.c_double)
c_int64_p = ct.POINTER(ct.c_int64)
c_uint8_p = ct.POINTER(ct.c_uint8)
c_int32_p = ct.POINTER(ct.c_int32)


class McsParams(ct.Structure):
    """`mcs_params`: the scalars/flags of src/main_loops.jl:236-264."""
    _fields_ = [
        ("abi_version", ct.c_int32),
        ("n_ions", ct.c_int32), ("n_grid", ct.c_int32), ("n_itrs", ct.c_int32),
        ("n_pts_max", ct.c_int64),
        ("i_grid_feb", ct.c_int32), ("i_shock", ct.c_int32),
        ("num_psd_mom_bins", ct.c_int32), ("num_psd_tht_bins", ct.c_int32),
        ("psd_bins_per_dec_mom", ct.c_int32), ("psd_bins_per_dec_tht", ct.c_int32),
        ("psd_cos_fine", ct.c_double), ("psd_dcos", ct.c_double), ("psd_tht_min", ct.c_double),
        ("psd_mom_min", ct.c_double),
        ("gam0", ct.c_double), ("beta0", ct.c_double), ("u0", ct.c_double), ("u2", ct.c_double),
        ("bmag2", ct.c_double),
        ("pe_crit", ct.c_double), ("game_crit", ct.c_double), ("eta_mfp", ct.c_double),
        ("energy_transfer_frac", ct.c_double),
        ("feb_upstream", ct.c_double), ("feb_downstream", ct.c_double), ("x_grid_stop", ct.c_double),
        ("B_CMBz", ct.c_double), ("age_max", ct.c_double),
        ("xn_per_fine", ct.c_double), ("xn_per_coarse", ct.c_double),
        ("use_custom_epsB", ct.c_int32), ("do_rad_losses", ct.c_int32), ("do_retro", ct.c_int32),
        ("do_tcuts", ct.c_int32),
        ("dont_DSA", ct.c_int32), ("dont_scatter", ct.c_int32), ("use_custom_frg", ct.c_int32),
        ("track_thermal", ct.c_int32),
        ("state_fp32", ct.c_int32),
    ]


class McsSoa(ct.Structure):
    """`mcs_soa`: host struct-of-arrays population (Float64 / Int / Bool as in Julia)."""
    _fields_ = [
        ("weight", c_double_p), ("ptot_pf", c_double_p), ("pb_pf", c_double_p), ("x_PT_cm", c_double_p),
        ("xn_per", c_double_p), ("prp_x_cm", c_double_p), ("acctime_sec", c_double_p), ("phi_rad", c_double_p),
        ("grid", c_int64_p), ("tcut", c_int64_p),
        ("downstream", c_uint8_p), ("inj", c_uint8_p),
    ]


class McsConsumerIn(ct.Structure):
    """`mcs_consumer_in`: host tables for the tally consumers (include/mcs.h)."""
    _fields_ = [
        ("mom_log_cgs", c_double_p), ("mom_edge_cgs", c_double_p), ("cos_edge", c_double_p),
        ("cos_center", c_double_p), ("pt_center", c_double_p), ("zone_pop", c_double_p),
        ("density_loc", c_double_p), ("cold_pressure", c_double_p),
        ("rest_energy", ct.c_double), ("mc", ct.c_double), ("n0", ct.c_double), ("gam0", ct.c_double),
        ("therm_from_hist", ct.c_int),
    ]


F64_FIELDS = ("weight", "ptot_pf", "pb_pf", "x_PT_cm", "xn_per", "prp_x_cm", "acctime_sec", "phi_rad")
I64_FIELDS = ("grid", "tcut")
U8_FIELDS = ("downstream", "inj")


class Population:
    """Numpy struct-of-arrays particle population; the `*_new` / `*_saved`
    arrays of src/MonteCarloScattering.jl:556-585."""

    def __init__(self, n: int):
        self.n = int(n)
        for f in F64_FIELDS:
            setattr(self, f, np.zeros(self.n, dtype=np.float64))
        for f in I64_FIELDS:
            setattr(self, f, np.zeros(self.n, dtype=np.int64))
        for f in U8_FIELDS:
            setattr(self, f, np.zeros(self.n, dtype=np.uint8))

    def soa(self) -> McsSoa:
        s = McsSoa()
        for f in F64_FIELDS:
            setattr(s, f, getattr(self, f).ctypes.data_as(c_double_p))
        for f in I64_FIELDS:
            setattr(s, f, getattr(self, f).ctypes.data_as(c_int64_p))
        for f in U8_FIELDS:
            setattr(s, f, getattr(self, f).ctypes.data_as(c_uint8_p))
        return s

    def fields(self):
        return F64_FIELDS + I64_FIELDS + U8_FIELDS

    def take(self, idx) -> "Population":
        out = Population(len(idx))
        for f in self.fields():
            getattr(out, f)[:] = getattr(self, f)[idx]
        return out

    def slice(self, a: int, b: int) -> "Population":
        return self.take(np.arange(a, b))

    @staticmethod
    def concat(pops) -> "Population":
        out = Population(sum(p.n for p in pops))
        for f in out.fields():
            if pops:
                getattr(out, f)[:] = np.concatenate([getattr(p, f) for p in pops])
        return out


# int64 tally slots after the n_grid num_crossings entries (enum in mcs.h)
IC = {name: i for i, name in enumerate([
    "STEPS_HELIX", "STEPS_RETRO", "HELIX_CAP", "PPERP_CLAMP", "PSP_CLAMP", "MOMBIN_CLAMP",
    "REASON0", "REASON1", "REASON2", "REASON3", "REASON4", "TCUT_OVERRUN", "RNG_DRAWS", "ZONE_FAIL", "RETRO_CAP"])}
IC_COUNT = len(IC)

FN = {name: i for i, name in enumerate(
    ["sin", "cos", "asin", "acos", "atan2", "log10", "mod2pi", "sqrt", "div", "hypot1", "uniform"])}


class Layout:
    """Python mirror of `mcs_tally_layout` (include/mcs.h): offsets, in doubles,
    of every fp64 tally inside the flat buffer."""

    def __init__(self, P: McsParams):
        nm, nt, ng = P.num_psd_mom_bins + 2, P.num_psd_tht_bins + 2, P.n_grid
        pm = PSD_MAX + 1
        self.shapes: Dict[str, tuple] = {}
        self.offsets: Dict[str, int] = {}
        o = 0
        for name, shape in [
            ("psd", (ng, nt, nm)), ("therm_sf", (ng, nt, nm)), ("therm_pf", (ng, nt, nm)),
            ("esc_psd_up", (pm, pm)), ("esc_psd_down", (pm, pm)),
            ("pxx_flux", (ng,)), ("pxz_flux", (ng,)), ("energy_flux", (ng,)),
            ("esc_flux", (P.n_ions,)),
            ("px_esc_feb", (P.n_itrs, P.n_ions)), ("energy_esc_feb", (P.n_itrs, P.n_ions)),
            ("esc_energy_eff", (P.n_ions, pm)), ("esc_num_eff", (P.n_ions, pm)),
            ("weight_coupled", (P.n_ions, NA_C)), ("spectra_coupled", (P.n_ions, NA_C, pm)),
            ("spectra_sf", (ng, pm)), ("spectra_pf", (ng, pm)),
            ("energy_transfer_pool", (ng,)), ("energy_recv_pool", (ng,)),
            ("scalars", (4,)),
        ]:
            self.offsets[name] = o
            self.shapes[name] = shape      # C-order shape == reversed Julia shape
            o += int(np.prod(shape))
        self.total = o
        self.n_i64 = ng + IC_COUNT
        self.n_grid = ng

    def view(self, flat: np.ndarray, name: str) -> np.ndarray:
        o = self.offsets[name]
        shp = self.shapes[name]
        return flat[o:o + int(np.prod(shp))].reshape(shp)


def _as_dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def lib_path() -> str:
    # MCS_HIP_LIB: pick another build of the SAME library (kernel tuning variants)
    name = os.environ.get("MCS_HIP_LIB", "libmcs_hip.so")
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", name)


class MissingNativeLibrary(RuntimeError):
    pass


_LIB = None


def load_library() -> ct.CDLL:
    """Load libmcs_hip.so.  There is NO fallback: the product path is the HIP
    library or nothing (a missing library is an error, never a CPU substitute)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise MissingNativeLibrary(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). No CPU fallback exists for the transport path.")
    lib = ct.CDLL(path)
    vp = ct.c_void_p
    i32, i64, dbl = ct.c_int, ct.c_int64, ct.c_double
    soa_p = ct.POINTER(McsSoa)
    sig = {
        "mcs_abi_version": (i32, []),
        "mcs_last_error": (ct.c_char_p, []),
        "mcs_create": (i32, [ct.POINTER(McsParams), i32, vp, ct.POINTER(vp)]),
        "mcs_destroy": (i32, [vp]),
        "mcs_sync": (i32, [vp]),
        "mcs_bind_tallies": (i32, [vp, vp, i64, vp, i64]),
        "mcs_tallies_f64_devptr": (vp, [vp]),
        "mcs_tallies_i64_devptr": (vp, [vp]),
        "mcs_set_grid": (i32, [vp, i32] + [c_double_p] * 9),
        "mcs_set_cuts": (i32, [vp, i32, c_double_p, i32, c_double_p, i32, c_double_p, c_double_p, c_double_p]),
        "mcs_begin_iteration": (i32, [vp, i32]),
        "mcs_begin_species": (i32, [vp, i32, i32, dbl, dbl, dbl, dbl, dbl]),
        "mcs_set_fluxes": (i32, [vp, c_double_p, c_double_p, c_double_p]),
        "mcs_pop_upload": (i32, [vp, i64, soa_p]),
        "mcs_pop_download": (i32, [vp, i64, soa_p]),
        "mcs_saved_download": (i32, [vp, i64, soa_p, c_uint8_p]),
        "mcs_pop_size": (i64, [vp]),
        "mcs_init_pop": (i32, [vp, i64, i64, i64, c_double_p, c_double_p, dbl, i32, i32, i32]),
        "mcs_init_pop_binned": (i32, [vp, i64, i64, i64, i32, c_double_p, c_double_p, c_int64_p, dbl, i32, i32, i32]),
        "mcs_run_pcut": (i32, [vp, i32, i64, c_int64_p]),
        "mcs_run_pcut_strided": (i32, [vp, i32, i64, i64, c_int64_p]),
        "mcs_run_pcut_indexed": (i32, [vp, i32, vp, c_int64_p]),
        "mcs_saved_gidx": (i32, [vp, i64, vp]),
        "mcs_init_pop_binned_strided": (i32, [vp, i64, i64, i64, i64, i32, c_double_p, c_double_p, c_int64_p, dbl, i32, i32, i32]),
        "mcs_new_pcut": (i32, [vp, i64, c_int64_p]),
        "mcs_saved_export": (i32, [vp, i64, vp, vp, vp]),
        "mcs_split_import": (i32, [vp, i64, i64, vp, vp, i64, i64, i64, i64]),
        "mcs_set_debug_finals": (i32, [vp, i32]),
        "mcs_set_retro_cap": (i32, [vp, i64]),
        "mcs_run_pcut_host": (i32, [vp, i32, i64, i64, soa_p, soa_p, c_uint8_p, c_int64_p]),
        "mcs_read_tallies": (i32, [vp, c_double_p, c_int64_p]),
        "mcs_read_tallies_part": (i32, [vp, i64, i64, c_double_p, c_int64_p]),
        "mcs_write_tallies_part": (i32, [vp, i64, i64, c_double_p]),
        "mcs_num_cus": (i32, [vp]),
        "mcs_write_tallies": (i32, [vp, c_double_p, c_int64_p]),
        "mcs_eval_fn": (i32, [vp, i32, i64, c_double_p, c_double_p, c_double_p]),
        "mcs_final_download": (i32, [vp, i64, c_int32_p, c_int32_p, c_int32_p, c_double_p, c_double_p]),
        "mcs_last_kernel_ms": (dbl, [vp]),
        "mcs_set_launch": (i32, [vp, i32, i32]),
        "mcs_set_tail_slicing": (i32, [vp, i32]),
        "mcs_last_launches": (i32, [vp]),
        "mcs_last_kernel": (i32, [vp]),
        "mcs_get_layout": (i32, [ct.POINTER(McsParams), c_int64_p]),
        "mcs_dndp_cr": (i32, [vp, ct.POINTER(McsConsumerIn), c_double_p, c_int64_p]),
        "mcs_thermo_calcs": (i32, [vp, ct.POINTER(McsConsumerIn), c_double_p, c_double_p, c_double_p]),
        "mcs_photon_synch": (i32, [vp, c_double_p, c_double_p, dbl, i32, dbl, dbl, c_double_p, c_double_p]),
        "mcs_run_pcuts_fused": (i32, [vp, i32, i32, c_int64_p, c_int64_p, c_int64_p, c_int64_p, c_double_p]),
        "mcs_run_pcuts_pipelined": (i32, [vp, i32, i32, c_int64_p, ct.c_int64, ct.c_int64, c_int64_p, c_int64_p, c_int64_p, c_double_p, c_int64_p]),
        "mcs_dndp_2d": (i32, [vp, ct.POINTER(McsConsumerIn), dbl, dbl, c_double_p]),
        "mcs_photon_ic": (i32, [vp, c_double_p, dbl, i32, i32, c_double_p, c_double_p, i32, dbl, dbl, dbl, c_double_p, c_double_p]),
        "mcs_photon_pion": (i32, [vp, c_double_p, c_double_p, dbl, dbl, c_double_p, dbl, i32, i32, dbl, dbl, c_double_p, c_double_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


EXPORTED_SYMBOLS = [
    "mcs_abi_version", "mcs_last_error", "mcs_create", "mcs_destroy", "mcs_sync", "mcs_bind_tallies",
    "mcs_tallies_f64_devptr", "mcs_tallies_i64_devptr", "mcs_set_grid", "mcs_set_cuts", "mcs_begin_iteration",
    "mcs_begin_species", "mcs_set_fluxes", "mcs_pop_upload", "mcs_pop_download", "mcs_saved_download",
    "mcs_pop_size", "mcs_init_pop", "mcs_init_pop_binned", "mcs_run_pcut", "mcs_new_pcut", "mcs_run_pcut_host", "mcs_read_tallies", "mcs_read_tallies_part", "mcs_num_cus",
    "mcs_write_tallies", "mcs_eval_fn", "mcs_final_download", "mcs_last_kernel_ms", "mcs_set_launch",
    "mcs_get_layout", "mcs_dndp_cr", "mcs_thermo_calcs",
    "mcs_run_pcut_strided", "mcs_run_pcut_indexed", "mcs_saved_gidx", "mcs_init_pop_binned_strided", "mcs_saved_export", "mcs_split_import", "mcs_set_debug_finals", "mcs_set_retro_cap",
    "mcs_set_tail_slicing", "mcs_last_launches", "mcs_last_kernel", "mcs_write_tallies_part", "mcs_photon_synch",
    "mcs_dndp_2d", "mcs_photon_ic", "mcs_run_pcuts_fused", "mcs_photon_pion", "mcs_run_pcuts_pipelined",
]
