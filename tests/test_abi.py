"""The C-ABI library: loads on a machine without a GPU, exports every symbol that
include/mcs.h declares, agrees with the ctypes mirror on struct layout, and REFUSES to
compute without a device (no CPU fallback)."""
import ctypes as ct
import os
import re

import numpy as np
import pytest

from conftest import ROOT, mcs, make_problem


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mcs.h")).read()
    src = src[src.index("typedef struct mcs_ctx mcs_ctx;"):]
    return sorted(set(re.findall(r"\b(mcs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = mcs.capi.load_library()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"libmcs_hip.so does not export {s}"
    assert sorted(mcs.capi.EXPORTED_SYMBOLS) == syms
    assert lib.mcs_abi_version() == mcs.capi.MCS_ABI_VERSION


def test_layout_matches_header():
    lib = mcs.capi.load_library()
    prob = make_problem(64)
    L = mcs.capi.Layout(prob.params)
    out = (ct.c_int64 * 24)()
    assert lib.mcs_get_layout(ct.byref(prob.params), out) == 0
    names = ["psd", "therm_sf", "therm_pf", "esc_psd_up", "esc_psd_down", "pxx_flux", "pxz_flux", "energy_flux", "esc_flux",
             "px_esc_feb", "energy_esc_feb", "esc_energy_eff", "esc_num_eff", "weight_coupled", "spectra_coupled",
             "spectra_sf", "spectra_pf", "energy_transfer_pool", "energy_recv_pool", "scalars"]
    for i, n in enumerate(names):
        assert out[i] == L.offsets[n], n
    assert out[len(names)] == L.total
    assert out[len(names) + 1] == prob.params.num_psd_mom_bins + 2
    assert out[len(names) + 2] == (prob.params.num_psd_mom_bins + 2) * (prob.params.num_psd_tht_bins + 2)


def test_no_cpu_fallback():
    """Without a GPU mcs_create must fail with a message; with one it must succeed."""
    import torch
    lib = mcs.capi.load_library()
    prob = make_problem(64)
    h = ct.c_void_p(None)
    rc = lib.mcs_create(ct.byref(prob.params), 0, None, ct.byref(h))
    if torch.cuda.is_available():
        assert rc == 0
        lib.mcs_destroy(h)
    else:
        assert rc != 0 and len(lib.mcs_last_error()) > 0


def test_create_rejects_unsupported_modes():
    """The reference's error() sites become error returns (scattering.jl:52-53, prob_return.jl:134)."""
    lib = mcs.capi.load_library()
    for field in ("use_custom_frg",):
        prob = make_problem(64)
        setattr(prob.params, field, 1)
        h = ct.c_void_p(None)
        assert lib.mcs_create(ct.byref(prob.params), 0, None, ct.byref(h)) != 0
        assert b"custom f(r_g)" in lib.mcs_last_error()
    prob = make_problem(64)
    prob.params.do_retro = 0
    h = ct.c_void_p(None)
    assert lib.mcs_create(ct.byref(prob.params), 0, None, ct.byref(h)) != 0
    assert b"analytical PRP" in lib.mcs_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setenv("MCS_HIP_LIB", "libdoes_not_exist.so")
    monkeypatch.setattr(mcs.capi, "_LIB", None)
    with pytest.raises(mcs.capi.MissingNativeLibrary):
        mcs.capi.load_library()


def test_kernel_register_allocation_is_as_documented():
    """The build keeps the compiler's kernel-resource-usage remarks of mcs_transport.hip; the transport kernels must stay
    inside the documented bounds (tools/check_resources.py: occupancy, VGPR spills, scratch, LDS) -- a build that spilled
    differently once miscompiled the l_save store (csrc/Makefile)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_resources.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mcs_k_transport_plain" in r.stdout and "2 waves/SIMD" in r.stdout
