"""MI355X-native per-particle transport path of MonteCarloScattering.jl.

Only what the hot path needs lives here:
  csrc/        HIP kernels (gfx950) + the C ABI declared in include/mcs.h
  capi.py      ctypes mirror of the ABI
  inputs.py    host-side input builder (grid, profile, PSD bins, injection)
  driver.py    the iteration/species/pcut nest around the batched kernel
  hip_backend.py  the one and only compute backend (no CPU fallback)
  consumers.py  host tables + call order of the tally consumers (ion_finalize: dN/dp, pressures)
  iter_finalize.py  iter_finalize + smooth_grid_par: the profile update between iterations (BASELINE config[2])

The directory name contains a dot, so it is loaded through `_mcs_loader.load()`
(repo root) under the module name `mcs_amd`.
"""
from . import constants, capi, inputs, driver, consumers, iter_finalize  # noqa: F401

__all__ = ["constants", "capi", "inputs", "driver", "consumers", "iter_finalize"]
