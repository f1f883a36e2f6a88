#!/usr/bin/env python3
"""Generate tests/golden/full_1e6.npz: BASELINE config[1] at FULL size -- 10^6 protons, single
unmodified gamma0 = 5 shock, all 45 stock pcuts, one iteration -- run once on the CPU oracle
(det math, OpenMP over particles; ~1.4e10 steps, about ten minutes on 8 cores), reduced to
binned spectra small enough to commit:

  *_mom[zone][k]   = sum over angle bins of psd / therm_sf / therm_pf   (dN(p) per zone, shock frame:
                     what get_dNdp_cr sums, src/particle_counter.jl:81-85)
  *_tht[zone][j]   = sum over momentum bins                             (angular distribution per zone)
  esc_psd_*_mom/_tht  the two marginals of the escape spectra
  every small tally array in full (fluxes, escape scalars and efficiencies, coupled weights and
  spectra, pools, scalars), the int64 tallies (crossings per zone, exits by reason, step and draw
  counts) and the per-pcut population sizes (n_pts_use, n_saved, i_mult).

Like every fixture here it is the ORACLE's output (the reference holds no vectors and cannot run):
oracle parity, not reference parity.  The GPU test (tests/test_gpu_full_size.py) runs the same
iteration through the C ABI and requires the integers to be equal and every fp64 array to agree
within 1e-11 of its maximum (order of the atomic adds; the threaded oracle has the same freedom).

    python tests/golden/make_golden_full.py [N] [threads]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _mcs_loader
mcs = _mcs_loader.load()

BIG = ("psd", "therm_sf", "therm_pf")
ESC = ("esc_psd_up", "esc_psd_down")


def reduce_tallies(L, T, I, stats):
    """The committed reduction; the GPU test applies the same function to the HIP tallies."""
    out = {}
    for name in BIG:
        a = L.view(T, name)                       # [zone][tht][mom]
        out[name + "_mom"] = a.sum(axis=1)
        out[name + "_tht"] = a.sum(axis=2)
    for name in ESC:
        a = L.view(T, name)                       # [tht][mom]
        out[name + "_mom"] = a.sum(axis=0)
        out[name + "_tht"] = a.sum(axis=1)
    for name in L.offsets:
        if name in BIG or name in ESC:
            continue
        a = L.view(T, name)
        if a.size > 4096:                         # spectra_coupled, spectra_sf/pf: sparse
            nz = np.flatnonzero(a.ravel())
            out[name + "_idx"] = nz.astype(np.int64)
            out[name + "_val"] = a.ravel()[nz].copy()
        else:
            out[name] = a.copy()
    out["tallies_i64"] = I.copy()
    out["stats"] = np.array([[s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult] for s in stats], dtype=np.int64)
    return out


def main():
    import orc
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    orc.build()
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N)
    prob = mcs.inputs.build_problem(cfg)
    be = orc.OracleBackend(mcs.capi, "det", nthreads=threads)
    be.create(prob)
    t0 = time.perf_counter()
    res = mcs.driver.run(prob, be, None, n_itrs=1, verbose=True)
    dt = time.perf_counter() - t0
    L = mcs.capi.Layout(prob.params)
    out = reduce_tallies(L, res.tallies_f64, res.tallies_i64, res.stats)
    out["meta"] = np.array(f"N={N} protons, 45 stock pcuts, 1 iteration, oracle det math, {threads} threads, "
                           f"{res.steps_helix + res.steps_retro} steps in {dt:.0f} s")
    name = "full_1e6.npz" if N == 1_000_000 else f"full_{N}.npz"
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB; {out['meta']}")


if __name__ == "__main__":
    main()
