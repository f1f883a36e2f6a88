"""Replay a golden case on a backend and compare with the committed fixture."""
import importlib.util
import os

import numpy as np

from conftest import ROOT, mcs

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)

CASES = list(make_golden.CASES)


def replay_and_compare(backend_factory, name, tally_rtol):
    """Per-particle data must match the fixture BIT FOR BIT; fp64 tallies up to summation
    order (tally_rtol = 0 for the single-threaded oracle itself)."""
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    prob, spec = make_golden.build_case(name)
    be = backend_factory(prob)
    got = make_golden.run_case(be, prob, spec, False)
    L = mcs.capi.Layout(prob.params)
    keys = [k for k in fx.files if k != "meta"]
    assert sorted(keys) == sorted(got.keys())
    for k in keys:
        a, b = got[k], fx[k]
        if "_tallies_" in k:
            continue
        assert a.shape == b.shape, k
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8)), \
            f"{name}:{k} differs in {(a != b).sum()} of {a.size} entries"
    for ion in range(1, len(prob.cfg.species) + 1):
        assert np.array_equal(got[f"ion{ion}_tallies_i64"], fx[f"ion{ion}_tallies_i64"]), f"{name}: int64 tallies, ion {ion}"
        ref = np.zeros(L.total)
        ref[fx[f"ion{ion}_tallies_idx"]] = fx[f"ion{ion}_tallies_val"]
        cur = np.zeros(L.total)
        cur[got[f"ion{ion}_tallies_idx"]] = got[f"ion{ion}_tallies_val"]
        assert int(got[f"ion{ion}_tallies_floor_count"][0]) == int(fx[f"ion{ion}_tallies_floor_count"][0])
        for tname in L.offsets:
            x, y = L.view(cur, tname), L.view(ref, tname)
            scale = float(np.max(np.abs(y)))
            if scale == 0:
                assert not np.any(x), f"{name}: tally {tname} should be empty"
                continue
            err = float(np.max(np.abs(x - y))) / scale
            assert err <= tally_rtol, f"{name}: tally {tname} (ion {ion}) off by {err:.3e}"
    if hasattr(be, "destroy"):
        be.destroy()
