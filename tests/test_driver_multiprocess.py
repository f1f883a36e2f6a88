"""The multi-GPU path of the host driver, rehearsed on CPU: world_size 2 and 3 over gloo
with the CPU oracle as backend must reproduce the single-process run (particle shards with
global RNG keys, all-gather of n_saved, the two forms of the cross-rank new_pcut, one
sum-all-reduce of the tallies)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, mcs, orc


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _launch(world, out, N, npc, extra=()):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), out, "oracle", str(N), str(npc), *extra],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    try:
        for p in procs:
            o, _ = p.communicate(timeout=600)
            assert p.returncode == 0, o.decode()[-3000:]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def _single(N, npc, two=False, n_itrs=2):
    kw = {}
    if two:
        kw = dict(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1)], energy_transfer_frac=0.1)
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=n_itrs, **kw)
    prob = mcs.inputs.build_problem(cfg)
    be = orc.OracleBackend(mcs.capi, "det", 1)
    be.create(prob)
    return prob, mcs.driver.run(prob, be, None, n_itrs=n_itrs, max_pcuts=npc)


def _check_equal(prob, got, ref, rtol=1e-12):
    stats_ref = np.array([[s.i_iter, s.i_ion, s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult] for s in ref.stats])
    assert np.array_equal(got["stats"], stats_ref)              # same populations at every pcut
    assert np.array_equal(got["i"], ref.tallies_i64)            # same steps, crossings, exits
    from conftest import assert_tallies_close
    assert_tallies_close(mcs.capi.Layout(prob.params), got["f"], ref.tallies_f64, rtol=rtol)


@pytest.mark.parametrize("world,two,gather_max", [(2, False, 1 << 17), (3, False, 1 << 17), (2, True, 1 << 17), (3, False, 10), (2, True, 10)])
def test_sharded_run_equals_single_process(tmp_path, world, two, gather_max):
    """gather_max = 2^17: every pcut takes the gather split; 10: the local split with index lists (run_pcut_indexed),
    several in a row, two iterations, two species."""
    N, npc = 150, 7
    out = str(tmp_path / f"w{world}.npz")
    _launch(world, out, N, npc, (("2",) if two else ()) + (f"gather_max={gather_max}", "skew_max=3.0"))
    prob, ref = _single(N, npc, two)
    got = np.load(out)
    _check_equal(prob, got, ref)
    if gather_max == 10:      # (the first pcuts, where everybody is saved and i_mult = 1, exchange nothing: "identity")
        assert ((got["split"] == "local") | (got["split"] == "identity")).sum() >= 4 and (got["split"] == "local").sum() >= 1


@pytest.mark.parametrize("gather_max", [1 << 17, 40])
def test_late_pcuts_are_balanced_across_ranks(tmp_path, gather_max):
    """The whole iteration on 3 ranks.  In the late pcuts a handful of particles is saved and each is
    replicated hundreds of times (src/cuts.jl:42); every rank must still carry a third of that
    population (max/mean <= 1.01), and the run must equal the single-process one.  With
    gather_max = 40 the early pcuts (many saved, evenly spread) take the local split and the
    switch local -> gather is exercised; with the default every pcut gathers."""
    world, N = 3, 1500
    out = str(tmp_path / "late.npz")
    _launch(world, out, N, 45, ("itrs=1", f"gather_max={gather_max}"))
    prob, ref = _single(N, 45, n_itrs=1)
    got = np.load(out)
    _check_equal(prob, got, ref)
    st, n_use_max, split = got["stats"], got["n_use_max"], got["split"]
    n_use, n_saved = st[:, 3], st[:, 4]
    assert n_saved.min() == 0 and ((n_saved > 0) & (n_saved <= 5)).any()      # the iteration ran out; a late pcut had <= 5 parents
    if gather_max == 40:
        assert (split == "local").any() and (split == "gather").any()
    for j in range(1, len(st)):
        if split[j - 1] == "gather":            # population j was dealt out r, r+W, r+2W, ...
            assert n_use_max[j] == -(-n_use[j] // world)
            if n_use[j] >= 1000:
                assert n_use_max[j] * world <= 1.01 * n_use[j]
        else:                                   # local split: within the accepted skew
            assert n_use_max[j] * world <= 1.1 * 1.001 * n_use[j] + world


def test_world_8_whole_iteration(tmp_path):
    """The shape of BASELINE configs [3] / [4] -- 8 ranks -- rehearsed over gloo with the oracle as backend: a whole iteration
    (all pcuts, until nobody is saved) must equal the single-process run; after every gathered split the ranks hold the same
    number of particles +- 1; the first pcuts, where every particle is saved and i_mult = 1, exchange nothing ("identity").
    (N = 4e4: about a minute with 8 single-threaded ranks + one 8-thread reference run; the 8-GPU box runs 10^6 per rank.)"""
    world, N = 8, 40000
    out = str(tmp_path / "w8.npz")
    _launch(world, out, N, 45, ("itrs=1", "gather_max=400"))
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=1)
    prob = mcs.inputs.build_problem(cfg)
    be = orc.OracleBackend(mcs.capi, "det", 8); be.create(prob)
    ref = mcs.driver.run(prob, be, None, n_itrs=1, max_pcuts=45)
    got = np.load(out)
    _check_equal(prob, got, ref, rtol=1e-11)      # (sums of 1e5..1e6 terms in two different orders: 8 partial sums vs 8 threads)
    st, n_use_max, split = got["stats"], got["n_use_max"], got["split"]
    n_use, n_saved = st[:, 3], st[:, 4]
    assert n_saved.min() == 0 and len(st) >= 25
    assert (split == "identity").sum() >= 3 and (split == "local").any() and (split == "gather").any()
    for j in range(1, len(st)):
        if split[j - 1] == "gather":
            assert n_use_max[j] == -(-n_use[j] // world)
            if n_use[j] >= 1000:
                assert n_use_max[j] * world <= 1.01 * n_use[j]
        elif split[j - 1] == "identity":
            assert n_use_max[j] == n_use_max[j - 1]
        else:
            assert n_use_max[j] * world <= 1.1 * 1.001 * n_use[j] + world


def test_oracle_strided_init_and_indexed_run():
    """The oracle twins of mcs_init_pop_binned_strided / mcs_run_pcut_indexed / mcs_saved_gidx: a particle's history
    depends on its global index only (the GPU form of this test is test_strided_init_and_indexed_run)."""
    from conftest import make_problem, oracle_backend, start_species, assert_pop_equal
    N, W = 600, 3
    prob = make_problem(N)
    ob = oracle_backend(prob)
    inj = start_species(ob, prob)
    N = inj.n_pts_use
    full = ob.get_population()
    ob.run_pcut(1, 0)
    fin, (sav, lsave) = ob.finals(), ob.get_saved()
    lsave = lsave.copy()
    for r in range(W):
        ob.init_pop(inj, r, (N - r + W - 1) // W, N, W)
        assert_pop_equal(ob.get_population(), full.take(np.arange(r, N, W)), f"strided init, rank {r}")
    sel = np.sort(np.random.default_rng(5).choice(N, size=177, replace=False))
    ob.set_population(full.take(sel))
    ns = ob.run_pcut_indexed(1, sel.astype(np.int64))
    fa = ob.finals()
    for k in fa:
        assert np.array_equal(fa[k], fin[k][sel]), k
    assert ns == int(lsave[sel].sum())
    assert np.array_equal(ob.saved_gidx().numpy(), sel[lsave[sel] == 1])
    ob.destroy()


def test_overlapped_iterations_equal_sequential_on_the_oracle():
    """driver.run_overlapped with the CPU oracle as backend (two contexts, two host threads): every iteration is the one the
    sequential run computes and the merged final state equals the sequential one (the GPU form of this test is
    test_overlapped_iterations_equal_sequential)."""
    from conftest import assert_tallies_close
    N, n_itrs, npc = 300, 3, 9
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=n_itrs)
    prob = mcs.inputs.build_problem(cfg)
    be = orc.OracleBackend(mcs.capi, "det", 1); be.create(prob)
    seq = mcs.driver.run(prob, be, None, n_itrs=n_itrs, max_pcuts=npc, finalize=True)
    be.destroy()
    bes = [orc.OracleBackend(mcs.capi, "det", 1), orc.OracleBackend(mcs.capi, "det", 1)]
    for b in bes:
        b.create(prob)
    ovl = mcs.driver.run_overlapped(prob, bes, n_itrs=n_itrs, max_pcuts=npc)
    key = lambda r: [(s.i_iter, s.i_ion, s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult) for s in r.stats]
    assert key(ovl) == key(seq)
    assert np.array_equal(ovl.tallies_i64, seq.tallies_i64)
    assert [s for _, _, s in ovl.local_steps] == [s for _, _, s in seq.local_steps]
    assert_tallies_close(mcs.capi.Layout(prob.params), ovl.tallies_f64, seq.tallies_f64, rtol=1e-12)
    for (ia, fa, _), (ib, fb, _) in zip(ovl.iter_finals, seq.iter_finals):
        assert ia == ib and abs(fa.Gamma_downstream / fb.Gamma_downstream - 1) < 1e-12
    for b in bes:
        b.destroy()
