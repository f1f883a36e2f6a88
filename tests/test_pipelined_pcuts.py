"""The pipelined pcut loop (include/mcs.h: mcs_run_pcuts_pipelined; DESIGN.md "Pipelined pcuts"): a pcut's long histories finish beside
the next pcut, which requires an order of the next population that does not depend on the schedule -- children of the saved particles
that are NOT long (fewer than long_draws random draws in the pcut) first, then those of the saved long ones, both in index order.  The
index keys a child's random stream (src/particle_loop.jl:35-40), so the order is part of the result: the oracle orders the same way
(orc_set_long_draws / orc_new_pcut_ordered) and the comparison stays what it is for the per-pcut loop -- per-pcut counts and every
integer tally bit for bit, the fp64 tallies to the order of the adds.

CPU: the oracle's order against a restatement from the saved arrays; long_draws = 0 is the reference's order; the driver's rule for
the pcuts in which long histories are told apart.  GPU: the pipelined loop against the oracle, where most pcuts have to wait for
their long histories (a small long_draws) and where most do not; with a long_draws nobody reaches, against the per-pcut loop."""
import numpy as np
import pytest

from conftest import mcs, make_problem, oracle_backend, assert_tallies_close


def _stats(r):
    return [(s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult) for s in r.stats]


def test_oracle_orders_long_histories_last():
    prob = make_problem(1500)
    ob = oracle_backend(prob, nthreads=8)
    from conftest import start_species
    start_species(ob, prob)
    B = 300
    ob.set_long_draws(B)
    seen_long = 0
    for i_pcut in range(1, 12):
        ns = ob.run_pcut(i_pcut, 0)
        if ns == 0:
            break
        saved, l_save = ob.get_saved()
        n_long = ob.count_long_saved()
        seen_long += n_long
        assert 0 <= n_long <= ns
        w = saved.weight.copy(); p = saved.pb_pf.copy(); x = saved.phi_rad.copy()
        i_mult = max(prob.cfg.N_PTS_PCUT // ns, 1)
        n_new = ob.new_pcut(i_mult)
        assert n_new == ns * i_mult
        pop = ob.get_population()
        # the parents in the order of the children: every i_mult-th child; a parent's copies are adjacent and carry weight / i_mult
        par_p, par_x, par_w = pop.pb_pf[::i_mult], pop.phi_rad[::i_mult], pop.weight[::i_mult] * i_mult
        idx = np.flatnonzero(l_save)
        # the first ns - n_long parents are saved particles in index order, the others too, and together they are all saved particles
        key = lambda P, X: np.array([P, X]).T
        a, b = key(par_p[:ns - n_long], par_x[:ns - n_long]), key(par_p[ns - n_long:], par_x[ns - n_long:])
        allp = key(p[idx], x[idx])
        # (parallel momentum, gyrophase) identify a saved particle: two continuous random numbers
        lookup = {tuple(v): j for j, v in enumerate(allp)}
        ja = np.array([lookup[tuple(v)] for v in a], dtype=int); jb = np.array([lookup[tuple(v)] for v in b], dtype=int)
        assert np.all(np.diff(ja) > 0) and np.all(np.diff(jb) > 0)
        assert sorted(np.concatenate([ja, jb]).tolist()) == list(range(ns))
        assert np.allclose(par_w, np.concatenate([w[idx][ja], w[idx][jb]]), rtol=1e-15)
        for c in range(1, i_mult):
            assert np.array_equal(pop.pb_pf[c::i_mult], par_p)
    assert seen_long > 0            # the case was not vacuous
    ob.destroy()


def test_long_draws_zero_is_the_reference_order():
    prob = make_problem(600)
    a = oracle_backend(prob, nthreads=8); b = oracle_backend(prob, nthreads=8)
    b.set_long_draws(0)
    ra = mcs.driver.run(prob, a, n_itrs=1)
    rb = mcs.driver.run(prob, b, n_itrs=1, long_draws=0)
    assert _stats(ra) == _stats(rb) and np.array_equal(ra.tallies_i64, rb.tallies_i64)
    # and a long_draws nobody reaches changes nothing either
    c = oracle_backend(prob, nthreads=8)
    rc = mcs.driver.run(prob, c, n_itrs=1, long_draws=2_000_000_000, long_imult_max=0)
    assert _stats(ra) == _stats(rc) and np.array_equal(ra.tallies_i64, rc.tallies_i64)
    # while a small one reorders the populations: other streams, statistically the same run
    d = oracle_backend(prob, nthreads=8)
    rd = mcs.driver.run(prob, d, n_itrs=1, long_draws=300, long_imult_max=0)
    assert _stats(ra) != _stats(rd)
    for be in (a, b, c, d):
        be.destroy()


def test_driver_tells_long_histories_apart_where_the_rule_says():
    """long_imult_max: long histories are told apart in pcut 1 and in pcuts whose predecessor split by at most that factor."""
    prob = make_problem(600)
    ob = oracle_backend(prob, nthreads=8)
    calls = []
    real = ob.set_long_draws
    ob.set_long_draws = lambda v: (calls.append(int(v)), real(v))[1]
    r = mcs.driver.run(prob, ob, n_itrs=1, long_draws=500, long_imult_max=3)
    per_pcut = calls[-len(r.stats):]               # (one call before the loop, then one per pcut)
    assert len(calls) == len(r.stats) + 1
    want = [500] + [500 if s.i_mult <= 3 else 0 for s in r.stats[:-1]]
    assert per_pcut == want and 0 in want and 500 in want[1:]
    with pytest.raises(ValueError):
        class _Comm:      # a stand-in for a 2-rank group: the order is single-rank
            enabled, rank, world = True, 0, 2
        mcs.driver.run(prob, ob, comm=_Comm(), n_itrs=1, long_draws=500)
    ob.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("N,B,imax", [(3000, 400, 0), (20000, 2000, 8), (60000, 6144, 8)])
def test_gpu_pipelined_loop_equals_the_oracle_in_the_same_order(N, B, imax):
    from conftest import hip_backend
    prob = make_problem(N)
    ob = oracle_backend(prob, nthreads=16)
    ro = mcs.driver.run(prob, ob, n_itrs=1, long_draws=B, long_imult_max=imax)
    hb = hip_backend(prob)
    rg = mcs.driver.run(prob, hb, n_itrs=1, long_draws=B, long_imult_max=imax)
    assert hb.last_kernel() == 11                          # the sliced form of the PLAIN kernel ran
    assert _stats(ro) == _stats(rg)
    assert np.array_equal(ro.tallies_i64, rg.tallies_i64)
    assert_tallies_close(mcs.capi.Layout(prob.params), ro.tallies_f64, rg.tallies_f64, rtol=1e-11)
    assert ro.steps_helix == rg.steps_helix and ro.steps_retro == rg.steps_retro
    ob.destroy(); hb.destroy()


@pytest.mark.gpu
def test_gpu_pipelined_loop_species_mix_with_energy_transfer_and_losses():
    """BASELINE config[4]'s species mix (p + He + e-, ion -> electron energy transfer, radiative losses): the sliced forms of the PLAIN_ETF
    and LOSSY kernels (kinds 13 and 12), every species against the oracle in the same population order"""
    from conftest import hip_backend
    me_mp = mcs.constants.ME / mcs.constants.MP
    sp = [mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1), mcs.inputs.Species(me_mp, -1.0, 1e6, 1.2)]
    prob = make_problem(4000, species=sp, energy_transfer_frac=0.1, radiation_losses=True)
    ob = oracle_backend(prob, nthreads=16)
    ro = mcs.driver.run(prob, ob, n_itrs=1, long_draws=1500, long_imult_max=8)
    hb = hip_backend(prob)
    kinds = []
    rg = mcs.driver.run(prob, hb, n_itrs=1, long_draws=1500, long_imult_max=8, on_species_end=lambda *a: kinds.append(hb.last_kernel()))
    assert kinds == [13, 13, 12]
    assert _stats(ro) == _stats(rg)
    assert np.array_equal(ro.tallies_i64, rg.tallies_i64)
    assert_tallies_close(mcs.capi.Layout(prob.params), ro.tallies_f64, rg.tallies_f64, rtol=1e-11)
    for (ia, ja, fa, ka), (ib, jb, fb, kb) in zip(ro.per_species, rg.per_species):
        assert (ia, ja) == (ib, jb) and np.array_equal(ka, kb)
        assert_tallies_close(mcs.capi.Layout(prob.params), fa, fb, rtol=1e-11)
    ob.destroy(); hb.destroy()


@pytest.mark.gpu
def test_gpu_pipelined_loop_over_iterations_and_tiny_populations():
    """two iterations on one context (buffers, streams and counters are reused), and populations smaller than a workgroup"""
    from conftest import hip_backend
    for N, B in ((2500, 600), (48, 200)):
        prob = make_problem(N, num_iterations=2)
        ob = oracle_backend(prob, nthreads=16)
        ro = mcs.driver.run(prob, ob, n_itrs=2, long_draws=B, long_imult_max=8)
        hb = hip_backend(prob)
        rg = mcs.driver.run(prob, hb, n_itrs=2, long_draws=B, long_imult_max=8)
        assert _stats(ro) == _stats(rg) and len({s.i_iter for s in rg.stats}) == 2
        assert np.array_equal(ro.tallies_i64, rg.tallies_i64)
        assert_tallies_close(mcs.capi.Layout(prob.params), ro.tallies_f64, rg.tallies_f64, rtol=1e-11)
        ob.destroy(); hb.destroy()


@pytest.mark.gpu
def test_gpu_pipelined_loop_without_long_histories_equals_the_per_pcut_loop():
    """A long_draws nobody reaches: no export, no late group -- the machinery alone (two sets of saved arrays, the masked streams, the
    split by status byte) must reproduce the per-pcut loop."""
    from conftest import hip_backend
    prob = make_problem(30000)
    a = hip_backend(prob); b = hip_backend(prob)
    ra = mcs.driver.run(prob, a, n_itrs=1, fused_pcuts=False)
    rb = mcs.driver.run(prob, b, n_itrs=1, long_draws=2_000_000_000, long_imult_max=0)
    assert _stats(ra) == _stats(rb)
    assert np.array_equal(ra.tallies_i64, rb.tallies_i64)
    assert_tallies_close(mcs.capi.Layout(prob.params), ra.tallies_f64, rb.tallies_f64, rtol=1e-11)
    # and the general kernel's sliced form (a species the PLAIN conditions exclude: electrons)
    sp = [mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(mcs.constants.ME / mcs.constants.MP, -1.0, 1e6, 1.0)]
    prob2 = make_problem(4000, species=sp)
    ob = oracle_backend(prob2, nthreads=16)
    ro = mcs.driver.run(prob2, ob, n_itrs=1, long_draws=1000, long_imult_max=8)
    c = hip_backend(prob2)
    rc = mcs.driver.run(prob2, c, n_itrs=1, long_draws=1000, long_imult_max=8)
    assert _stats(ro) == _stats(rc)
    assert np.array_equal(ro.tallies_i64, rc.tallies_i64)
    assert_tallies_close(mcs.capi.Layout(prob2.params), ro.tallies_f64, rc.tallies_f64, rtol=1e-11)
    for be in (a, b, ob, c):
        be.destroy()
