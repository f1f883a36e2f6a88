"""BASELINE configs at their full sizes on the GPU, through the C ABI.

config[1] (10^6 protons, one iteration, all 45 pcuts): against the committed reduction of ONE full
run of the CPU oracle (tests/golden/full_1e6.npz, made by tests/golden/make_golden_full.py): integer
tallies and population sizes equal, every binned spectrum within 1e-11 of its maximum (1e-10 for the flux vectors and
scalar accumulators that sum 1e7..1e8 terms per entry, see LONG_SUM_RTOL).

config[2]'s population (10^7 particles) and config[1]'s again: size-independent properties checked in
EVERY pcut the iteration reaches -- the late ones included, where the whole population is 10^5..10^7
replicas of one or two saved particles piling onto single histogram bins.
"""
import os

import numpy as np
import pytest

from conftest import ROOT, mcs, make_problem, oracle_backend, hip_backend, start_species, bits

pytestmark = pytest.mark.gpu
TALLY_RTOL = 1e-11
# Arrays whose entries are sums of 1e7..1e8 terms at this size: the three flux vectors (signed terms per zone: upstream-
# and downstream-going crossings cancel to a tenth of their gross sum) and the per-species / per-time-cut / per-momentum-
# bin accumulators (esc_flux: 1.6e7 IDENTICAL weights, whose rounding errors do not average out but add up along whatever
# order the adds take -- one serial sum in the oracle, per-block partial sums in LDS on the GPU).
# A-priori bound, not a fitted one: a sum of n non-negative fp64 terms differs between two summation orders by at most
# ~n * 2^-53 of the sum (each add rounds by <= half an ulp of the running total); n = 1.6e7 gives 1.8e-9, and for the flux
# vectors the cancellation multiplies it by ten.  The bound used, 1e-10, is TIGHTER than that worst case (rounding errors of
# unequal terms mostly average out: measured 1.5e-11 .. 2.7e-11); 1e-11 holds for every array that sums < 1e6 terms per
# entry.  A compensated reference sum would not tighten it: the GPU's own order-dependent rounding is the same size.
# (The reference rounds its fluxes to 13 digits for the same reason, src/iter_finalize.jl:46-54.)
LONG_SUM_RTOL = 1e-10
LONG_SUMS = ("pxx_flux", "pxz_flux", "energy_flux", "esc_flux", "px_esc_feb", "energy_esc_feb", "esc_energy_eff", "esc_num_eff",
             "weight_coupled", "spectra_coupled_val", "scalars")


def _load_reducer():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_full", os.path.join(ROOT, "tests", "golden", "make_golden_full.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.reduce_tallies


def test_config1_full_size_vs_oracle_fixture():
    fix = np.load(os.path.join(ROOT, "tests", "golden", "full_1e6.npz"))
    N = 1_000_000
    prob = make_problem(N)
    hb = hip_backend(prob)
    res = mcs.driver.run(prob, hb, None, n_itrs=1)
    hb.destroy()
    got = _load_reducer()(mcs.capi.Layout(prob.params), res.tallies_f64, res.tallies_i64, res.stats)
    assert np.array_equal(got["stats"], fix["stats"])                 # n_pts_use, n_saved, i_mult of all pcuts
    assert np.array_equal(got["tallies_i64"], fix["tallies_i64"])     # crossings per zone, exits by reason, steps, draws
    assert int(fix["stats"][:, 2].min()) == 0 and len(fix["stats"]) >= 30
    worst = ("", 0.0)
    for k in fix.files:
        if k in ("stats", "tallies_i64", "meta"):
            continue
        a, b = got[k], fix[k]
        assert a.shape == b.shape, k
        if k.endswith("_idx"):
            assert np.array_equal(a, b), k
            continue
        scale = float(np.max(np.abs(b))) if b.size else 0.0
        if scale == 0.0:
            assert not np.any(a), k
            continue
        err = float(np.max(np.abs(a - b))) / scale
        if err > worst[1]:
            worst = (k, err)
        tol = LONG_SUM_RTOL if k in LONG_SUMS else TALLY_RTOL
        assert err <= tol, f"{k}: max|gpu - oracle| / max|oracle| = {err:.3e}"
    print(f"config[1] at 1e6: {len(fix.files) - 3} binned arrays within {TALLY_RTOL}; worst {worst[0]} {worst[1]:.2e}; {fix['meta']}")


def test_long_sum_tolerance_is_the_gpu_add_order_noise(monkeypatch):
    """Where LONG_SUM_RTOL comes from, measured instead of asserted: the SAME iteration (same particles, bit for bit) twice on the
    GPU with two different add orders -- the default (16 tally replicas, per-block LDS staging of the fluxes) and one replica with
    a different number of blocks -- and a pairwise (numpy) re-summation of the long per-bin accumulators of the fixture's
    reducer.  The two GPU runs differ from each other in the long sums by as much as either differs from the oracle's serial
    sum: the bound is order noise of the GPU's own adds, not an error of the serial reference, and a compensated reference would
    not tighten it.  (History: round 2 set 1e-11 for everything, the run gpurun_out/r2_t2.log failed on energy_flux at 2.7e-11 and
    esc_flux at 1.5e-11; the a-priori bound n * 2^-53 for n = 1.6e7 adds is 1.8e-9.)"""
    fix = np.load(os.path.join(ROOT, "tests", "golden", "full_1e6.npz"))
    N = 1_000_000
    red = _load_reducer()
    runs = []
    for env in ({}, {"MCS_TALLY_REPLICAS_OFF": "1", "MCS_FUSED_PCUTS": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        prob = make_problem(N)
        hb = hip_backend(prob)
        if env:
            hb.set_launch(300, 256)          # another block -> particle mapping: other partial sums
        res = mcs.driver.run(prob, hb, None, n_itrs=1)
        hb.destroy()
        runs.append(red(mcs.capi.Layout(prob.params), res.tallies_f64, res.tallies_i64, res.stats))
    a, b = runs
    assert np.array_equal(a["tallies_i64"], b["tallies_i64"]) and np.array_equal(a["stats"], b["stats"])      # the same particles
    worst_gg = worst_go = 0.0
    for k in LONG_SUMS:
        if k not in fix.files:
            continue
        scale = float(np.max(np.abs(fix[k])))
        if scale == 0.0:
            continue
        gg = float(np.max(np.abs(a[k] - b[k]))) / scale
        go = max(float(np.max(np.abs(a[k] - fix[k]))), float(np.max(np.abs(b[k] - fix[k])))) / scale
        assert gg <= LONG_SUM_RTOL and go <= LONG_SUM_RTOL, (k, gg, go)
        worst_gg, worst_go = max(worst_gg, gg), max(worst_go, go)
    # the GPU disagrees with ITSELF by the same order as with the oracle (within a factor of a few either way), above 1e-12
    assert worst_gg > 1e-13 and worst_go > 1e-13
    assert worst_gg > worst_go / 20, (worst_gg, worst_go)
    print(f"long sums: GPU vs GPU (another add order) {worst_gg:.2e}, GPU vs oracle fixture {worst_go:.2e}, bound {LONG_SUM_RTOL}")


def _property_run(N, n_prefix=4096, prefix_pcuts=5, prob=None, i_iter=1):
    """One species through every pcut it reaches.  Per pcut: (i) every particle ends in exactly one way and the
    saved flags are the reason-0 particles; (ii) the counters' exits equal the particles that ended; (iii) the first
    n_prefix particles equal the oracle's bit for bit while the prefix stays aligned (pcuts 1-4 save everybody);
    (iv) the split population is i_mult copies of each saved particle, in order, with weight / i_mult.  At the end:
    weight is conserved through all splits, the upstream-escape tallies carry exactly the weight of the particles
    that escaped upstream (LDS-staged and wave-reduced tallies at full size), no zone search failed."""
    prob = make_problem(N) if prob is None else prob      # (a caller's problem: e.g. one whose profile an iteration has updated)
    hb = hip_backend(prob)
    start_species(hb, prob, i_iter)
    ng, IC = prob.n_grid, mcs.capi.IC
    pop = hb.get_population()
    w_in = float(pop.weight.sum())
    ob = oracle_backend(prob, nthreads=8)
    start_species(ob, prob, i_iter)
    ob.set_population(pop.slice(0, n_prefix))
    w_out = w_esc_up = 0.0
    n_done = n_up = 0
    reached = 0
    n_checked = []          # particles compared with the oracle in each of the first pcuts
    I_prev = hb.read_tallies()[1]
    for ip in range(1, len(prob.pcuts) + 1):
        n_use = pop.n
        ns = hb.run_pcut(ip, 0)
        reached = ip
        f = hb.finals()
        saved, l_save = hb.get_saved()
        assert int(l_save.sum()) == ns and np.array_equal(f["reason"] == 0, l_save == 1), f"pcut {ip}"
        assert f["reason"].min() >= 0 and f["reason"].max() <= 4
        ended = f["reason"] != 0
        w_out += float(pop.weight[ended].sum())
        up = f["reason"] == 2
        w_esc_up += float(pop.weight[up].sum()); n_up += int(up.sum())
        n_done += int(ended.sum())
        I = hb.read_tallies()[1]
        d = I - I_prev; I_prev = I
        assert sum(int(d[ng + IC[f"REASON{r}"]]) for r in range(1, 5)) == int(ended.sum()), f"pcut {ip}"
        assert int(d[ng + IC["REASON0"]]) == ns and int(ended.sum()) + ns == n_use
        assert int(d[ng + IC["STEPS_HELIX"]]) == int(np.minimum(f["helix"], 10000).astype(np.int64).sum())
        assert int(d[ng + IC["STEPS_RETRO"]]) == int(f["retro"].astype(np.int64).sum())
        nso = 0
        if ip <= prefix_pcuts and n_prefix > 0:
            nso = ob.run_pcut(ip, 0)
            fo = ob.finals()
            for k in fo:
                assert np.array_equal(bits(f[k][:n_prefix]), bits(fo[k])), f"pcut {ip}: prefix {k}"
            n_checked.append(n_prefix)
        if ns == 0:
            break
        im = max(N // ns, 1)
        if ip < prefix_pcuts and n_prefix > 0:
            # the children of the prefix's saved particles are the first nso * im particles of the next population, with the
            # same global indices (the split keeps the order): the oracle follows with ITS split of the prefix
            n_prefix = ob.new_pcut(im) if nso > 0 else 0
            assert n_prefix == nso * im
        assert hb.new_pcut(im) == ns * im
        pop = hb.get_population()
        src = np.flatnonzero(l_save)
        o = np.unique(np.concatenate([np.arange(0, pop.n, 7), np.arange(max(pop.n - 1000, 0), pop.n)]))   # a sample of the new indices
        par = src[o // im]
        for fld in pop.fields():
            want = getattr(saved, fld)[par]
            if fld == "weight":
                want = want / float(im)
            assert np.array_equal(bits(getattr(pop, fld)[o]), bits(want)), f"pcut {ip}: split field {fld} (i_mult {im}, {ns} parents)"
    assert reached >= 30, "the iteration should run into the late pcuts"
    assert len(n_checked) >= 3 and min(n_checked[:3]) >= 256, n_checked
    w_left = float(pop.weight.sum()) if ns else 0.0
    assert abs(w_out + w_left - w_in) < 1e-9 * w_in
    T, I = hb.read_tallies()
    L = mcs.capi.Layout(prob.params)
    assert int(I[ng + IC["ZONE_FAIL"]]) == 0 and int(I[ng + IC["RETRO_CAP"]]) == 0
    assert sum(int(I[ng + IC[f"REASON{r}"]]) for r in range(1, 5)) == n_done
    assert int(I[ng + IC["REASON2"]]) == n_up
    # reason-2 weight reaches esc_flux (LDS scalar staging) and esc_num_eff (LDS per-bin staging) exactly once
    assert abs(float(L.view(T, "esc_flux")[0]) - w_esc_up) <= 1e-10 * max(w_esc_up, 1e-300)
    assert abs(float(L.view(T, "esc_num_eff").sum()) - w_esc_up) <= 1e-10 * max(w_esc_up, 1e-300)
    hb.destroy(); ob.destroy()
    return reached, n_done


def test_config1_properties_every_pcut_1e6():
    reached, n_done = _property_run(1_000_000)
    print(f"1e6 protons: {reached} pcuts reached, {n_done} exits")


def test_config2_population_properties_every_pcut_1e7():
    reached, n_done = _property_run(10_000_000)
    print(f"1e7 protons: {reached} pcuts reached, {n_done} exits")


def test_config3_per_gpu_size_properties_5e7():
    """BASELINE config[3] (1e8 particles over 2 / 4 / 8 GPUs) at its LARGEST per-GPU size, 5e7 particles on one GPU: the same
    per-pcut properties (every particle ends once, counters equal exits, splits are i_mult copies in order, weight conserved,
    the first 4096 particles equal the oracle's bit for bit) with 10 GB of population buffers resident.  The multi-GPU run
    itself cannot be tested on a one-GPU box; what a rank of it computes is this."""
    reached, n_done = _property_run(50_000_000)
    print(f"5e7 protons: {reached} pcuts reached, {n_done} exits")


def test_config2_1e7_second_iteration_on_the_updated_profile():
    """BASELINE config[2] at its own size WITH its own loop: iteration 1 of 10^7 protons through driver.run with smoothing on
    (K4 consumers on the device, iter_finalize, smooth_grid_par, new tables through mcs_set_grid), then iteration 2 -- on the
    modified profile every zone crossing takes the frame transform -- through the per-pcut property checks of _property_run,
    the first 4096 particles against the oracle on the same updated tables."""
    N = 10_000_000
    itf = mcs.iter_finalize
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=2)
    prob = mcs.inputs.build_problem(cfg)
    u_before = prob.ux.copy()
    hb = hip_backend(prob)
    res = mcs.driver.run(prob, hb, None, n_itrs=1, smoothing=itf.SmoothingConfig(smooth_shocks=True), species_tallies="light")
    hb.destroy()
    (_, fin, ion), = res.iter_finals
    P, n = prob.params, prob.n_grid
    assert fin.profile_changed and not np.array_equal(prob.ux, u_before)
    u = prob.ux[1:n + 1]
    assert np.all(np.diff(u) <= 1e-12 * P.u0) and u.max() <= P.u0 * (1 + 1e-12) and u.min() >= P.u2 * (1 - 1e-12)
    assert prob.ux[P.i_shock - 3] < 0.999 * P.u0                      # a precursor has formed
    assert 4.0 / 3.0 - 0.02 < fin.Gamma_downstream < 5.0 / 3.0 + 0.02
    assert np.all(np.isfinite(ion.P_psd_par)) and ion.P_psd_par.max() > 0
    ng, IC = n, mcs.capi.IC
    I = res.tallies_i64
    assert int(I[ng + IC["ZONE_FAIL"]]) == 0 and int(I[ng + IC["RETRO_CAP"]]) == 0
    st = [(s.n_pts_use, s.n_saved, s.i_mult) for s in res.stats]
    assert all(b[0] == a[1] * a[2] for a, b in zip(st, st[1:])) and len(st) >= 30      # every population is the split of the previous one
    assert sum(int(I[ng + IC[f"REASON{r}"]]) for r in range(1, 5)) == sum(a[0] - a[1] for a in st)
    reached, n_done = _property_run(N, prob=prob, i_iter=2)
    print(f"config[2], 1e7 protons: iteration 1 {res.steps_helix + res.steps_retro} steps, profile updated "
          f"(u_x at shock-3: {prob.ux[P.i_shock - 3] / P.u0:.4f} u0); iteration 2 on it: {reached} pcuts, {n_done} exits")
