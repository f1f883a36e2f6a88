"""Parity of the gfx950 path with the CPU oracle, THROUGH THE C ABI (libmcs_hip.so), on a
real MI355X.  Bar: bit-exact for everything per particle (end state, saved population,
split population, RNG consumption, step counts, integer tallies); fp64 tallies agree up to
the order of the atomic adds: |gpu - oracle| <= 1e-11 * max|oracle| per tally array."""
import ctypes as ct
import sys

import numpy as np
import pytest

from conftest import (mcs, orc, make_problem, oracle_backend, hip_backend, start_species, bits, assert_pop_equal,
                      assert_tallies_close)
from golden_common import CASES, replay_and_compare

pytestmark = pytest.mark.gpu
TALLY_RTOL = 1e-11


def test_native_library_is_the_compute_path():
    import torch
    assert torch.cuda.is_available()
    lib = mcs.capi.load_library()
    assert "libmcs_hip" in lib._name


def test_math_and_rng_bit_parity():
    prob = make_problem(64)
    hb, ob = hip_backend(prob), oracle_backend(prob)
    rng = np.random.default_rng(0)
    n = 1_000_000
    dp = ct.POINTER(ct.c_double)

    def oev(fn, a, b=None):
        a = np.ascontiguousarray(a); b = a if b is None else np.ascontiguousarray(b); out = np.zeros_like(a)
        ob.lib.orc_eval_fn(mcs.capi.FN[fn], len(a), a.ctypes.data_as(dp), b.ctypes.data_as(dp), out.ctypes.data_as(dp))
        return out
    cases = {"sin": (rng.uniform(-10, 10, n), None), "cos": (rng.uniform(-10, 10, n), None),
             "asin": (rng.uniform(-1, 1, n), None), "acos": (rng.uniform(-1, 1, n), None),
             "atan2": (rng.normal(size=n), rng.normal(size=n)), "log10": (10 ** rng.uniform(-30, 30, n), None),
             "mod2pi": (rng.uniform(-20, 20, n), None), "sqrt": (10 ** rng.uniform(-40, 40, n), None),
             "div": (rng.normal(size=n), rng.normal(size=n)), "hypot1": (10 ** rng.uniform(-6, 10, n), None),
             "uniform": (np.floor(rng.uniform(0, 2 ** 40, n)), np.floor(rng.uniform(0, 30000, n)))}
    for fn, (a, b) in cases.items():
        g, o = hb.eval_fn(fn, a, b), oev(fn, a, b)
        assert np.array_equal(bits(g), bits(o)), f"{fn}: {(g != o).sum()} of {n} results differ from the oracle"
    hb.destroy()


@pytest.mark.parametrize("name", CASES)
def test_golden_vectors_on_gpu(name):
    replay_and_compare(hip_backend, name, tally_rtol=TALLY_RTOL)


def _lockstep(prob, N, n_pcuts, check_split=True, setup=None):
    hb, ob = hip_backend(prob), oracle_backend(prob, nthreads=8)
    if setup is not None:
        setup(hb); setup(ob)
    start_species(hb, prob); start_species(ob, prob)
    assert_pop_equal(hb.get_population(), ob.get_population(), "init_pop (K3)")
    for ip in range(1, n_pcuts + 1):
        nsg, nso = hb.run_pcut(ip, 0), ob.run_pcut(ip, 0)
        assert nsg == nso
        fg, fo = hb.finals(), ob.finals()
        for k in fg:
            assert np.array_equal(bits(fg[k]), bits(fo[k])), f"pcut {ip}: final {k} differs for {(fg[k] != fo[k]).sum()} particles"
        sg, lg = hb.get_saved(); so, lo = ob.get_saved()
        assert np.array_equal(lg, lo)
        assert_pop_equal(sg, so, f"pcut {ip}: saved arrays")
        if nso == 0:
            break
        im = max(N // nso, 1)
        assert hb.new_pcut(im) == ob.new_pcut(im)
        if check_split:
            assert_pop_equal(hb.get_population(), ob.get_population(), f"pcut {ip}: new_pcut (K2)")
    Tg, Ig = hb.read_tallies(); To, Io = ob.read_tallies()
    assert np.array_equal(Ig, Io)
    assert_tallies_close(hb.layout, Tg, To, TALLY_RTOL)
    hb.destroy()
    return Io


@pytest.mark.parametrize("budget", [1, 7, 60])
def test_sliced_tail_is_bit_identical(budget):
    """mcs_set_tail_slicing: waves export their live particles `budget` trips after the queue ran dry and later launches
    resume them, spread over the chip's waves.  A history must not depend on how often it was suspended: end states,
    saved arrays, split populations, integer tallies equal the oracle's bit for bit, through nine pcuts."""
    N = 20000
    prob = make_problem(N)
    launches = []

    def setup(be):
        if hasattr(be, "set_tail_slicing"):
            be.set_tail_slicing(budget)
            orig = be.run_pcut
            def run(ip, off, _o=orig, _b=be):
                r = _o(ip, off); launches.append(_b.last_launches()); return r
            be.run_pcut = run
    _lockstep(prob, N, 9, setup=setup)
    assert max(launches) > 1, launches            # the runs really were sliced


def test_sliced_tail_general_kernel_mixed_species():
    """The same through the general kernel: electrons with radiative losses, energy transfer, an oblique field."""
    from golden_common import replay_and_compare as rc

    def factory(prob):
        be = hip_backend(prob)
        be.set_tail_slicing(2)
        return be
    for name in ("mixed_n96", "electrons_crafted_n64"):
        rc(factory, name, tally_rtol=TALLY_RTOL)


def test_tcut_print_rewrite_on_device_buffers():
    """driver.run(tcut_print=True): the per-iteration rewrite of weight_coupled / spectra_coupled (src/io.jl:28-45) goes back
    into the device buffer (mcs_write_tallies_part), so iteration 2 accumulates on top of the normalised spectra -- GPU and
    oracle through the same driver, two iterations."""
    N, npc = 4000, 12
    pg, po = make_problem(N, num_iterations=2), make_problem(N, num_iterations=2)
    hb, ob = hip_backend(pg), oracle_backend(po, nthreads=8)
    rg = mcs.driver.run(pg, hb, None, n_itrs=2, max_pcuts=npc, tcut_print=True)
    ro = mcs.driver.run(po, ob, None, n_itrs=2, max_pcuts=npc, tcut_print=True)
    assert np.array_equal(rg.tallies_i64, ro.tallies_i64)
    assert_tallies_close(hb.layout, rg.tallies_f64, ro.tallies_f64, TALLY_RTOL)
    sc = hb.layout.view(rg.tallies_f64, "spectra_coupled")[0, :len(pg.tcuts)]
    assert sc.sum() > 1.5 and np.all(sc[:, :pg.params.num_psd_mom_bins + 1] >= 1e-99)      # normalised spectra, floored
    Tg, _ = hb.read_tallies()
    assert np.array_equal(hb.layout.view(Tg, "spectra_coupled"), hb.layout.view(rg.tallies_f64, "spectra_coupled"))   # the device holds it
    hb.destroy()


def test_retro_walk_cap():
    """The reference's retro_time loop has no bound (src/prob_return.jl:257); here one walk ends after
    MCS_RETRO_CAP inner steps with i_reason 3 and a counter.  With the cap lowered to 4 steps many walks
    reach it: particles, step counts and the RETRO_CAP counter must equal the oracle's."""
    N = 3000
    prob = make_problem(N)
    I = _lockstep(prob, N, 9, setup=lambda be: be.set_retro_cap(4))
    assert I[prob.n_grid + mcs.capi.IC["RETRO_CAP"]] > 50
    I = _lockstep(prob, N, 7)                       # default cap: never reached
    assert I[prob.n_grid + mcs.capi.IC["RETRO_CAP"]] == 0


def test_non_finite_particles_are_refused():
    """A NaN position never satisfies an exit test: the reference would loop forever, the GPU would hold the
    lease until the caps.  The ABI refuses such input (mcs_pop_upload, mcs_run_pcut_host)."""
    prob = make_problem(64)
    hb, ob = hip_backend(prob), oracle_backend(prob)
    start_species(hb, prob); start_species(ob, prob)
    good = ob.get_population()
    for field, val in (("x_PT_cm", np.nan), ("prp_x_cm", np.inf), ("pb_pf", np.nan), ("acctime_sec", np.inf),
                       ("phi_rad", np.nan), ("weight", np.nan), ("ptot_pf", np.inf), ("xn_per", 0.0), ("xn_per", np.nan)):
        bad = good.slice(0, 8)
        getattr(bad, field)[3] = val
        with pytest.raises(RuntimeError, match="must be finite"):
            hb.set_population(bad)
        with pytest.raises(RuntimeError, match="must be finite"):
            hb.run_pcut_host(1, bad, 0)
    hb.set_population(good.slice(0, 8))             # the context is still usable
    assert hb.run_pcut(1, 0) == 8
    hb.destroy()


def test_saved_export_and_split_import():
    """The multi-GPU form of new_pcut (mcs_saved_export / mcs_split_import) against its numpy twin, and
    against the one-GPU split: the union over three ranks' strided slices is mcs_new_pcut's population."""
    import torch
    N = 5000
    prob = make_problem(N)
    hb, ob = hip_backend(prob), oracle_backend(prob, nthreads=8)
    start_species(hb, prob); start_species(ob, prob)
    for ip in range(1, 6):
        ns = hb.run_pcut(ip, 0); assert ns == ob.run_pcut(ip, 0)
        if ip < 5:
            hb.new_pcut(1); ob.new_pcut(1)
    assert 0 < ns < N
    im = N // ns
    # a strided shard of the same population: its export carries the global indices first + k * stride
    gi, f64, meta = hb.export_saved(ns + 7)
    go, fo, mo = ob.export_saved(ns + 7)
    assert torch.equal(gi.cpu(), go) and torch.equal(meta.cpu(), mo)
    assert np.array_equal(bits(f64.cpu().numpy()), bits(fo.numpy()))
    pieces = []
    for r in range(3):
        n_loc = (ns * im - r + 2) // 3
        hb.import_split(f64, meta, ns, im, r, 3, n_loc)
        ob.import_split(fo, mo, ns, im, r, 3, n_loc)
        assert_pop_equal(hb.get_population(), ob.get_population(), f"split_import, rank {r} of 3")
        pieces.append(hb.get_population())
    # argument checks: a slice that reaches past the split population
    with pytest.raises(RuntimeError, match="reaches past"):
        hb.import_split(f64, meta, ns, im, 0, 1, ns * im + 1)
    # the one-GPU split, interleaved, equals the three slices
    hb2 = hip_backend(prob)
    start_species(hb2, prob)
    for ip in range(1, 6):
        hb2.run_pcut(ip, 0)
        hb2.new_pcut(1 if ip < 5 else im)
    full = hb2.get_population()
    for r in range(3):
        assert_pop_equal(pieces[r], full.take(np.arange(r, full.n, 3)), f"rank {r}: strided slice of the one-GPU split")
    # a strided run: same particles as the contiguous run of the full population
    hb.import_split(f64, meta, ns, im, 1, 3, (ns * im - 1 + 2) // 3)
    n1 = hb.run_pcut(6, 1, 3)
    hb2.run_pcut(6, 0)
    fa, fb = hb.finals(), hb2.finals()
    for k in fa:
        assert np.array_equal(bits(fa[k]), bits(fb[k][1::3])), k
    assert n1 == int((fb["reason"][1::3] == 0).sum())
    hb.destroy(); hb2.destroy()


def test_strided_init_and_indexed_run_wave_specialised_kernel(monkeypatch):
    """the same entry points with the wave-specialised kernel forced: what config[3]'s 1.25e7 particles per GPU select (>= MCS_WS_AUTO_MIN)"""
    monkeypatch.setenv("MCS_K1_WS", "1")
    test_strided_init_and_indexed_run()


def test_strided_init_and_indexed_run():
    """The multi-GPU entry points of round 2: mcs_init_pop_binned_strided deals the injection out (rank r of W holds
    global particles r, r + W, ...), mcs_run_pcut_indexed runs a shard whose global indices are an arbitrary list,
    mcs_saved_gidx returns the global indices of the saved ones.  All against the plain one-GPU calls: a particle's
    history depends on its global index only."""
    import torch
    N, W = 6000, 3
    prob = make_problem(N)
    hb = hip_backend(prob)
    inj = start_species(hb, prob)
    N = inj.n_pts_use
    full = hb.get_population()
    ns_full = hb.run_pcut(1, 0)
    fin, (sav, lsave) = hb.finals(), hb.get_saved()
    for r in range(W):
        n_loc = (N - r + W - 1) // W
        hb.init_pop(inj, r, n_loc, N, W)
        assert_pop_equal(hb.get_population(), full.take(np.arange(r, N, W)), f"strided init, rank {r} of {W}")
    # an arbitrary ascending index list (what a local split of an interleaved shard produces)
    rng = np.random.default_rng(5)
    sel = np.sort(rng.choice(N, size=1777, replace=False))
    hb.set_population(full.take(sel))
    g = torch.from_numpy(sel.astype(np.int64)).cuda()
    ns = hb.run_pcut_indexed(1, g)
    fa = hb.finals()
    for k in fa:
        assert np.array_equal(bits(fa[k]), bits(fin[k][sel])), k
    assert ns == int(lsave[sel].sum())
    assert np.array_equal(hb.saved_gidx().cpu().numpy(), sel[lsave[sel] == 1])
    gi, f64, meta = hb.export_saved(ns)           # the export carries the listed indices too
    assert np.array_equal(gi.cpu().numpy(), sel[lsave[sel] == 1])
    assert np.array_equal(bits(f64[0].cpu().numpy()), bits(sav.weight[sel][lsave[sel] == 1]))
    # argument checks
    with pytest.raises(RuntimeError, match="shard must lie inside"):
        hb.init_pop(inj, 1, N // W + 5, N, W)
    hb.destroy()


def _run_worker(world_env, args, timeout=200):
    import os, socket, subprocess, sys
    from conftest import ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world_env):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world_env), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), *args],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            assert p.returncode == 0, o.decode()[-3000:]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def test_rccl_world1_rehearsal():
    """The multi-GPU code path over RCCL itself (process group "nccl", bound torch tally tensors, all-gather of the
    counts and of the saved particles, mcs_saved_export / mcs_split_import, in-place all-reduce) with the one rank
    a one-GPU box can host, against the plain single-process run.  gather_max = 1500 makes the early pcuts take the
    local split and the late ones the gather."""
    import os, tempfile
    N, npc = 3000, 12
    out = os.path.join(tempfile.mkdtemp(), "w1.npz")
    _run_worker(1, [out, "hip", str(N), str(npc), "gather_max=1500"])
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=2)
    prob = mcs.inputs.build_problem(cfg)
    hb = hip_backend(prob)
    ref = mcs.driver.run(prob, hb, None, n_itrs=2, max_pcuts=npc)
    got = np.load(out)
    stats_ref = np.array([[s.i_iter, s.i_ion, s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult] for s in ref.stats])
    assert np.array_equal(got["stats"], stats_ref)
    assert np.array_equal(got["i"], ref.tallies_i64)
    assert (got["split"] == "local").any() and (got["split"] == "gather").any()
    assert_tallies_close(mcs.capi.Layout(prob.params), got["f"], ref.tallies_f64, 1e-10)
    hb.destroy()


def test_two_ranks_light_tallies_and_finalize():
    """What bench.py runs with N > 1: two ranks (gloo, one device), two species, the light tally hand-over at species ends
    (only the part behind the three histograms crosses to the host until the last iteration) and ion_finalize /
    iter_finalize on rank 0 -- against the plain single-process run."""
    import os, tempfile
    N, npc = 3000, 9
    out = os.path.join(tempfile.mkdtemp(), "w2l.npz")
    _run_worker(2, [out, "hip-gloo", str(N), str(npc), "2", "gather_max=800", "species_tallies=light", "finalize=1"])
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=2,
                            species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1)],
                            energy_transfer_frac=0.1)
    prob = mcs.inputs.build_problem(cfg)
    hb = hip_backend(prob)
    ref = mcs.driver.run(prob, hb, None, n_itrs=2, max_pcuts=npc, finalize=True)
    got = np.load(out)
    stats_ref = np.array([[s.i_iter, s.i_ion, s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult] for s in ref.stats])
    assert np.array_equal(got["stats"], stats_ref)
    assert np.array_equal(got["i"], ref.tallies_i64)
    assert_tallies_close(mcs.capi.Layout(prob.params), got["f"], ref.tallies_f64, 1e-10)
    hb.destroy()


def test_protons_ragged_population():
    """N not a multiple of the wavefront size; 12 pcuts of the stock ladder."""
    N = 3001
    _lockstep(make_problem(N), N, 12)


def test_oblique_field_tables():
    """theta_B != 0 and u_z != 0 tables (the reference's input check refuses oblique shocks,
    but the transforms are written for them: transformers.jl:523-607): exercises the gyro
    term of the move and the general boosts."""
    N = 700
    prob = make_problem(N)
    th = np.where(prob.x_grid_cm < 0, 0.35, 0.8)
    prob.theta = th
    prob.uz = 0.05 * prob.ux
    prob.utot = np.hypot(prob.ux, prob.uz)
    _lockstep(prob, N, 8)


def modified_profile(prob, depth=0.3, scale_rg=50.0):
    """A precursor: u_x falls smoothly from u0 to (1-depth) u0 ahead of the subshock, so u_x (and
    gamma_sf) differs in every upstream zone near the shock -- what the reference's profile
    smoothing produces after the first iteration.  Every zone crossing there needs transform_p_PSP."""
    C = mcs.constants.C
    x = prob.x_grid_cm
    u0 = prob.ux[1]
    up = x < 0
    ux = prob.ux.copy()
    ux[up] = u0 * (1 - depth * np.exp(x[up] / (scale_rg * prob.rg0)))
    prob.ux = ux
    prob.utot = np.hypot(prob.ux, prob.uz)
    prob.gam_sf = 1 / np.sqrt(1 - (prob.utot / C) ** 2)
    return prob


def test_modified_shock_profile():
    N = 900
    prob = modified_profile(make_problem(N))
    _lockstep(prob, N, 9)


def test_modified_profile_with_energy_transfer():
    ME_MP = mcs.constants.ME / mcs.constants.MP
    N = 400
    prob = make_problem(N, species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(ME_MP, -1.0, 1e6, 1.0)],
                        energy_transfer_frac=0.2, radiation_losses=True)
    modified_profile(prob)
    _lockstep(prob, N, 6)


def test_custom_epsB_flag_and_downstream_feb():
    N = 500
    prob = make_problem(N, FEB_downstream=(30.0, 0.0), b_field_turbulence=1.0)
    prob.params.use_custom_epsB = 1
    _lockstep(prob, N, 8)


def test_init_pop_binned_equals_per_particle_arrays():
    """mcs_init_pop_binned (O(bins) upload, what the driver uses) against mcs_init_pop on the expanded
    per-particle arrays (the reference's form of the call), whole population and a shard of it."""
    N = 5000
    prob = make_problem(N)
    inj = mcs.inputs.init_pop_host(prob, 1)
    n = inj.n_pts_use
    assert np.array_equal(np.repeat(inj.bin_ptot, inj.bin_count), inj.ptot_pf) and inj.bin_start[-1] == n
    hb = hip_backend(prob)
    for lo, hi in ((0, n), (1234, 4321)):
        start_species(hb, prob)
        hb.init_pop(inj, lo, hi - lo, n)
        a = hb.get_population()
        hb.init_pop_arrays(inj, lo, hi - lo, n)
        assert_pop_equal(a, hb.get_population(), f"binned vs arrays, shard [{lo},{hi})")
    # argument checks
    bs = inj.bin_start.copy(); bs[-1] += 1
    rc = hb.lib.mcs_init_pop_binned(hb.h, n, 0, n, len(inj.bin_ptot), inj.bin_ptot.ctypes.data_as(mcs.capi.c_double_p),
                                    inj.bin_weight.ctypes.data_as(mcs.capi.c_double_p), bs.ctypes.data_as(mcs.capi.c_int64_p),
                                    inj.x_start_cm, inj.i_grid_start, int(inj.relativistic), int(inj.fast_push))
    assert rc != 0 and b"bin_start" in hb.lib.mcs_last_error()
    hb.destroy()


def test_host_buffer_drop_in_call():
    """mcs_run_pcut_host(in, saved_out, l_save): the literal replacement of the loop at
    src/main_loops.jl:228-292 gives the same arrays as the resident path."""
    N = 777
    prob = make_problem(N)
    hb, ob = hip_backend(prob), oracle_backend(prob)
    start_species(hb, prob); start_species(ob, prob)
    pop = ob.get_population()
    for ip in (1, 2):
        saved, l_save, ns = hb.run_pcut_host(ip, pop, 0)
        assert ns == ob.run_pcut(ip, 0)
        so, lo = ob.get_saved()
        assert np.array_equal(l_save, lo)
        assert_pop_equal(saved, so, "host-buffer saved arrays")
        ob.new_pcut(1)
        pop = ob.get_population()
    hb.destroy()


def test_edge_cases_and_errors():
    prob = make_problem(64)
    hb = hip_backend(prob)
    start_species(hb, prob)
    # empty population
    hb.set_population(mcs.capi.Population(0))
    assert hb.run_pcut(1, 0) == 0 and hb.new_pcut(1) == 0
    # single particle
    ob = oracle_backend(prob)
    start_species(ob, prob)
    one = ob.get_population().slice(5, 6)
    hb.set_population(one); ob.set_population(one)
    assert hb.run_pcut(1, 5) == ob.run_pcut(1, 5)
    assert np.array_equal(bits(hb.finals()["ptot"]), bits(ob.finals()["ptot"]))
    # zero-momentum particle (reference quirk G6) and bad zone index are rejected, not simulated
    start_species(ob, prob)
    bad = ob.get_population().slice(0, 4)
    bad.ptot_pf[2] = 0.0
    with pytest.raises(RuntimeError, match="ptot_pf must be > 0"):
        hb.set_population(bad)
    bad = ob.get_population().slice(0, 4)
    bad.grid[1] = prob.n_grid + 5
    with pytest.raises(RuntimeError, match="grid index"):
        hb.set_population(bad)
    with pytest.raises(RuntimeError, match="i_pcut out of range"):
        hb.run_pcut(len(prob.pcuts) + 1, 0)
    hb.destroy()


@pytest.mark.parametrize("case", ["plain", "plain_etf"])
def test_specialised_and_general_kernel_agree(monkeypatch, case):
    """The common configuration runs mcs_k_transport_plain (compile-time flags), the same with ion -> electron energy transfer
    on (the ions of a multi-species run) mcs_k_transport_plain_etf; MCS_FORCE_GENERAL=1 keeps the general kernel.  Same problem
    through both: identical particles, identical integer tallies."""
    N = 4000
    kw = {} if case == "plain" else dict(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1)],
                                         energy_transfer_frac=0.1, radiation_losses=True)
    prob = make_problem(N, **kw)
    out = []
    for force in ("0", "1"):
        monkeypatch.setenv("MCS_FORCE_GENERAL", force)
        hb = hip_backend(prob)
        start_species(hb, prob)
        fin = []
        for ip in range(1, 10):
            ns = hb.run_pcut(ip, 0)
            fin.append((hb.finals(), hb.get_saved()))
            hb.new_pcut(max(N // ns, 1))
        out.append((fin, hb.read_tallies()))
        assert hb.last_kernel() == (0 if force == "1" else (1 if case == "plain" else 6))
        hb.destroy()
    (fa, (Ta, Ia)), (fb, (Tb, Ib)) = out
    for (xa, (sa, la)), (xb, (sb, lb)) in zip(fa, fb):
        for k in xa:
            assert np.array_equal(bits(xa[k]), bits(xb[k])), k
        assert np.array_equal(la, lb)
        assert_pop_equal(sa, sb, "saved arrays, plain vs general kernel")
    assert np.array_equal(Ia, Ib)
    assert_tallies_close(mcs.capi.Layout(prob.params), Ta, Tb, TALLY_RTOL)


@pytest.mark.parametrize("case,N", [("plain", 4000), ("plain", 70000), ("plain_etf", 4000), ("plain", 300)])
def test_wave_specialised_kernel_is_bit_identical(monkeypatch, case, N):
    """MCS_K1_WS=1 runs the common configuration through mcs_k_transport_ws (csrc/mcs_transport_ws.inc): particles are 29-word
    records that move between lanes and waves through LDS rings, the rare code is served 64 lanes wide by whichever wave finds 64
    particles pending.  State and RNG stream position travel with the particle, so every particle, saved array and integer tally
    must equal those of the lane-owns-particle kernel (last_kernel 1 / 6) -- which the other tests pin to the oracle -- and the
    binned tallies differ by the order of their adds only.  Sizes: several blocks with a full population (70 000: 137 blocks), a
    few sparse blocks (4000), less than one block (300)."""
    kw = {} if case == "plain" else dict(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1)],
                                         energy_transfer_frac=0.1, radiation_losses=True)
    prob = make_problem(N, **kw)
    out = []
    for ws in ("1", "0"):
        monkeypatch.setenv("MCS_K1_WS", ws)
        hb = hip_backend(prob)
        start_species(hb, prob)
        fin = []
        for ip in range(1, 10):
            n = hb.pop_size()
            ns = hb.run_pcut(ip, 0)
            fin.append((hb.finals(), hb.get_saved()))
            if ns == 0: break
            hb.new_pcut(max(n // ns, 1))
        out.append((fin, hb.read_tallies()))
        assert hb.last_kernel() == {("1", "plain"): 7, ("1", "plain_etf"): 8, ("0", "plain"): 1, ("0", "plain_etf"): 6}[(ws, case)]
        hb.destroy()
    (fa, (Ta, Ia)), (fb, (Tb, Ib)) = out
    assert len(fa) == len(fb)
    for (xa, (sa, la)), (xb, (sb, lb)) in zip(fa, fb):
        for k in xa:
            assert np.array_equal(bits(xa[k]), bits(xb[k])), k
        assert np.array_equal(la, lb)
        assert_pop_equal(sa, sb, "saved arrays, wave-specialised vs lane-owns-particle kernel")
    assert np.array_equal(Ia, Ib)
    assert_tallies_close(mcs.capi.Layout(prob.params), Ta, Tb, TALLY_RTOL)


@pytest.mark.parametrize("case,N", [("plain", 6000), ("general", 6000), ("plain_etf", 4000), ("plain", 40)])
def test_tail_loop_is_bit_identical(monkeypatch, case, N):
    """Once a launch's queue has run dry, a wave with at most MCS_TAIL_LOOP (12) live particles runs its passes in a loop of their own
    (transport_body "tail loop": the draw-dependent half from the tail ring, the state-dependent half with its literals in scalar
    registers, one exit test; a plain zone crossing that ended the loop handled right behind it) instead of the six-pass trip of the
    bulk.  Same statements on the same values in the same order: every particle, saved array and integer tally must equal those of
    MCS_TAIL_LOOP=0 -- which the other tests of this file ran against the oracle until the loop existed, as they now run the loop
    against it.  Sizes: a few blocks that reach the tail at once (6000, 4000), less than one wave (40: in the loop from the start);
    the PLAIN kernel, the general one (MCS_FORCE_GENERAL=1) and the ions' kernel of a species mix."""
    kw = {} if case != "plain_etf" else dict(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1)],
                                             energy_transfer_frac=0.1, radiation_losses=True)
    prob = make_problem(N, **kw)
    monkeypatch.setenv("MCS_FORCE_GENERAL", "1" if case == "general" else "0")
    out = []
    for tl in ("12", "0"):
        monkeypatch.setenv("MCS_TAIL_LOOP", tl)
        hb = hip_backend(prob)
        start_species(hb, prob)
        fin = []
        for ip in range(1, 12):
            n = hb.pop_size()
            ns = hb.run_pcut(ip, 0)
            fin.append((hb.finals(), hb.get_saved()))
            if ns == 0: break
            hb.new_pcut(max(n // ns, 1))
        out.append((fin, hb.read_tallies()))
        assert hb.last_kernel() == {"plain": 1, "general": 0, "plain_etf": 6}[case]
        hb.destroy()
    (fa, (Ta, Ia)), (fb, (Tb, Ib)) = out
    assert len(fa) == len(fb)
    for (xa, (sa, la)), (xb, (sb, lb)) in zip(fa, fb):
        for k in xa:
            assert np.array_equal(bits(xa[k]), bits(xb[k])), k
        assert np.array_equal(la, lb)
        assert_pop_equal(sa, sb, "saved arrays, tail loop on vs off")
    assert np.array_equal(Ia, Ib)
    assert_tallies_close(mcs.capi.Layout(prob.params), Ta, Tb, TALLY_RTOL)


def test_wave_specialised_kernel_is_picked_by_population_size(monkeypatch):
    """Without MCS_K1_WS the library picks the wave-specialised kernel for launches of at least MCS_WS_AUTO_MIN particles (6e6: it is
    level with the lane-owns-particle kernel at 4e6 and 3.8 % faster at 1e7, profiles/r04_ws_kernel_ab.txt) and the lane-owns-particle
    kernel below -- per launch, so a species whose population shrinks changes kernel on the way; the fused species loop decides by the
    largest population its chunk can hold.  The 1e7 and 5e7 tests of tests/test_gpu_full_size.py therefore run through it by default."""
    monkeypatch.delenv("MCS_K1_WS", raising=False)
    prob = make_problem(30000)
    for auto_min, want in (("20000", 7), ("40000", 1), (None, 1)):
        if auto_min is None: monkeypatch.delenv("MCS_WS_AUTO_MIN", raising=False)
        else: monkeypatch.setenv("MCS_WS_AUTO_MIN", auto_min)
        hb = hip_backend(prob)
        start_species(hb, prob)
        hb.run_pcut(1, 0)
        assert hb.last_kernel() == want, (auto_min, hb.last_kernel())
        hb.destroy()


@pytest.mark.parametrize("case", ["crafted", "thermal_mixed"])
def test_lossy_kernel_agrees_with_general_and_oracle(monkeypatch, case):
    """Electrons with radiative losses run mcs_k_transport_lossy: the loss of every pass (particle_loop.jl:302-326) and the
    refreshes that hang on the momentum in line in the common pass, where the general kernel visits the rare region in every
    pass (MCS_FORCE_GENERAL=1).  Same statements on the same values: particles and integer tallies identical between the two
    kernels and the oracle.  `crafted`: the hand-placed relativistic electrons of the golden case electrons_crafted_n64 at 4096
    particles (strong field: losses that matter, the constant-mfp branch below p_e,crit, PRP shortening, retro walks with
    losses, zero-energy exits); `thermal_mixed`: the electron species of the p + e- configuration of the golden case mixed_n96
    (x_spec detectors, injection probability < 1, energy transfer)."""
    from golden_common import make_golden
    name = "electrons_crafted_n64" if case == "crafted" else "mixed_n96"
    spec = make_golden.CASES[name]
    kw = dict(spec["cfg"])
    kw["species"] = [mcs.inputs.Species(**sp) for sp in kw["species"]]
    xs_rg = kw.pop("XSPEC_rg", None)
    N = 4096
    def build():
        cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, **kw)
        if xs_rg is not None:
            cfg.XSPEC = tuple(x * mcs.inputs.build_problem(cfg).rg0 for x in xs_rg)
        return mcs.inputs.build_problem(cfg)
    i_ion = len(kw["species"])          # the electrons
    def run(backend, prob):
        cfg = prob.cfg; sp = cfg.species[i_ion - 1]
        backend.begin_iteration(1)
        inj = mcs.inputs.init_pop_host(prob, i_ion)
        pmax = mcs.inputs.get_pmax_cutoff(prob.Emax_keV, prob.Emax_per_aa_keV, prob.pmax, sp.aa)
        backend.begin_species(1, i_ion, sp.aa, abs(sp.zz), pmax, sp.density, 1.0 / cfg.species[-1].density)
        backend.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
        if case == "crafted": backend.set_population(make_golden.crafted_population("electrons", prob, N))
        else: backend.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
        fin = []
        for ip in range(1, 5):
            n = backend.pop_size()
            ns = backend.run_pcut(ip, 0)
            fin.append((backend.finals(), backend.get_saved()))
            if ns == 0: break
            backend.new_pcut(max(n // ns, 1))
        return fin, backend.read_tallies()
    out = []
    for force in ("0", "1"):
        monkeypatch.setenv("MCS_FORCE_GENERAL", force)
        prob = build()
        hb = hip_backend(prob)
        out.append(run(hb, prob))
        assert hb.last_kernel() == (2 if force == "0" else 0)
        L = hb.layout
        hb.destroy()
    prob = build()
    ob = oracle_backend(prob, nthreads=16)
    out.append(run(ob, prob))
    ob.destroy()
    (fa, (Ta, Ia)) = out[0]
    IC, ng = mcs.capi.IC, prob.n_grid
    assert int(Ia[ng + IC["STEPS_HELIX"]]) > 100 * N          # (the losses run in the helix loop, not in a plumbing case)
    for which, (fb, (Tb, Ib)) in (("general kernel", out[1]), ("oracle", out[2])):
        assert len(fa) == len(fb)
        for ip, ((xa, (sa, la)), (xb, (sb, lb))) in enumerate(zip(fa, fb), 1):
            for k in xa:
                assert np.array_equal(bits(xa[k]), bits(xb[k])), f"lossy kernel vs {which}: pcut {ip}, {k}"
            assert np.array_equal(la, lb)
            assert_pop_equal(sa, sb, f"saved arrays, lossy kernel vs {which}, pcut {ip}")
        assert np.array_equal(Ia, Ib), which
        assert_tallies_close(L, Ta, Tb, TALLY_RTOL)


@pytest.mark.parametrize("kind", ["protons", "general", "electrons", "oblique", "protons_ws"])
def test_fuzzed_caller_populations_vs_oracle(kind, monkeypatch):
    """mcs_pop_upload takes any population, not only the ones the path produces: 8192 random particles (conftest.fuzz_population --
    every combination of downstream / inj, positions from beyond the upstream FEB to downstream of x_grid_stop and within 1e-7
    r_g of the shock, six decades of momentum, the PRP on either side, ages around age_max, every time-cut index) through three
    pcuts on the GPU and on the oracle.  `protons`: the plain kernel; `general`: p + He, energy transfer, downstream FEB, injection
    probability 0.7, a low p_max; `electrons`: radiative losses in a strong field, the lossy kernel; `oblique`: a precursor profile
    with an oblique field.  All four finish reasons occur.  Particles, saved arrays and integer tallies bit for bit, binned
    tallies to 1e-11.  (This is the test that would have found the `inj` update missing after the first move of a
    downstream-flagged particle loaded at x < 0: DESIGN.md section 3.)"""
    from conftest import fuzz_population, fuzz_problem
    N = 8192
    if kind == "protons_ws":       # the same population through the wave-specialised kernel (csrc/mcs_transport_ws.inc)
        monkeypatch.setenv("MCS_K1_WS", "1")
        kind = "protons+ws"
    prob, aa = fuzz_problem(kind.split("+")[0], N)
    pop = fuzz_population(prob, N, 1, aa)
    assert int(((pop.downstream == 1) & (pop.inj == 0) & (pop.x_PT_cm < 0)).sum()) > 100
    sp = prob.cfg.species[0]
    inj = mcs.inputs.init_pop_host(prob, 1)
    pmax = mcs.inputs.get_pmax_cutoff(prob.Emax_keV, prob.Emax_per_aa_keV, prob.pmax, sp.aa)
    def run(be):
        be.begin_iteration(1)
        be.begin_species(1, 1, sp.aa, abs(sp.zz), pmax, sp.density, 1.0)
        be.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
        be.set_population(pop)
        out = []
        for ip in range(2, 5):
            n = be.pop_size()
            ns = be.run_pcut(ip, 0)
            out.append((be.finals(), be.get_saved()))
            if ns == 0: break
            be.new_pcut(max(n // ns, 1))
        return out, be.read_tallies()
    hb, ob = hip_backend(prob), oracle_backend(prob, nthreads=16)
    (fa, (Ta, Ia)), (fb, (Tb, Ib)) = run(hb), run(ob)
    assert hb.last_kernel() == {"protons": 1, "general": 0, "electrons": 2, "oblique": 0, "protons+ws": 7}[kind]
    reasons = np.zeros(5, dtype=np.int64)
    assert len(fa) == len(fb)
    for ip, ((xa, (sa, la)), (xb, (sb, lb))) in enumerate(zip(fa, fb), 2):
        for k in xa:
            assert np.array_equal(bits(xa[k]), bits(xb[k])), f"{kind}: pcut {ip}, {k}"
        assert np.array_equal(la, lb)
        assert_pop_equal(sa, sb, f"{kind}: saved arrays, pcut {ip}")
        reasons += np.bincount(xa["reason"], minlength=5)[:5]
    assert np.all(reasons[:4] > 0), reasons
    assert np.array_equal(Ia, Ib)
    assert_tallies_close(hb.layout, Ta, Tb, TALLY_RTOL)
    hb.destroy(); ob.destroy()


def test_parking_and_tail_consolidation_do_not_change_results(monkeypatch):
    """K1 reorders work: a lane that needs the full Code Blocks waits (F_WAIT) until the next refill releases the
    batch (MCS_PARK; round 1 parked the particle in global memory), several common passes run per trip through the loop
    header with deferred entry into the rare region, and sparse waves of a block merge -- particles move between lanes --
    after the work counter is exhausted (MCS_TAIL_MERGE, with the tail ring MCS_TAIL_RING on its idle lanes); the
    tallies go to per-block replicas that are folded into the buffer when it is read (MCS_TALLY_REPLICAS_OFF).
    The state and the RNG stream travel with the particle: with both switched off the particles are bit-identical.
    Few blocks, so that every lane is refilled many times and both mechanisms have work."""
    N = 6000
    prob = make_problem(N)
    out = []
    for on in ("1", "0"):
        monkeypatch.setenv("MCS_PARK", on)
        monkeypatch.setenv("MCS_TAIL_MERGE", on)
        monkeypatch.setenv("MCS_TAIL_RING", on)
        monkeypatch.setenv("MCS_DEFER_K", "8" if on == "1" else "1")
        monkeypatch.setenv("MCS_TALLY_REPLICAS_OFF", "0" if on == "1" else "1")   # 16 tally replicas folded at read time
        hb = hip_backend(prob)
        hb.set_launch(4, 256)
        start_species(hb, prob)
        fin = []
        for ip in range(1, 12):
            ns = hb.run_pcut(ip, 0)
            fin.append((hb.finals(), hb.get_saved()))
            hb.new_pcut(max(N // ns, 1))
        out.append((fin, hb.read_tallies()))
        hb.destroy()
    (fa, (Ta, Ia)), (fb, (Tb, Ib)) = out
    for (xa, (sa, la)), (xb, (sb, lb)) in zip(fa, fb):
        for k in xa:
            assert np.array_equal(bits(xa[k]), bits(xb[k])), k
        assert np.array_equal(la, lb)
        assert_pop_equal(sa, sb, "saved arrays, with and without parking / tail consolidation")
    assert np.array_equal(Ia, Ib)
    assert_tallies_close(mcs.capi.Layout(prob.params), Ta, Tb, TALLY_RTOL)


def test_results_do_not_depend_on_launch_geometry():
    """Which lane runs which particle is irrelevant: few blocks (lanes are refilled many
    times) and the automatic geometry give bit-identical particles."""
    N = 5000
    prob = make_problem(N)
    outs = []
    for blocks in (3, 0):
        hb = hip_backend(prob)
        hb.set_launch(blocks, 256 if blocks else 0)
        start_species(hb, prob)
        fin = []
        for ip in range(1, 8):
            ns = hb.run_pcut(ip, 0)
            fin.append(hb.finals())
            hb.new_pcut(max(N // ns, 1))
        outs.append((fin, hb.read_tallies()))
        hb.destroy()
    for a, b in zip(outs[0][0], outs[1][0]):
        for k in a:
            assert np.array_equal(bits(a[k]), bits(b[k]))
    assert np.array_equal(outs[0][1][1], outs[1][1][1])
    assert_tallies_close(mcs.capi.Layout(prob.params), outs[0][1][0], outs[1][1][0], TALLY_RTOL)


def test_two_ranks_device_side_merge():
    """Two ranks of the host driver sharing this GPU (CUDA tensors over gloo): the device-side
    tally merge (bound torch tensors, in-place all-reduce, rank-0-only baselines) must
    reproduce the single-process HIP run."""
    import os, tempfile
    N, npc = 3000, 9
    out = os.path.join(tempfile.mkdtemp(), "w2.npz")
    _run_worker(2, [out, "hip-gloo", str(N), str(npc), "2", "gather_max=800"])
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=2,
                            species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1)],
                            energy_transfer_frac=0.1)
    prob = mcs.inputs.build_problem(cfg)
    hb = hip_backend(prob)
    ref = mcs.driver.run(prob, hb, None, n_itrs=2, max_pcuts=npc)
    got = np.load(out)
    stats_ref = np.array([[s.i_iter, s.i_ion, s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult] for s in ref.stats])
    assert np.array_equal(got["stats"], stats_ref)
    assert np.array_equal(got["i"], ref.tallies_i64)
    assert_tallies_close(mcs.capi.Layout(prob.params), got["f"], ref.tallies_f64, 1e-10)
    hb.destroy()


def test_full_iteration_binned_spectra_vs_oracle():
    """North-star parity statement: a whole iteration (all 45 pcuts, K3 + K1 + K2 resident on the GPU,
    driven by driver.run) against the CPU oracle on the same seeded inputs.  Integer tallies
    (num_crossings, exit reasons, step counts) must be equal; every fp64 array of binned spectra
    (psd, thermal histograms, escape spectra, fluxes, coupled spectra, pools, scalars) must agree
    to 1e-11 of its maximum -- the only difference allowed is the order of the atomic adds."""
    N = 20_000
    prob = make_problem(N)
    ob = oracle_backend(prob, nthreads=16)
    hb = hip_backend(prob)
    ro = mcs.driver.run(prob, ob, n_itrs=1)
    rg = mcs.driver.run(prob, hb, n_itrs=1)
    assert np.array_equal(rg.tallies_i64, ro.tallies_i64)
    assert (rg.steps_helix, rg.steps_retro) == (ro.steps_helix, ro.steps_retro)
    assert [(s.n_pts_use, s.n_saved, s.i_mult) for s in rg.stats] == [(s.n_pts_use, s.n_saved, s.i_mult) for s in ro.stats]
    assert_tallies_close(hb.layout, rg.tallies_f64, ro.tallies_f64, TALLY_RTOL)
    hb.destroy(); ob.destroy()


def test_config4_mixed_species_full_iteration_vs_oracle():
    """BASELINE config[4]'s species mix on one GPU, fp64: protons + He + electrons with radiative losses and ion ->
    electron energy transfer (the general kernel: every_pass / odd_cfg paths), a whole iteration through driver.run
    (three species, all pcuts, tallies accumulated across species as the reference does) against the oracle.  Same
    bar as the proton iteration: integers and per-pcut populations equal, binned spectra to 1e-11."""
    N = 8_000
    me_mp = mcs.constants.ME / mcs.constants.MP
    prob = make_problem(N, species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1),
                                    mcs.inputs.Species(me_mp, -1.0, 1e6, 1.2)],
                        energy_transfer_frac=0.1, radiation_losses=True)
    ob = oracle_backend(prob, nthreads=32)
    hb = hip_backend(prob)
    ro = mcs.driver.run(prob, ob, n_itrs=1)
    rg = mcs.driver.run(prob, hb, n_itrs=1)
    assert len({s.i_ion for s in rg.stats}) == 3
    assert np.array_equal(rg.tallies_i64, ro.tallies_i64)
    assert (rg.steps_helix, rg.steps_retro) == (ro.steps_helix, ro.steps_retro)
    assert [(s.i_ion, s.n_pts_use, s.n_saved, s.i_mult) for s in rg.stats] == [(s.i_ion, s.n_pts_use, s.n_saved, s.i_mult) for s in ro.stats]
    assert_tallies_close(hb.layout, rg.tallies_f64, ro.tallies_f64, TALLY_RTOL)
    hb.destroy(); ob.destroy()


def test_light_tally_readback():
    """mcs_read_tallies_part / driver.run(species_tallies="light"): at a species end the host takes only the part of the tally
    buffer behind the three big histograms (they have consumers on the device); the final result is the complete buffer and
    equals the one of a run that moved everything at every species end."""
    N = 4000
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=2)
    prob = mcs.inputs.build_problem(cfg)
    hb = hip_backend(prob)
    seen = []
    rl = mcs.driver.run(prob, hb, None, n_itrs=2, max_pcuts=8, species_tallies="light", finalize=True,
                        on_species_end=lambda it, ion, f, i: seen.append((it, f.copy())))
    f_full, i_full = hb.read_tallies()
    f_light, i_light = hb.read_tallies_light()
    L = hb.layout
    o = L.offsets["esc_psd_up"]
    assert np.array_equal(i_full, i_light) and np.array_equal(f_full[o:], f_light[o:]) and not f_light[:o].any()
    assert f_full[:o].max() > 1e-90     # (the histograms are not empty: their floor is 1e-99)
    hb.destroy()
    hb = hip_backend(prob)
    rf = mcs.driver.run(prob, hb, None, n_itrs=2, max_pcuts=8, finalize=True)
    assert np.array_equal(rl.tallies_i64, rf.tallies_i64)
    assert_tallies_close(L, rl.tallies_f64, rf.tallies_f64, TALLY_RTOL)
    assert not seen[0][1][:o].any() and seen[1][1][:o].any()          # iteration 1 light, the last one complete
    (_, fa, ia), (_, fb, ib) = rl.iter_finals[-1], rf.iter_finals[-1]
    assert abs(fa.Gamma_downstream / fb.Gamma_downstream - 1) < 1e-10
    hb.destroy()


def test_overlapped_iterations_equal_sequential():
    """driver.run_overlapped: with a fixed shock profile the iterations are independent, so two of them share the GPU on
    two contexts / streams / host threads.  Every iteration must be the one the sequential run computes (same populations,
    same integer tallies per iteration, same iter_finalize results), and the merged state after the last one -- the last
    iteration's buffer with the never-reset tallies summed over the contexts -- must equal the sequential run's."""
    N, n_itrs, npc = 6000, 3, 14
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=n_itrs)
    prob = mcs.inputs.build_problem(cfg)
    hb = hip_backend(prob)
    seq = mcs.driver.run(prob, hb, None, n_itrs=n_itrs, max_pcuts=npc, finalize=True)
    hb.destroy()
    bes = [hip_backend(prob), hip_backend(prob)]
    ovl = mcs.driver.run_overlapped(prob, bes, n_itrs=n_itrs, max_pcuts=npc)
    key = lambda r: [(s.i_iter, s.i_ion, s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult) for s in r.stats]
    assert key(ovl) == key(seq)
    assert np.array_equal(ovl.tallies_i64, seq.tallies_i64)
    assert (ovl.steps_helix, ovl.steps_retro) == (seq.steps_helix, seq.steps_retro)
    assert [s for _, _, s in ovl.local_steps] == [s for _, _, s in seq.local_steps]
    assert_tallies_close(bes[0].layout, ovl.tallies_f64, seq.tallies_f64, TALLY_RTOL)
    for (ia, fa, _), (ib, fb, _) in zip(ovl.iter_finals, seq.iter_finals):
        assert ia == ib and abs(fa.Gamma_downstream / fb.Gamma_downstream - 1) < 1e-10
        assert abs(fa.q_esc_cal_px - fb.q_esc_cal_px) <= 1e-10 * max(abs(fb.q_esc_cal_px), 1e-30)
    for b in bes:
        b.destroy()


def test_linearity_in_the_weights():
    """Doubling all weights doubles every tally (power-of-two scaling is exact): 2e5 protons, 6 pcuts.
    (The full-size runs are in tests/test_gpu_full_size.py.)"""
    N = 1_000_000
    prob = make_problem(N)
    hb = hip_backend(prob)
    start_species(hb, prob)
    pop0 = hb.get_population()
    hb.destroy()
    n = 200_000
    res = []
    for scale in (1.0, 2.0):
        hb = hip_backend(prob)
        start_species(hb, prob)
        p = pop0.slice(0, n)
        p.weight *= scale
        hb.set_population(p)
        for ip in range(1, 7):
            ns = hb.run_pcut(ip, 0)
            hb.new_pcut(max(n // ns, 1))
        res.append(hb.read_tallies())
        hb.destroy()
    L = mcs.capi.Layout(prob.params)
    base = oracle_backend(prob)
    start_species(base, prob)
    B, _ = base.read_tallies()                       # the baseline fills (1e-99 floors, fast-push fluxes)
    assert np.array_equal(res[0][1], res[1][1])
    for name in ("psd", "therm_sf", "esc_psd_down", "pxx_flux", "energy_flux", "spectra_coupled"):
        a = L.view(res[0][0], name) - L.view(B, name)
        b = L.view(res[1][0], name) - L.view(B, name)
        scale_ = np.max(np.abs(b)) + 1e-300
        assert np.max(np.abs(2 * a - b)) / scale_ < 1e-10, name


# ---- the fp32-state variant against fp64, statistically.  The bounds are NOT constants fitted to a passing run: every quantity is
# compared with the seed ensemble of the fp64 path computed in the test -- K_ENS realisations (the seeds of iterations 1 .. K_ENS,
# src/particle_loop.jl:35-40) give its mean and its standard deviation s, and an fp32 realisation must lie within N_SIGMA * s of
# that mean.  N_SIGMA: a Gaussian 5-sigma bound on ~10^2 compared numbers fails by chance less than once in 10^4 runs; the factor
# 1.5 covers the uncertainty of a standard deviation estimated from K_ENS = 5 samples (relative error 1 / sqrt(2 (K - 1)) = 35 %).
# (Round 3 fitted 0.04 / 0.12 dex / 3 % / 7 % after a red run -- gpurun_out/r3/t_full2.log: total steps 4.47 % against a 3 % bound.)
K_ENS, N_SIGMA = 5, 5 * 1.5


def _realisation(N, fp32, it, n_iters, **kw):
    from mcs_amd import hip_backend as hbm
    prob = make_problem(N, state_fp32=fp32, num_iterations=n_iters, **kw)
    hb = hbm.HipBackend(0); hb.create(prob)
    r = mcs.driver.run(prob, hb, None, n_itrs=1, first_iter=it)
    hb.destroy()
    return prob, r


def _dndp_log(prob, L, T, zone):
    P = prob.params
    a = L.view(T, "psd")[zone - 1].sum(axis=0)
    mb = prob.psd_mom_bounds
    k = np.arange(1, P.num_psd_mom_bins)
    pc = 10.0 ** (0.5 * (mb[k] + mb[k + 1]))
    sel = (pc > 30.0) & (pc < 3.0e4)
    return np.log10(np.maximum(a[k][sel], 1e-300))


def _within(x32, ens, what, floor=0.0, pooled=False):
    """every fp32 value within N_SIGMA ensemble standard deviations (+ floor) of the ensemble mean, element-wise.  `pooled`: the
    entries are neighbouring bins of one spectrum with the same kind of noise -- a standard deviation estimated from five samples
    is itself uncertain by 35 %, and among 30 bins some estimate comes out at a third of the truth; the pooled estimate (root mean
    variance over the bins, ~120 degrees of freedom) is the floor of every bin's own."""
    ens = np.asarray(ens, dtype=np.float64)
    m, sd = ens.mean(axis=0), ens.std(axis=0, ddof=1)
    if pooled:
        sd = np.maximum(sd, np.sqrt(np.mean(sd ** 2)))
    for x in x32:
        dev = np.abs(np.asarray(x, dtype=np.float64) - m)
        lim = N_SIGMA * sd + floor
        assert np.all(dev <= lim), (what, float(np.max(dev / np.maximum(lim, 1e-300))), np.asarray(x).ravel()[:4], m.ravel()[:4], sd.ravel()[:4])


def test_fp32_state_variant_statistical_agreement():
    """BASELINE config[4]: the fp32-state kernel (state and per-step arithmetic in fp32, normalised units; tallies fp64) against
    the fp64 path, 2e5 protons, whole iterations.  Histories drift apart (roundings differ), so the agreement is statistical:
    three fp32 realisations against the ensemble of five fp64 realisations (see K_ENS / N_SIGMA above) in the fitted downstream
    slope and the bin-by-bin log10 dN/dp over the power law (30 .. 3e4 m_p c) in two zones, the population of every
    well-populated pcut and the total number of steps; the fp64 ensemble mean slope also has to sit at the Keshet & Waxman index
    within its own scatter; no zone-search failures, no capped retro walks."""
    from test_physics import keshet_waxman_slope, dndp_slope
    N = 200_000
    r64 = [_realisation(N, False, it, K_ENS) for it in range(1, K_ENS + 1)]
    r32 = [_realisation(N, True, it, K_ENS) for it in range(1, 4)]
    prob = r64[0][0]
    P, L = prob.params, mcs.capi.Layout(prob.params)
    for zone in (P.i_shock + 3, P.i_shock + 10):
        e = [dndp_slope(prob, L, r.tallies_f64, zone) for _, r in r64]
        _within([dndp_slope(prob, L, r.tallies_f64, zone) for _, r in r32], e, f"slope, zone {zone}")
        assert abs(np.mean(e) - keshet_waxman_slope(P)) < 0.05, (zone, e)      # (physics check of tests/test_physics.py, on the ensemble mean)
        _within([_dndp_log(prob, L, r.tallies_f64, zone) for _, r in r32], [_dndp_log(prob, L, r.tallies_f64, zone) for _, r in r64], f"log10 dN/dp, zone {zone}", pooled=True)
    # populations per pcut (the runs of an ensemble reach the same pcuts while they are well populated)
    npc = min(len(r.stats) for _, r in r64 + r32)
    ns64 = np.array([[r.stats[i].n_saved for i in range(npc)] for _, r in r64], dtype=np.float64)
    ns32 = np.array([[r.stats[i].n_saved for i in range(npc)] for _, r in r32], dtype=np.float64)
    big = ns64.mean(axis=0) > N // 20
    assert big.sum() >= 12
    _within(list(ns32[:, big] / ns64[:, big].mean(axis=0)), ns64[:, big] / ns64[:, big].mean(axis=0), "n_saved per pcut (relative)", pooled=True)
    _within([[r.steps_helix + r.steps_retro] for _, r in r32], [[r.steps_helix + r.steps_retro] for _, r in r64], "total steps")
    ng, IC = P.n_grid, mcs.capi.IC
    for _, r in r32:
        assert int(r.tallies_i64[ng + IC["ZONE_FAIL"]]) == 0 and int(r.tallies_i64[ng + IC["RETRO_CAP"]]) == 0


def test_fp32_state_variant_mixed_species():
    """BASELINE config[4]'s other half: the fp32-state kernel on the species mix (protons + He + electrons, radiative losses,
    ion -> electron energy transfer) against the fp64 path, 10^5 particles per species: two fp32 realisations against the
    ensemble of five fp64 realisations, per ion species -- slope and bin-by-bin log10 dN/dp just downstream, populations of the
    well-populated pcuts, total steps; the thermal electrons end at the helix cap in their first pcut in both precisions (Q5)."""
    from test_physics import dndp_slope
    N = 100_000
    me_mp = mcs.constants.ME / mcs.constants.MP
    kw = dict(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1), mcs.inputs.Species(me_mp, -1.0, 1e6, 1.2)],
              energy_transfer_frac=0.1, radiation_losses=True)
    r64 = [_realisation(N, False, it, K_ENS, **kw) for it in range(1, K_ENS + 1)]
    r32 = [_realisation(N, True, it, K_ENS, **kw) for it in range(1, 3)]
    prob = r64[0][0]
    P, L = prob.params, mcs.capi.Layout(prob.params)
    ng, IC = P.n_grid, mcs.capi.IC
    by_ion = lambda r: {ion: f for (_, ion, f, _) in r.per_species}
    zone = P.i_shock + 3
    for ion in (1, 2):                       # protons, He: the accelerated power law
        _within([dndp_slope(prob, L, by_ion(r)[ion], zone) for _, r in r32], [dndp_slope(prob, L, by_ion(r)[ion], zone) for _, r in r64], f"slope, ion {ion}")
        _within([_dndp_log(prob, L, by_ion(r)[ion], zone) for _, r in r32], [_dndp_log(prob, L, by_ion(r)[ion], zone) for _, r in r64], f"log10 dN/dp, ion {ion}", pooled=True)
        def pops(r):
            return {s.i_pcut: s.n_saved for s in r.stats if s.i_ion == ion}
        common = sorted(set.intersection(*[set(pops(r)) for _, r in r64 + r32]))
        e = np.array([[pops(r)[i] for i in common] for _, r in r64], dtype=np.float64)
        x = np.array([[pops(r)[i] for i in common] for _, r in r32], dtype=np.float64)
        big = e.mean(axis=0) > N // 10
        assert big.sum() >= 8, (ion, e.mean(axis=0))
        _within(list(x[:, big] / e[:, big].mean(axis=0)), e[:, big] / e[:, big].mean(axis=0), f"n_saved per pcut (relative), ion {ion}", pooled=True)
    for _, r in r64 + r32:
        el = [s for s in r.stats if s.i_ion == 3]
        assert el[0].n_saved == 0 and len(el) == 1      # Q5: the cap ends every thermal electron in pcut 1
    for _, r in r32:
        assert int(r.tallies_i64[ng + IC["ZONE_FAIL"]]) == 0 and int(r.tallies_i64[ng + IC["RETRO_CAP"]]) == 0
    _within([[r.steps_helix + r.steps_retro] for _, r in r32], [[r.steps_helix + r.steps_retro] for _, r in r64], "total steps")


def test_config0_stock_input_shape_vs_oracle():
    """BASELINE config[0]: the stock mc_in.toml at 10^4 particles, one iteration -- two species (protons and electrons),
    no-scatter = true and no-DSA = true (the committed file is a scatter-free plumbing run, /root/reference/mc_in.toml:136,139),
    radiative losses, energy-transfer-frac 0.1, compressed turbulence in the field, the custom-eps_B flag, the stock time and
    momentum cuts -- on the GPU and on the oracle through the same driver; integers, populations and binned tallies must
    agree as in every other whole-iteration test.  One thing of the stock file cannot be taken literally: DENZ_ION = [1, 0]
    gives the electrons zero weight and electron_weight_fac = 1/0 (quirk G4: n_e = n_p is used).  Round 4: the custom-eps_B
    PROFILE comes from the reference's initialiser (`set_custom_εB!`, restated in inputs.build_problem), no longer from the flag
    set on the compressed-turbulence tables."""
    N = 10_000
    me_mp = mcs.constants.ME / mcs.constants.MP
    kw = dict(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(me_mp, -1.0, 1e6, 1.0)],
              no_scatter=True, no_DSA=True, radiation_losses=True, energy_transfer_frac=0.1, b_field_turbulence=1.0,
              b_field_amplify=1.0, electron_energy_mfp_threshold=1e4, use_custom_epsB=True)
    pg, po = make_problem(N, **kw), make_problem(N, **kw)
    assert pg.params.use_custom_epsB == 1 and np.ptp(pg.btot[1:pg.n_grid + 1]) > 0       # the field of the eps_B profile, not a constant
    hb, ob = hip_backend(pg), oracle_backend(po, nthreads=16)
    rg = mcs.driver.run(pg, hb, None, n_itrs=1, finalize=True)
    ro = mcs.driver.run(po, ob, None, n_itrs=1, finalize=True)
    assert len({s.i_ion for s in rg.stats}) == 2
    assert [(s.i_ion, s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult) for s in rg.stats] == [(s.i_ion, s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult) for s in ro.stats]
    assert np.array_equal(rg.tallies_i64, ro.tallies_i64)
    assert_tallies_close(hb.layout, rg.tallies_f64, ro.tallies_f64, TALLY_RTOL)
    ng, IC = pg.n_grid, mcs.capi.IC
    assert int(rg.tallies_i64[ng + IC["RNG_DRAWS"]]) >= 0 and rg.steps_helix > N      # scatter-free: the particles advect through the grid
    (_, fg, _), = rg.iter_finals
    (_, fo, _), = ro.iter_finals
    assert abs(fg.Gamma_downstream / fo.Gamma_downstream - 1) < 1e-9
    hb.destroy(); ob.destroy()


def _run_fp32(monkeypatch, loop, N, n_pcuts, **kw):
    """finals + saved arrays of every pcut and the tallies of the fp32-state variant, plain loop or organised kernel"""
    monkeypatch.setenv("MCS_F32_LOOP", "1" if loop else "0")
    prob = make_problem(N, state_fp32=True, **kw)
    hb = hip_backend(prob)
    out = []
    for i_ion in range(1, len(prob.cfg.species) + 1):
        start_species(hb, prob, 1, i_ion) if i_ion == 1 else None
        if i_ion > 1:
            sp = prob.cfg.species[i_ion - 1]
            inj = mcs.inputs.init_pop_host(prob, i_ion)
            pmax = mcs.inputs.get_pmax_cutoff(prob.Emax_keV, prob.Emax_per_aa_keV, prob.pmax, sp.aa)
            hb.begin_species(1, i_ion, sp.aa, abs(sp.zz), pmax, sp.density, 1.0 / prob.cfg.species[-1].density)
            hb.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
            hb.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
        for ip in range(1, n_pcuts + 1):
            ns = hb.run_pcut(ip, 0)
            f = hb.finals(); sv, ls = hb.get_saved()
            out.append((i_ion, ip, ns, f, sv, ls))
            if ns == 0:
                break
            hb.new_pcut(max(N // ns, 1))
    T, I = hb.read_tallies()
    hb.destroy()
    return out, T, I, prob


@pytest.mark.parametrize("case", ["protons", "mixed", "oblique_modified"])
def test_fp32_kernels_agree(monkeypatch, case):
    """The fp32-state variant exists twice: as a plain per-lane loop (mcs_k_transport_f32_loop: the reference's loop body as
    written, in fp32 -- what defines the variant) and organised like the fp64 kernel (mcs_k_transport_f32: flag-driven common
    pass, cached slowly varying quantities, position thresholds, deferred tallies).  Both use the same fp32 helper functions in
    the same order per particle, so they must agree BIT FOR BIT on every particle's end state and saved state, on the integer
    tallies, and on the fp64 tallies up to the order of the adds."""
    kw, N, npc = {}, 6000, 9
    tweak = None
    if case == "mixed":
        me_mp = mcs.constants.ME / mcs.constants.MP
        kw = dict(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(me_mp, -1.0, 1e6, 1.0)], energy_transfer_frac=0.1,
                  radiation_losses=True, XSPEC=[-0.5, 0.05, 2.0], INJFR=[0.7, 1.0], b_field_turbulence=1.0, shock_speed=3.0)
        N, npc = 1500, 7
    res = []
    for loop in (True, False):
        if case == "oblique_modified":
            # (tables are edited after build_problem: done inside through a patched make_problem)
            import conftest
            orig = conftest.make_problem
            def mp(Nn, **k2):
                pr = orig(Nn, **k2)
                modified_profile(pr)
                pr.theta = np.where(pr.x_grid_cm < 0, 0.35, 0.8); pr.uz = 0.05 * pr.ux; pr.utot = np.hypot(pr.ux, pr.uz)
                pr.gam_sf = 1 / np.sqrt(1 - (pr.utot / mcs.constants.C) ** 2)
                return pr
            monkeypatch.setattr(sys.modules[__name__], "make_problem", mp)
        res.append(_run_fp32(monkeypatch, loop, N, npc, **kw))
    (oa, Ta, Ia, prob), (ob_, Tb, Ib, _) = res
    assert len(oa) == len(ob_) and len(oa) >= 5
    for (ia, pa, nsa, fa, sa, la), (ib, pb, nsb, fb, sb, lb) in zip(oa, ob_):
        assert (ia, pa, nsa) == (ib, pb, nsb), (ia, pa, nsa, nsb)
        for k in fa:
            assert np.array_equal(bits(fa[k]), bits(fb[k])), f"ion {ia} pcut {pa}: final {k} differs for {(fa[k] != fb[k]).sum()} particles"
        assert np.array_equal(la, lb)
        assert_pop_equal(sa, sb, f"ion {ia} pcut {pa}: saved arrays")
    assert np.array_equal(Ia, Ib)
    assert_tallies_close(mcs.capi.Layout(prob.params), Tb, Ta, TALLY_RTOL)


@pytest.mark.parametrize("kind", ["protons", "general", "electrons", "oblique"])
def test_fp32_kernels_agree_on_fuzzed_caller_populations(monkeypatch, kind):
    """The two forms of the fp32-state variant (plain loop / organised like K1) on the random caller-provided populations of
    test_fuzzed_caller_populations_vs_oracle: the plain loop runs the `inj` update, the exit tests and the zone reload in every
    pass as the reference writes them, the organised kernel only where its flags and events say so -- they must agree bit for
    bit on states the path never produces too."""
    from conftest import fuzz_population, fuzz_problem
    N = 8192
    res = []
    for loop in ("1", "0"):
        monkeypatch.setenv("MCS_F32_LOOP", loop)
        prob, aa = fuzz_problem(kind, N)
        prob.params.state_fp32 = 1
        pop = fuzz_population(prob, N, 1, aa)
        sp = prob.cfg.species[0]
        inj = mcs.inputs.init_pop_host(prob, 1)
        pmax = mcs.inputs.get_pmax_cutoff(prob.Emax_keV, prob.Emax_per_aa_keV, prob.pmax, sp.aa)
        hb = hip_backend(prob)
        hb.begin_iteration(1)
        hb.begin_species(1, 1, sp.aa, abs(sp.zz), pmax, sp.density, 1.0)
        hb.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
        hb.set_population(pop)
        out = []
        for ip in range(2, 5):
            n = hb.pop_size()
            ns = hb.run_pcut(ip, 0)
            out.append((hb.finals(), hb.get_saved()))
            if ns == 0: break
            hb.new_pcut(max(n // ns, 1))
        assert hb.last_kernel() == (4 if loop == "1" else (5 if kind == "electrons" else 3))
        res.append((out, hb.read_tallies(), hb.layout))
        hb.destroy()
    (fa, (Ta, Ia), L), (fb, (Tb, Ib), _) = res
    assert len(fa) == len(fb)
    reasons = np.zeros(5, dtype=np.int64)
    for ip, ((xa, (sa, la)), (xb, (sb, lb))) in enumerate(zip(fa, fb), 2):
        for k in xa:
            assert np.array_equal(bits(xa[k]), bits(xb[k])), f"{kind}: pcut {ip}, {k} differs for {(xa[k] != xb[k]).sum()} particles"
        assert np.array_equal(la, lb)
        assert_pop_equal(sa, sb, f"{kind}: saved arrays, pcut {ip}")
        reasons += np.bincount(xa["reason"], minlength=5)[:5]
    assert np.all(reasons[:3] > 0), reasons
    assert np.array_equal(Ia, Ib)
    assert_tallies_close(L, Ta, Tb, TALLY_RTOL)


# ---- the fp32-state variant, bit level (round 4) ----------------------------------------------------------------------------------
def _fp32_exact_pair(prob_maker, setup, pcuts, monkeypatch):
    """Run `pcuts` through the device's exact fp32 loop kernel (MCS_F32_EXACT=1, last_kernel 9) and through the CPU restatement
    (oracle/mcs_oracle_f32.inc, OracleBackend.f32_exact): ((finals, saved, l_save) per pcut, tallies) of both."""
    monkeypatch.setenv("MCS_F32_EXACT", "1")
    out = []
    for side in ("gpu", "cpu"):
        prob = prob_maker()
        prob.params.state_fp32 = 1
        be = hip_backend(prob) if side == "gpu" else oracle_backend(prob, nthreads=1)
        if side == "cpu":
            be.f32_exact = True
        setup(be, prob)
        fin = []
        for ip in pcuts:
            n = be.pop_size()
            ns = be.run_pcut(ip, 0)
            fin.append((be.finals(), be.get_saved()))
            if ns == 0: break
            be.new_pcut(max(n // ns, 1))
        if side == "gpu":
            assert be.last_kernel() == 9
        out.append((fin, be.read_tallies(), mcs.capi.Layout(prob.params)))
        be.destroy()
    return out


def _assert_bit_identical(out, what):
    (fa, (Ta, Ia), L), (fb, (Tb, Ib), _) = out
    assert len(fa) == len(fb)
    reasons = np.zeros(5, dtype=np.int64)
    for ip, ((xa, (sa, la)), (xb, (sb, lb))) in enumerate(zip(fa, fb)):
        for k in xa:
            assert np.array_equal(bits(xa[k]), bits(xb[k])), f"{what}: pcut #{ip}, {k} differs for {(xa[k] != xb[k]).sum()} of {len(xa[k])} particles"
        assert np.array_equal(la, lb)
        assert_pop_equal(sa, sb, f"{what}: saved arrays, pcut #{ip}")
        reasons += np.bincount(xa["reason"], minlength=5)[:5]
    assert np.array_equal(Ia, Ib), f"{what}: integer tallies"
    assert_tallies_close(L, Ta, Tb, TALLY_RTOL)
    return reasons


@pytest.mark.parametrize("name", CASES + ["protons_n3000"])
def test_fp32_exact_kernel_equals_cpu_restatement(monkeypatch, name):
    """The fp32-state variant had no bit-level check: its kernels use v_rcp_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32, which no CPU
    reproduces.  mcs_k_transport_f32_loop_exact is the plain-loop kernel compiled with the EXACT primitives of
    include/mcs_math_f32.h (correctly rounded + - * / sqrt, polynomial sin / cos / asin), and oracle/mcs_oracle_f32.inc restates
    it on the CPU: end states, saved arrays, l_save and integer tallies must be equal bit for bit, binned tallies to 1e-11 -- on
    the four golden cases (thermal protons; p + e- with losses, energy transfer, x_spec detectors and an injection probability;
    the no-scatter / no-DSA plumbing run; the crafted relativistic electrons) replayed in fp32 state, and on 3000 protons through
    nine pcuts."""
    from golden_common import make_golden
    monkeypatch.setenv("MCS_F32_EXACT", "1")
    res = []
    for side in ("gpu", "cpu"):
        if name == "protons_n3000":
            prob, spec = make_problem(3000), dict(N=3000, n_pcuts=9)
        else:
            prob, spec = make_golden.build_case(name)
        prob.params.state_fp32 = 1
        if side == "gpu":
            be = hip_backend(prob)
        else:
            be = oracle_backend(prob, nthreads=1); be.f32_exact = True
        res.append(make_golden.run_case(be, prob, spec, False))
        if side == "gpu":
            assert be.last_kernel() == 9
        be.destroy()
    got, want = res
    L = mcs.capi.Layout(prob.params)
    assert sorted(got.keys()) == sorted(want.keys())
    ended = 0
    for k in want:
        if "_tallies_" in k:
            continue
        a, b = got[k], want[k]
        assert a.shape == b.shape, k
        assert np.array_equal(bits(a), bits(b)), f"{name}:{k} differs in {(a != b).sum()} of {a.size} entries"
        if k.endswith("final_reason"):
            ended += int((b > 0).sum())
    assert ended > 0
    for ion in range(1, len(prob.cfg.species) + 1):
        assert np.array_equal(got[f"ion{ion}_tallies_i64"], want[f"ion{ion}_tallies_i64"]), f"{name}: int64 tallies, ion {ion}"
        ref = np.zeros(L.total); ref[want[f"ion{ion}_tallies_idx"]] = want[f"ion{ion}_tallies_val"]
        cur = np.zeros(L.total); cur[got[f"ion{ion}_tallies_idx"]] = got[f"ion{ion}_tallies_val"]
        assert_tallies_close(L, cur, ref, TALLY_RTOL)


@pytest.mark.parametrize("kind", ["protons", "general", "electrons", "oblique"])
def test_fp32_exact_kernel_equals_cpu_restatement_on_fuzzed_populations(monkeypatch, kind):
    """... and on the random caller-provided populations of test_fuzzed_caller_populations_vs_oracle (every combination of
    downstream / inj, positions from beyond the upstream FEB to downstream of x_grid_stop, six decades of momentum, ages around
    age_max): all four finish reasons occur."""
    from conftest import fuzz_population, fuzz_problem
    N = 4096
    holder = {}
    def maker():
        prob, aa = fuzz_problem(kind, N)
        holder["aa"] = aa
        return prob
    def setup(be, prob):
        pop = fuzz_population(prob, N, 1, holder["aa"])
        sp = prob.cfg.species[0]
        inj = mcs.inputs.init_pop_host(prob, 1)
        pmax = mcs.inputs.get_pmax_cutoff(prob.Emax_keV, prob.Emax_per_aa_keV, prob.pmax, sp.aa)
        be.begin_iteration(1)
        be.begin_species(1, 1, sp.aa, abs(sp.zz), pmax, sp.density, 1.0)
        be.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
        be.set_population(pop)
    out = _fp32_exact_pair(maker, setup, range(2, 5), monkeypatch)
    r = _assert_bit_identical(out, kind)
    assert np.all(r[:3] > 0), r


def test_fp32_hardware_primitives_against_the_exact_build(monkeypatch):
    """The shipping fp32 kernels (hardware reciprocal / square root / sine / cosine, ~1 ulp) against the exact build of the same
    loop: histories start identical and part where a rounding flips a compare.  Bounds a priori: per scatter the two builds
    differ by O(1e-7) relative, so after one pcut of ~1e2-1e3 passes (a) most short histories are still IDENTICAL in their
    discrete outcome (exit reason, step count), (b) populations differ by the particles that flipped -- a few per cent --, and
    (c) binned spectra agree to the Monte-Carlo scatter of that population size, estimated in the test from the exact build run
    with another seed stream (iteration 2), not from a constant."""
    N = 40000
    def run(env, i_iter=1):
        for k, v in env.items(): monkeypatch.setenv(k, v)
        prob = make_problem(N, num_iterations=2); prob.params.state_fp32 = 1
        hb = hip_backend(prob)
        start_species(hb, prob, i_iter=i_iter)
        per = []
        for ip in range(1, 8):
            n = hb.pop_size(); ns = hb.run_pcut(ip, 0)
            per.append((n, ns, hb.finals()))
            if ns == 0: break
            hb.new_pcut(max(n // ns, 1))
        T, I = hb.read_tallies(); L = hb.layout; k = hb.last_kernel()
        hb.destroy()
        return per, T, I, L, k
    ex, Te, Ie, L, ke = run({"MCS_F32_EXACT": "1", "MCS_F32_LOOP": "0"})
    hw, Th, Ih, _, kh = run({"MCS_F32_EXACT": "0", "MCS_F32_LOOP": "1"})
    e2, T2, I2, _, _ = run({"MCS_F32_EXACT": "1", "MCS_F32_LOOP": "0"}, i_iter=2)
    assert ke == 9 and kh == 4
    # (a) the first heavy pcut starts from identical populations: most histories keep their discrete outcome
    first = next(i for i, (n, ns, f) in enumerate(ex) if f["helix"].mean() > 20)
    same = (ex[first][2]["reason"] == hw[first][2]["reason"]) & (ex[first][2]["helix"] == hw[first][2]["helix"])
    assert same.mean() > 0.5, same.mean()
    # (b), (c): against the seed-to-seed scatter of the exact build itself
    for (n_e, ns_e, _), (n_h, ns_h, _), (n_2, ns_2, _) in zip(ex, hw, e2):
        if ns_e > 2000:
            noise = abs(ns_2 - ns_e) / ns_e
            assert abs(ns_h - ns_e) / ns_e < max(4 * noise, 4 / np.sqrt(ns_e)), (ns_e, ns_h, ns_2)
    se, sh, s2 = (np.asarray(L.view(T, "pxx_flux")) for T in (Te, Th, T2))
    scale = np.abs(se).max()
    noise = np.abs(s2 - se).max() / scale
    assert np.abs(sh - se).max() / scale < max(3 * noise, 1e-3), (np.abs(sh - se).max() / scale, noise)


@pytest.mark.parametrize("case", ["protons", "mixed_species", "fp32_state"])
def test_fused_pcut_loop_equals_per_pcut_calls(case):
    """mcs_run_pcuts_fused queues a species' whole pcut loop on the device: n_saved (count of the l_save flags, cross-checked
    against the kernel's own counter on the device), i_mult = max(n_target / n_saved, 1) (src/cuts.jl:42) and the size of the next
    population are decided there, the next transport launch reads its population size from device memory -- one read-back per
    species instead of one per pcut.  Through the same driver with and without it: the same populations, saved counts and i_mult in
    every pcut, the same integer tallies, binned tallies up to add order; the final populations are equal bit for bit."""
    kw = {}
    if case == "mixed_species":
        kw = dict(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1),
                           mcs.inputs.Species(mcs.constants.ME / mcs.constants.MP, -1.0, 1e6, 1.2)], energy_transfer_frac=0.1, radiation_losses=True)
    if case == "fp32_state":
        kw = dict(state_fp32=True)
    N = 30000
    res = []
    for fused in (True, False):
        prob = make_problem(N, **kw)
        hb = hip_backend(prob)
        r = mcs.driver.run(prob, hb, None, n_itrs=1, fused_pcuts=fused)
        res.append((r, hb.get_population(), hb.layout))
        hb.destroy()
    (ra, pa, L), (rb, pb, _) = res
    sa = [(s.i_ion, s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult) for s in ra.stats]
    sb = [(s.i_ion, s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult) for s in rb.stats]
    assert sa == sb and len(sa) >= 20
    assert np.array_equal(ra.tallies_i64, rb.tallies_i64)
    assert_tallies_close(L, ra.tallies_f64, rb.tallies_f64, TALLY_RTOL)
    assert all(s.kernel_ms > 0 for s in ra.stats)
