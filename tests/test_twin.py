"""The hot-path oracle against an independent restatement (tests/twin/mcs_twin.py, plain Python written from the Julia
files, glibc math, its own Philox).  The reference holds no vectors for this path, so this is the strongest pin available:
two restatements by different routes -- C++ via oracle/mcs_oracle.cpp (the thing every GPU test is compared with) and
Python via the twin -- must agree on every discrete outcome and to 1e-11 on momenta, positions and tallies.

The libm build of the oracle is used (the twin's math is glibc through Python's `math`); the deterministic-math build,
which the GPU reproduces bit for bit, is tied to the libm build by tests/test_golden.py."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, mcs, make_problem, oracle_backend

sys.path.insert(0, os.path.join(ROOT, "tests", "twin"))
import mcs_twin  # noqa: E402

FIELDS = ("weight", "ptot_pf", "pb_pf", "x_PT_cm", "xn_per", "prp_x_cm", "acctime_sec", "phi_rad", "grid", "tcut", "downstream", "inj")
RTOL = 1e-11


def _pop_dict(pop):
    return {f: getattr(pop, f).copy() for f in FIELDS}


def _close(a, b, what, rtol=RTOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = np.maximum(np.abs(b), 1e-300)
    err = np.max(np.abs(a - b) / scale) if a.size else 0.0
    assert err <= rtol, f"{what}: max relative difference {err:.3e}"


def _to_population(pop):
    n = len(pop["weight"])
    p = mcs.capi.Population(n)
    for f in FIELDS:
        getattr(p, f)[:] = pop[f]
    return p


def _begin_species(prob, ob, i_ion, first):
    """What main_loops does at the top of a species (main_loops.jl:97-164), on the oracle backend; returns the twin for it
    and the snapshots against which increments are compared (the oracle's buffers carry the 1e-99 floors and the analytic
    fast-push fluxes of begin_iteration / begin_species; its counters run on over the species)."""
    cfg = prob.cfg
    sp = cfg.species[i_ion - 1]
    if first:
        ob.begin_iteration(1)
    inj = mcs.inputs.init_pop_host(prob, i_ion)
    pmax = mcs.inputs.get_pmax_cutoff(prob.Emax_keV, prob.Emax_per_aa_keV, prob.pmax, sp.aa)
    ewf = 1.0 / cfg.species[-1].density if cfg.species[-1].density else float("inf")
    ob.begin_species(1, i_ion, sp.aa, abs(sp.zz), pmax, sp.density, ewf)
    ob.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
    ob.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
    To, Io = ob.read_tallies()
    tw = mcs_twin.Twin(prob, 1, i_ion, sp.aa, abs(sp.zz), pmax, sp.density, ewf)
    tw.T["energy_recv_pool"][...] = ob.layout.view(To, "energy_recv_pool")       # an INPUT of the species (main_loops.jl:164)
    tw_base = {k: v.copy() for k, v in tw.T.items()}
    return tw, tw_base, To.copy(), Io.copy()


def _compare_species(prob, ob, tw, bases, pop, n_pcuts, N_target, RTOL=RTOL, first_pcut=1):
    """Run `n_pcuts` pcuts of the population on the oracle backend `ob` and on the twin `tw`, comparing after every pcut."""
    tw_base, To_base, Io_base = bases
    L = ob.layout
    steps = 0
    for ip in range(first_pcut, first_pcut + n_pcuts):
        ob.set_population(_to_population(pop))
        ns = ob.run_pcut(ip, 0)
        fo = ob.finals()
        so, lo = ob.get_saved()
        ft, saved_rows = tw.run_pcut(ip, pop)
        # discrete outcomes: exit reason, step counts, saved flags
        for k in ("reason", "helix", "retro"):
            assert np.array_equal(ft[k], fo[k]), f"pcut {ip}: {k} differs for particles {np.flatnonzero(ft[k] != fo[k])[:8]}"
        assert np.array_equal(np.array([r is not None for r in saved_rows]), lo == 1)
        _close(ft["ptot"], fo["ptot"], f"pcut {ip}: final ptot_pf", RTOL)
        # (positions: sums of steps of either sign, so the error of one is a multiple of an ulp of the LARGEST position the
        # history visited, not of where it ended; bounded relative to the step scale of the population)
        assert np.max(np.abs(ft["x"] - fo["x"])) <= RTOL * np.max(np.abs(fo["x"])), f"pcut {ip}: final x"
        for j, r in enumerate(saved_rows):
            if r is None:
                continue
            for f in FIELDS:
                v = getattr(so, f)[j]
                if f in ("grid", "tcut", "downstream", "inj"):
                    assert int(r[f]) == int(v), (ip, j, f)
                elif f == "phi_rad":
                    # the phase takes asin(s) with |s| clamped just below 1 (scattering.jl:93-101), where d asin / ds ~ 1e4 .. 1e8:
                    # one ulp of difference between two libm call sequences becomes 1e-12 .. 1e-9 rad.  It does not feed back
                    # into momenta or positions for a parallel field; bounded absolutely
                    assert abs(r[f] - v) <= 1e-8, (ip, j, f, r[f], v)
                elif f == "x_PT_cm":
                    assert abs(r[f] - v) <= RTOL * float(np.max(np.abs(fo["x"]))), (ip, j, f, r[f], v)
                else:
                    assert abs(r[f] - v) <= RTOL * max(abs(v), 1e-300), (ip, j, f, r[f], v)
        steps += int(fo["helix"].astype(np.int64).sum())
        if ns == 0:
            break
        im = max(N_target // ns, 1)
        pop = mcs_twin.split_population(saved_rows, im)
    # tallies of the species: increments since the species began
    To, Io = ob.read_tallies()
    ng = prob.n_grid
    assert np.array_equal(tw.num_crossings, Io[:ng])
    for name, k in mcs.capi.IC.items():
        assert tw.cnt[name] == int(Io[ng + k] - Io_base[ng + k]), (name, tw.cnt[name], int(Io[ng + k] - Io_base[ng + k]))
    for name in L.offsets:
        a = tw.T[name] - tw_base[name]
        d = L.view(To, name) - L.view(To_base, name)
        scale = float(np.max(np.abs(d)))
        if scale == 0.0:
            assert not np.any(a), name
            continue
        err = float(np.max(np.abs(a - d))) / scale
        assert err <= RTOL, f"tally {name}: max|twin - oracle| / max|oracle| = {err:.3e}"
    return steps


def test_physical_constants_agree():
    c = mcs.constants
    assert (mcs_twin.MP, mcs_twin.ME, mcs_twin.C, mcs_twin.QCGS) == (c.MP, c.ME, c.C, c.QCGS)


def test_philox_known_answers():
    """Random123's published known-answer vectors for Philox4x32-10 (the twin's own implementation)."""
    assert mcs_twin.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert mcs_twin.philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert mcs_twin.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_protons_through_the_first_pcuts():
    """300 thermal protons, the unmodified shock of BASELINE config[1], pcuts 1-6 (the first heavy pcut is number 5:
    ~1500 passes per particle with shock crossings, PRP returns with retro walks, time cuts, zone-crossing tallies)."""
    N = 300
    prob = make_problem(N)
    ob = oracle_backend(prob, math="libm", nthreads=1)
    tw, *bases = _begin_species(prob, ob, 1, True)
    pop = _pop_dict(ob.get_population())
    steps = _compare_species(prob, ob, tw, bases, pop, 6, N)
    assert steps > 200_000 and tw.cnt["STEPS_RETRO"] > 0 and tw.T["psd"].sum() > 0 and tw.num_crossings.sum() > 0
    ob.destroy()


def test_mixed_species_branches():
    """Every optional branch of the path on 60 particles per species: protons then electrons, radiative losses, ion ->
    electron energy transfer through the pool, x_spec detectors, injection probability < 1 (the no-DSA retry loop), a
    compressed downstream field -- the configuration of the golden case `mixed_n96`, three pcuts per species."""
    ME_MP = mcs.constants.ME / mcs.constants.MP
    N = 60
    prob = make_problem(N, species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(ME_MP, -1.0, 1e6, 1.0)],
                        energy_transfer_frac=0.1, radiation_losses=True, XSPEC=[-0.5, 0.05, 2.0], INJFR=[0.7, 1.0],
                        b_field_turbulence=1.0, shock_speed=3.0)
    ob = oracle_backend(prob, math="libm", nthreads=1)
    for i_ion in (1, 2):
        tw, *bases = _begin_species(prob, ob, i_ion, i_ion == 1)
        if i_ion == 2:
            assert tw.T["energy_recv_pool"].sum() > 0          # the ions donated
        pop = _pop_dict(ob.get_population())
        _compare_species(prob, ob, tw, bases, pop, 6 if i_ion == 1 else 2, N)
        if i_ion == 1:
            assert tw.T["energy_transfer_pool"].sum() > 0 and tw.T["spectra_sf"].sum() > 0
            assert tw.cnt["RNG_DRAWS"] > 2 * tw.cnt["STEPS_HELIX"]       # single draws were taken: the injection test of no_DSA_loop, prob_return
        else:
            assert tw.cnt["HELIX_CAP"] > 0                                # thermal electrons end at the cap (quirk Q5)
    ob.destroy()


def test_oblique_field_and_modified_profile():
    """theta_B != 0, u_z != 0 and a precursor in u_x (every upstream zone crossing goes through transform_p_PSP with the general
    boosts, transformers.jl:523-607; the gyro term of the move is non-zero): 80 protons through five pcuts."""
    N = 80
    prob = make_problem(N)
    x = prob.x_grid_cm
    up = x < 0
    ux = prob.ux.copy()
    ux[up] = prob.ux[1] * (1 - 0.3 * np.exp(x[up] / (50.0 * prob.rg0)))
    prob.ux = ux
    prob.theta = np.where(up, 0.35, 0.8)
    prob.uz = 0.05 * prob.ux
    prob.utot = np.hypot(prob.ux, prob.uz)
    prob.gam_sf = 1 / np.sqrt(1 - (prob.utot / mcs.constants.C) ** 2)
    ob = oracle_backend(prob, math="libm", nthreads=1)
    tw, *bases = _begin_species(prob, ob, 1, True)
    pop = _pop_dict(ob.get_population())
    # (with an oblique field the phase enters the move -- r_g sin(theta_B) (cos phi - cos phi_old) -- and the frame transforms, so
    # the asin amplification of the phase (see _compare_species) reaches positions and momenta: 1e-8 instead of 1e-11; the
    # discrete outcomes -- exit reasons, step counts of every history, saved flags, counters -- are still required to be EQUAL)
    _compare_species(prob, ob, tw, bases, pop, 5, N, RTOL=1e-8)
    assert tw.T["pxz_flux"].sum() > 0 and tw.T["therm_pf"].sum() > 0
    ob.destroy()


def test_crafted_protons_every_exit():
    """Hand-placed protons that leave by every door: upstream of the FEB and injected (reason 2 at once: esc_flux, esc_psd_up,
    the escaping-flux scalars of all_flux.jl:155-158 when they cross the FEB in a move), above p_max (reason 2 through the
    shock-frame test), older than age_max (reason 3), far downstream beyond 6.91 L_diff (reason 1), past the last time cut
    (D4), with the downstream FEB set (reason 1 through downstream_test's first branch)."""
    N = 48
    prob = make_problem(N, FEB_downstream=(40.0, 0.0), maximum_energy=(0.0, 0.0, 1e4))
    P = prob.params
    rng = np.random.default_rng(7)
    mpc = mcs.constants.MP * mcs.constants.C
    pop = {f: np.zeros(N) for f in FIELDS}
    kind = np.arange(N) % 6
    pop["weight"][:] = 1.0 / N
    pop["ptot_pf"][:] = mpc * 10 ** rng.uniform(0.5, 2.5, N)
    pop["ptot_pf"][kind == 1] = mpc * 2.0e4                                   # above p_max
    mu = rng.uniform(-1, 1, N)
    mu[kind == 0] = -0.99995                  # heading upstream against the 0.98 c flow (needs mu < -0.98 and gamma >> 1), towards the FEB
    pop["ptot_pf"][kind == 0] = mpc * 300.0
    pop["pb_pf"][:] = pop["ptot_pf"] * mu
    x_rg = 10 ** rng.uniform(-2, 0.8, N)
    x_rg[kind == 0] = -99.99 - 0.005 * rng.uniform(0, 1, (kind == 0).sum())    # just inside the upstream FEB (-100 rg0)
    x_rg[kind == 3] = 35.0 + rng.uniform(0, 4, (kind == 3).sum())              # near the downstream FEB (40 rg0)
    pop["x_PT_cm"][:] = x_rg * prob.rg0
    pop["grid"][:] = np.searchsorted(prob.x_grid_cm, pop["x_PT_cm"], side="right") - 1
    pop["downstream"][:] = 1
    pop["inj"][:] = (kind == 0) | (kind == 4)
    pop["xn_per"][:] = P.xn_per_fine
    pop["prp_x_cm"][:] = P.x_grid_stop
    pop["acctime_sec"][:] = 10 ** rng.uniform(0, 6, N)
    pop["acctime_sec"][kind == 2] = 4.0e11                                     # older than age_max = 3.15e11 s
    pop["phi_rad"][:] = rng.uniform(0, 2 * np.pi, N)
    pop["tcut"][:] = 1
    pop["tcut"][kind == 5] = len(prob.tcuts) + 1                               # past the last time cut
    P.age_max = 3.15e11
    ob = oracle_backend(prob, math="libm", nthreads=1)
    tw, *bases = _begin_species(prob, ob, 1, True)
    # (pcut 12: p > 1000 m_p c -- nobody is saved at once, the histories run into their exits; then pcut 13 for the few that are saved)
    _compare_species(prob, ob, tw, bases, pop, 2, N, first_pcut=12)
    c = tw.cnt
    assert c["REASON2"] >= 8 and c["REASON3"] >= 8 and c["REASON1"] >= 1 and c["TCUT_OVERRUN"] > 0, c
    assert tw.T["esc_flux"][0] > 0 and tw.T["esc_psd_up"].sum() > 0 and tw.T["esc_psd_down"].sum() > 0
    assert abs(tw.T["scalars"][2]) > 0 and tw.T["scalars"][3] > 0               # the FEB crossing inside a move (quirk Q1)
    ob.destroy()


def test_crafted_electrons_losses_and_prp_shortening():
    """The hand-placed relativistic electrons of the golden case `electrons_crafted_n64` (strong field): radiative losses in
    the helix loop and inside retro_time, the electron PRP shortening (prob_return.jl:155-164), the constant-mfp branch
    below p_e,crit, zero-energy exits."""
    from golden_common import make_golden
    prob, spec = make_golden.build_case("electrons_crafted_n64")
    N = 24
    ob = oracle_backend(prob, math="libm", nthreads=1)
    tw, *bases = _begin_species(prob, ob, 1, True)
    pop_full = make_golden.crafted_population("electrons", prob, 64)
    pop = {f: getattr(pop_full, f)[:N].copy() for f in FIELDS}
    _compare_species(prob, ob, tw, bases, pop, 2, N)
    assert tw.cnt["REASON4"] + tw.cnt["HELIX_CAP"] + tw.cnt["STEPS_RETRO"] > 0
    ob.destroy()


@pytest.mark.parametrize("kind", ["protons", "general", "electrons", "oblique"])
def test_fuzzed_caller_populations_oracle_vs_twin(kind):
    """The random caller-provided populations of the GPU test test_fuzzed_caller_populations_vs_oracle (conftest.fuzz_population:
    every combination of downstream / inj, positions from beyond the upstream FEB to downstream of x_grid_stop, the PRP on either
    side, ages around age_max, every time-cut index), 120 particles, through the oracle and the twin: the oracle is right about
    the states the path itself never produces -- e.g. the `inj` update after the first move of a downstream-flagged particle
    loaded at x < 0 (particle_loop.jl:433-435), which the HIP path once missed."""
    from conftest import fuzz_population, fuzz_problem
    N = 120
    prob, aa = fuzz_problem(kind, N)
    ob = oracle_backend(prob, math="libm", nthreads=1)
    tw, *bases = _begin_species(prob, ob, 1, True)
    pop_full = fuzz_population(prob, N, 1, aa)
    assert int(((pop_full.downstream == 1) & (pop_full.inj == 0) & (pop_full.x_PT_cm < 0)).sum()) >= 3
    pop = {f: getattr(pop_full, f).copy() for f in FIELDS}
    _compare_species(prob, ob, tw, bases, pop, 2, N, first_pcut=2)
    assert tw.cnt["REASON1"] > 0 and tw.cnt["REASON2"] + tw.cnt["REASON3"] > 0
    ob.destroy()
