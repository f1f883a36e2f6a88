#!/usr/bin/env python3
"""bench.py -- headline benchmark of the transport path (BASELINE.json):
particle-scatter steps/s and wall time per iteration, 10^6 protons per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full iteration (`i_iter` body of src/main_loops.jl:52-391) of the
workload: BASELINE config[1] -- 10^6 protons per GPU, single unmodified gamma0 = 5
shock, fp64, the 45 stock pcuts, scattering and DSA on -- i.e. init_pop (K3), then
for every pcut the transport kernel (K1) + compaction/splitting (K2), then the
merge of the tallies.  Weak scaling: every rank carries 10^6 particles of ONE global
population of N x 10^6, dealt out like cards (rank r holds global particles r, r + N, ...; global RNG keys); per
pcut an all-gather of n_saved and of the saved particles' global indices (4 B each; the particles themselves only in
the late pcuts, when few are saved), one sum-all-reduce of the tallies per species (RCCL).  Synthetic data: the thermal
injection of the reference's own initialiser.  `value` = total (helix + retro) steps
of all ranks / max-over-ranks wall time of the K timed steps.

Extra objects in the JSON line:
  roofline     the transport kernel against the fp64 VALU peak (78.6 TFLOP/s): the path
               is scalar fp64 arithmetic, neither HBM- nor MFMA-bound;
               achieved = 400 algorithmic flop/step x steps / kernel time (HIP events on
               the kernel's stream, summed over the launches of the timed region).
  cpu_baseline the CPU oracle (C++ restatement, glibc libm, OpenMP over particles on all
               host cores) on a bounded sample of the same workload -- a surrogate for
               the Julia reference, which cannot run here; `single_thread` inside it is
               the same code on one core (the reference itself is serial), smaller sample.
  load_balance (N > 1) the largest local population over the mean, per pcut: worst and
               step-weighted mean over the timed iterations.
  overlapped_iterations  (N = 1) a separate leg AFTER the timed region, never part of `value`: the same iterations
               with two of them in flight on two contexts / streams (driver.run_overlapped) -- possible because the
               workload's shock profile is fixed, so its iterations are independent; --overlap 1 skips it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP = 400.0          # weighted algorithmic fp64 flop per step
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X: 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz


def cpu_baseline(mcs, n_sample, n_itrs=1, n_sample_1t=None):
    """Timed CPU leg (rank 0, N=1 only): the oracle is used here as the reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))           # the GPU box gives one GPU a share of the host cores

    def timed(n, threads):
        cfg = mcs.inputs.Config(N_PTS_INJ=n, N_PTS_PCUT=n, N_PTS_PCUT_HI=n)
        prob = mcs.inputs.build_problem(cfg)
        be = orc.OracleBackend(mcs.capi, "libm", nthreads=threads)
        be.create(prob)
        t0 = time.perf_counter()
        res = mcs.driver.run(prob, be, None, n_itrs=n_itrs)
        dt = time.perf_counter() - t0
        be.destroy()
        return res.steps_helix + res.steps_retro, dt

    steps, dt = timed(n_sample, cores)
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    out = {"value": steps / dt, "unit": "particle-scatter steps/s", "cores": cores, "kind": "port",
           "sample": f"{n_sample} protons, all 45 pcuts, {n_itrs} iteration, {steps} steps in {dt:.1f} s; "
                     f"C++ surrogate of the Julia reference (glibc libm, OpenMP dynamic over particles); cpu: {model}"}
    if n_sample_1t:
        s1, d1 = timed(n_sample_1t, 1)
        out["single_thread"] = {"value": s1 / d1, "unit": "particle-scatter steps/s", "cores": 1,
                                "sample": f"{n_sample_1t} protons, all 45 pcuts, {s1} steps in {d1:.1f} s (the reference is serial)"}
    return out


def overlapped_leg(mcs, hip_backend, prob, be0, local, args):
    """NOT the headline: the same `steps` iterations with `--overlap` of them in flight at a time (one context, stream and
    host thread each), timed like the main region.  Legitimate only because the workload's shock profile is fixed (the stock
    smooth-shocks = false): its iterations are independent realisations, and the tails of one launch are filled by the blocks
    of another's (driver.run_overlapped).  `value` above stays the one-after-the-other figure, as the reference runs."""
    import torch
    bes = [be0] + [hip_backend.HipBackend(local) for _ in range(args.overlap - 1)]
    for b in bes[1:]:
        b.create(prob)
    first = args.steps + args.warmup + 1
    n = -(-args.steps // len(bes)) * len(bes)          # whole rounds: every context carries the same number of iterations
    mcs.driver.run_overlapped(prob, bes, n_itrs=len(bes), first_iter=first)            # warm every context
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = mcs.driver.run_overlapped(prob, bes, n_itrs=n, first_iter=first + len(bes))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = sum(s for _, _, s in res.local_steps)
    for b in bes[1:]:
        b.destroy()
    return {"in_flight": len(bes), "value": steps / dt, "unit": "particle-scatter steps/s", "ms_per_step": dt / n * 1e3,
            "steps": n, "note": "independent iterations (fixed shock profile) sharing the GPU; not the headline value"}


def pipelined_leg(mcs, prob, be, args, first_iter, sm):
    """NOT the headline: the same `steps` iterations with the pcuts pipelined (mcs_run_pcuts_pipelined: a pcut's long histories finish
    beside the next pcut on CU-masked streams; the next population is ordered not-long before long, the oracle likewise --
    tests/test_pipelined_pcuts.py), timed like the main region.  Reported beside it because the per-launch roofline of the headline has
    no meaning for launches that share the chip."""
    import torch
    kw = dict(smoothing=sm, species_tallies=args.species_tallies, long_draws=args.long_draws)
    mcs.driver.run(prob, be, n_itrs=1, first_iter=first_iter, **kw)                        # warm (second stream, buffers)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = mcs.driver.run(prob, be, n_itrs=args.steps, first_iter=first_iter + 1, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = sum(s for _, _, s in res.local_steps)
    return {"long_draws": args.long_draws, "long_imult_max": int(os.environ.get("MCS_LONG_IMULT_MAX", "8")), "value": steps / dt,
            "unit": "particle-scatter steps/s", "ms_per_step": dt / args.steps * 1e3, "steps": args.steps,
            "main_launch_ms_per_step": sum(s.kernel_ms for s in res.stats) / args.steps,
            "note": "one iteration after the other, every pcut's long histories beside the next pcut; not the headline value"}


def workload_label(args):
    """What actually ran, built from the arguments: every kept bench line names its own workload."""
    n = args.particles
    size = f"{n:.0e}".replace("e+0", "e").replace("e+", "e") if n >= 1000 and n == float(f"{n:.0e}") else str(n)
    if n == 1_000_000 and not args.smooth:
        head = "BASELINE config[1]: "
    elif n == 10_000_000 and args.smooth:
        head = "BASELINE config[2] (evolving profile): "
    elif n == 10_000_000:
        head = "BASELINE config[2]'s population on a FIXED profile: "
    else:
        head = "non-headline size: "
    profile = ("shock profile replaced after every iteration (smooth_grid_par; every zone crossing takes the frame transform)"
               if args.smooth else "single unmodified gamma0=5 shock")
    what = "protons"
    if args.mixed:
        head = "BASELINE config[4]'s species mix: "
        what = "particles of each of protons, He (density 0.1) and electrons (radiative losses on, ion -> electron energy transfer 0.1)"
    prec = "fp32 particle state (the fp32-state variant; tallies fp64)" if args.fp32 else "fp64"
    if args.fp32 and not args.mixed:
        head = "fp32-state variant of " + head
    return (f"{head}{size} {what} per GPU, {profile}, 45 stock pcuts, scattering+DSA on, {prec}; one step = one full iteration "
            f"(per species: init_pop, 45 x (transport + new_pcut), tally merge [{args.species_tallies} read-back per species], ion_finalize consumers; "
            f"iter_finalize" + (" + profile update" if args.smooth else "") + ")")


KERNEL_NAMES = {0: "mcs_k_transport", 1: "mcs_k_transport_plain", 2: "mcs_k_transport_lossy", 3: "mcs_k_transport_f32", 4: "mcs_k_transport_f32_loop",
                5: "mcs_k_transport_f32_lossy", 6: "mcs_k_transport_plain_etf", 7: "mcs_k_transport_ws", 8: "mcs_k_transport_ws_etf",
                9: "mcs_k_transport_f32_loop_exact", 10: "mcs_k_transport_sliced", 11: "mcs_k_transport_plain_sliced", 12: "mcs_k_transport_lossy_sliced",
                13: "mcs_k_transport_plain_etf_sliced"}


def kernel_label(args, last_kernel=None):
    if last_kernel in (7, 8) and not args.mixed and not args.fp32:
        return (KERNEL_NAMES[last_kernel] + " (the wave-specialised form of the common configuration's kernel: the library picks it for populations of "
                "at least MCS_WS_AUTO_MIN = 6e6 particles, mcs_k_transport_plain below; mcs_last_kernel of the last launch)")
    if args.fp32:
        return "mcs_k_transport_f32" + (" (ions) + mcs_k_transport_f32_lossy (electrons)" if args.mixed else "") + ": the fp32-state variant; priced against the fp64 peak by the same 400-flop weight"
    if args.mixed:
        return "mcs_k_transport_plain_etf (ions: energy transfer on) + mcs_k_transport_lossy (electrons with radiative losses); all launches of the timed region"
    return "mcs_k_transport_plain (the specialisation of mcs_k_transport for this configuration)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--particles", type=int, default=1_000_000, help="particles per GPU")
    ap.add_argument("--cpu-sample", type=int, default=60000)
    ap.add_argument("--cpu-sample-1t", type=int, default=3000, help="particles of the single-thread CPU leg (0: skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--species-tallies", choices=("light", "full"), default="light",
                    help="what the host fetches at every species end: the 0.8 MB it computes with (light) or the whole 61 MB buffer (full)")
    ap.add_argument("--overlap", type=int, default=3,
                    help="extra leg, reported beside `value` and never in it: this many independent iterations in flight (driver.run_overlapped); 1 = skip")
    ap.add_argument("--long-draws", type=int, default=6144,
                    help="extra leg at N=1 (not the headline): the pcuts pipelined, long histories = this many random draws (0: skip)")
    ap.add_argument("--smooth", action="store_true", help="replace the shock profile after every iteration (smooth_grid_par): config[2]'s loop")
    ap.add_argument("--mixed", action="store_true", help="BASELINE config[4]'s species mix: protons + He + electrons, radiative losses, ion -> electron energy transfer (--particles per species)")
    ap.add_argument("--fp32", action="store_true", help="the fp32-state variant of the transport kernel (config[4]); tallies stay fp64")
    args = ap.parse_args()

    import torch
    import _mcs_loader
    mcs = _mcs_loader.load()
    from mcs_amd import hip_backend

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    # Rehearsal knobs (tests of the N > 1 code path on a one-GPU box): every rank on one device, gloo
    # instead of RCCL.  Never set by the driver.
    if "MCS_BENCH_ONE_DEVICE" in os.environ:
        local = int(os.environ["MCS_BENCH_ONE_DEVICE"])
    backend = os.environ.get("MCS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    # MCS_BENCH_FORCE_COMM=1 (rehearsal, never set by the driver): run the multi-rank code path -- RCCL process
    # group, bound torch tally tensors, all-gather(n_saved) per pcut, in-place all-reduce -- with whatever world size
    # there is, 1 included: a one-GPU box cannot host two RCCL ranks, but it can run that path once.
    force_comm = os.environ.get("MCS_BENCH_FORCE_COMM") == "1"
    if world > 1 or force_comm:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n_global = args.particles * world
    n_itrs = args.steps + args.warmup
    # (room in the per-iteration tallies for the iterations of the extra overlapped leg, which carry on the numbering)
    overlap_leg = world == 1 and not force_comm and args.overlap > 1 and not args.smooth and not args.mixed and not args.fp32
    n_extra = (2 * args.overlap + args.steps) if overlap_leg else 0
    pipe_leg = world == 1 and not force_comm and args.long_draws > 0 and not args.fp32 and not args.smooth
    pipe_first = n_itrs + n_extra + 1
    if pipe_leg:
        n_extra += 1 + args.steps
    kw = {}
    if args.mixed:
        me_mp = mcs.constants.ME / mcs.constants.MP
        kw = dict(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1), mcs.inputs.Species(me_mp, -1.0, 1e6, 1.2)],
                  energy_transfer_frac=0.1, radiation_losses=True)
    cfg = mcs.inputs.Config(N_PTS_INJ=n_global, N_PTS_PCUT=n_global, N_PTS_PCUT_HI=n_global, num_iterations=n_itrs + n_extra,
                            state_fp32=args.fp32, **kw)
    prob = mcs.inputs.build_problem(cfg)
    be = hip_backend.HipBackend(local, torch_tallies=world > 1 or force_comm)
    be.create(prob)
    comm = mcs.driver.Comm(world > 1 or force_comm, dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    state = {"steps": 0, "kernel_ms": 0.0, "t": None}
    marks = {}

    def on_species_end(i_iter, i_ion, G_f, G_i):
        ng = prob.n_grid
        IC = mcs.capi.IC
        marks[i_iter] = (time.perf_counter(), int(G_i[ng + IC["STEPS_HELIX"]] + G_i[ng + IC["STEPS_RETRO"]]))

    def on_iteration_end(i_iter):
        # the iteration boundary: after ion_finalize (K4: dN/dp, pressures) and iter_finalize (+ the profile update)
        if i_iter == args.warmup:
            barrier()
            marks["t0"] = time.perf_counter()

    if args.warmup == 0:
        barrier()
        marks["t0"] = time.perf_counter()
    # the whole i_iter body of src/main_loops.jl:52-391 is timed: init_pop, every pcut (K1 + K2), the merge of the tallies,
    # ion_finalize's consumers on the device (K4) and iter_finalize; --smooth also replaces the shock profile after every
    # iteration (smooth_grid_par), as in BASELINE config[2] -- the headline keeps the single unmodified shock of config[1]
    # (the stock mc_in.toml has smooth-shocks = false)
    sm = mcs.iter_finalize.SmoothingConfig(smooth_shocks=args.smooth)
    # (--species-tallies light, the default: at a species end the host takes the 0.8 MB of the tally buffer it computes with --
    # fluxes, escape and coupled spectra, pools, scalars, the int64 tallies; the three 20 MB histograms stay where their
    # consumers run (K4) and are fetched once, after the last iteration.  "full" moves all 61 MB every iteration.)
    res = mcs.driver.run(prob, be, comm, n_itrs=n_itrs, on_species_end=on_species_end, on_iteration_end=on_iteration_end,
                         smoothing=sm, species_tallies=args.species_tallies)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - marks["t0"]
    steps_total = marks[n_itrs][1] - (marks[args.warmup][1] if args.warmup > 0 else 0)
    # every transport launch of the timed iterations, the ones a fused species loop makes on an empty population included (they
    # are launches of the same kernel: the rocprofv3 kernel trace averages over them too)
    kern_ms = sum(s.kernel_ms for s in res.stats if s.i_iter > args.warmup) + sum(e[3] for e in res.empty_launches if e[0] > args.warmup)
    n_launch = sum(1 for s in res.stats if s.i_iter > args.warmup) + sum(1 for e in res.empty_launches if e[0] > args.warmup)
    # local steps of this rank in the timed region (for the per-kernel roofline)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # HBM bytes per K1 launch: NOT measured by this process -- the committed result of the separate
    # `rocprofv3 --pmc` passes over this same command (tools/profile_bench.sh), labelled as such
    traffic, traffic_src = None, None
    for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                traffic = json.load(f)["hbm_bytes_per_launch"]
            traffic_src = f"static: profiles/{name} (rocprofv3 --pmc passes of this command at the commit named there, not this run)"
            break
        except Exception:
            pass
    # this rank's own steps in the timed region (the kernel-level roofline is per GPU); rank 0 reports its own
    local_steps = sum(n for (it, _, n) in res.local_steps if it > args.warmup)
    timed = [s for s in res.stats if s.i_iter > args.warmup]
    skew = [s.n_use_max * world / max(s.n_pts_use, 1) for s in timed]
    if rank == 0:
        value = steps_total / elapsed
        ach = local_steps * FLOP_PER_STEP / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        out = {
            "metric": "particle-scatter steps/sec + wall-time per iter, 10^6 particles, 1->8 MI355X",      # (BASELINE.json's metric name, whatever --particles says: config.workload names what ran)
            "value": value, "unit": "particle-scatter steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 particle state, f64 tallies" if args.fp32 else "f64", "data": "synthetic",
            "config": {"workload": workload_label(args),
                       "particles_per_gpu": args.particles, "particles_total": n_global, "species_tallies": args.species_tallies,
                       "steps_per_iteration": steps_total / args.steps,
                       "parallelism": f"interleaved particle shards x{world}; per pcut all-gather(n_saved) + all-gather(saved global indices, 4 B each) [or of the saved particles when few]; all-reduce(tallies) per species"},
            "roofline": {"bound": "fp64_valu", "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / FP64_VALU_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_label(args, be.last_kernel()), "launches": n_launch,
                         "avg_launch_ms": kern_ms / max(n_launch, 1),
                         "kernel_steps_per_s": local_steps / (kern_ms * 1e-3) if kern_ms > 0 else 0.0,
                         "note": "400 algorithmic fp64 flop/step (SURVEY 8d) x steps / HIP-event kernel time; "
                                 "HBM traffic is << 1 B/step (68 B in + 69 B out per particle per pcut)"},
        }
        out["load_balance"] = {"max_over_mean_worst": max(skew), "max_over_mean_time_weighted":
                               sum(k * s.kernel_ms for k, s in zip(skew, timed)) / max(sum(s.kernel_ms for s in timed), 1e-30),
                               "gather_splits": sum(1 for s in timed if s.split == "gather"),
                               "local_splits": sum(1 for s in timed if s.split == "local"),
                               "note": "largest local population / mean, per pcut (1.0 = perfectly balanced)"}
        if "MCS_BENCH_ONE_DEVICE" in os.environ:
            # rehearsal of the N > 1 shape on one card: what the device holds with every rank's buffers resident
            free_b, total_b = torch.cuda.mem_get_info()
            out["rehearsal"] = {"ranks_on_one_device": world, "backend": backend, "device_mem_used_gb_at_end": (total_b - free_b) / 1e9,
                                "torch_peak_alloc_gb_rank0": torch.cuda.max_memory_allocated() / 1e9,
                                "note": "all ranks share ONE GPU: rates say nothing about scaling; host-side cost and footprint of the multi-rank path only"}
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(mcs, args.cpu_sample, n_sample_1t=args.cpu_sample_1t)
        if overlap_leg:
            out["overlapped_iterations"] = overlapped_leg(mcs, hip_backend, prob, be, local, args)
        if pipe_leg:
            # (an extra leg must never cost the headline line: a failure is reported in its place)
            try:
                out["pipelined_pcuts"] = pipelined_leg(mcs, prob, be, args, pipe_first, sm)
            except Exception as e:      # noqa: BLE001
                out["pipelined_pcuts"] = {"error": f"{type(e).__name__}: {e}"}
        if args.mixed and world == 1:
            # the photon leg of config[4] ("inverse-Compton/synch photon tallies"): ion_finalize's photon_calcs for the electron
            # species on the histograms the last iteration left on the device (K4 dN/dp, K5 synchrotron, K6 get_dNdp_2D + IC).
            # Reported beside the headline: it is O(bins) work per species, outside the particle loop the metric counts.
            i_e = len(prob.cfg.species)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fin = mcs.consumers.ion_finalize(prob, be, i_e)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            ph_s = mcs.consumers.photon_synch(prob, be, fin, i_e)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            ph_i = mcs.consumers.photon_ic(prob, be, i_e)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            out["photon_leg"] = {"ion_finalize_ms": (t1 - t0) * 1e3, "synch_ms": (t2 - t1) * 1e3, "dndp_2d_plus_ic_ms": (t3 - t2) * 1e3,
                                 "synch_zones_lit": int((ph_s.emis_erg_s > 1e-99).any(axis=1).sum()),
                                 "ic_zones_lit": int((ph_i.emis_erg > 1e-99).any(axis=1).sum()),
                                 "note": "electron species of the last iteration; wall time incl. table upload and the read-back of the spectra"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
