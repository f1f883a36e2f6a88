"""Host driver: the iteration -> species -> pcut nest of the reference's
`main_loops` (src/main_loops.jl:52-391) around the batched transport kernel.

The reference calls `particle_loop` once per particle (main_loops.jl:228-292);
here one backend call runs a whole pcut.  The population stays resident on the
device between pcuts (K1 transport, K2 compaction+split, K3 initial fill);
the host sees one 8-byte count per pcut.

Multi-GPU (one process per GPU, torch.distributed; RCCL over xGMI when the backend is NCCL).  The RNG key of a
particle is (iteration, species, pcut, GLOBAL particle index) whatever GPU it runs on, so the histories -- and every
tally up to the order of the sums -- are those of a one-GPU run.  Baseline fills (1e-99 floors, analytic fast-push
fluxes) live on rank 0 only so that the sum over ranks is the single-GPU result.
  * The injection is dealt out like cards: rank r of W holds global particles r, r + W, r + 2W, ... -- every rank gets
    the same mix of the momentum-sorted injection (mcs_init_pop_binned_strided).  (Round 1's contiguous ranges put all
    the cold particles on rank 0.)
  * Per pcut ONE small collective, all-gather(n_saved, n_local), then `new_pcut` (src/cuts.jl:34-98 rebuilds the
    population from ALL saved particles) in one of two ways:
      "local"   the bulk of a run (n_saved > `gather_max` and max(count) <= `skew_max` x mean): every rank replicates
                ITS saved particles (K2 exactly as on one GPU); no particle leaves its GPU.  The ranks exchange the
                index column only -- all-gather of the saved particles' global indices, 8 B each -- from which each
                rank finds the position of its saved particles in the global order (one searchsorted per peer) and so
                the global indices of their children, position * i_mult + j: the numbering of src/cuts.jl:66-92.  The
                next pcut runs with that index list (mcs_run_pcut_indexed).
      "gather"  few saved particles (late pcuts: a handful, each replicated 10^5 times, which a local split would
                leave on one or two ranks) or counts that drifted apart: all-gather of the saved particles themselves
                (mcs_saved_export, 72 B each), sorted by global index; rank r builds elements r, r + W, ... of the
                global split (mcs_split_import) -- balanced to one particle whatever the counts were, and the shard
                is an arithmetic progression again (mcs_run_pcut_strided).
  * Per species ONE sum-all-reduce of the flat tally buffers (fp64 + int64), on the device under RCCL.
"""
from __future__ import annotations

import dataclasses
import os
import time
from typing import Callable, List, Optional

import numpy as np

from . import inputs
from .inputs import Problem


@dataclasses.dataclass
class PcutStat:
    i_iter: int
    i_ion: int
    i_pcut: int
    n_pts_use: int          # global
    n_saved: int            # global
    i_mult: int
    n_use_max: int          # largest local population of this pcut over the ranks (load balance: max/mean)
    split: str              # how the NEXT population was built: "local" | "gather" | "identity" (all saved, i_mult 1: nothing moves) | "-" (last pcut)
    kernel_ms: float        # local kernel time (HIP events), nan for CPU backends
    wall_ms: float


@dataclasses.dataclass
class RunResult:
    tallies_f64: np.ndarray       # global (all-reduced) flat tallies after the last species
    tallies_i64: np.ndarray
    per_species: list             # [(i_iter, i_ion, f64, i64)] global tallies at each species end
    stats: List[PcutStat]
    steps_helix: int
    steps_retro: int
    iter_finals: list = dataclasses.field(default_factory=list)   # [(i_iter, IterFinal, IonFinal)] when run(finalize=True)
    iter_state: object = None     # iter_finalize.IterState after the last iteration (run(finalize=True))
    local_steps: list = dataclasses.field(default_factory=list)   # [(i_iter, i_ion, helix + retro steps made by THIS rank's kernels)]
    empty_launches: list = dataclasses.field(default_factory=list)   # [(i_iter, i_ion, i_pcut, kernel_ms)]: transport launches of a fused species
                                                                  # loop on an EMPTY population (the pcuts after the one that saved nobody)


class Comm:
    """Thin torch.distributed wrapper (None/1 rank -> no-ops)."""

    def __init__(self, enabled: bool = False, device=None):
        self.enabled = enabled
        self.rank, self.world = 0, 1
        self.device = device
        if enabled:
            import torch.distributed as dist
            self.dist = dist
            self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def all_gather_ints(self, vals) -> List[List[int]]:
        """A few int64 per rank -> [rank][j] (one collective, one device-to-host copy)."""
        vals = [int(v) for v in vals]
        if not self.enabled:
            return [vals]
        import torch
        # NCCL/RCCL needs device tensors; gloo gathers on the host
        dev = self.device if self.dist.get_backend() == "nccl" else None
        t = torch.tensor(vals, dtype=torch.int64, device=dev)
        out = torch.zeros(self.world * len(vals), dtype=torch.int64, device=dev)     # flat: gloo takes no other shape
        self.dist.all_gather_into_tensor(out, t)
        return out.view(self.world, len(vals)).tolist()

    def all_gather_int(self, v: int) -> List[int]:
        return [r[0] for r in self.all_gather_ints([v])]

    def all_gather_cols(self, t, counts):
        """t: [..., cap] on every rank, rank r's first counts[r] columns valid -> [..., sum(counts)], rank-major."""
        import torch
        if not self.enabled or self.world == 1:
            return t[..., :counts[0]].contiguous()
        flat = torch.zeros(self.world * t.numel(), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(flat, t.contiguous().view(-1))
        out = flat.view((self.world,) + tuple(t.shape))
        return torch.cat([out[r][..., :counts[r]] for r in range(self.world)], dim=-1).contiguous()

    def all_gather_rows(self, t):
        """t: [cap] on every rank -> [W, cap]."""
        import torch
        if not self.enabled or self.world == 1:
            return t.view(1, -1)
        flat = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)      # (filled completely by the collective)
        self.dist.all_gather_into_tensor(flat, t.contiguous().view(-1))
        return flat.view(self.world, t.numel())

    def all_reduce_sum_(self, tensor):
        if self.enabled:
            self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM)
        return tensor


def shard_range(n: int, rank: int, world: int):
    """Contiguous, balanced index ranges; rank r gets [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def run(prob: Problem, backend, comm: Optional[Comm] = None, n_itrs: Optional[int] = None,
        max_pcuts: Optional[int] = None, on_species_end: Optional[Callable] = None,
        verbose: bool = False, gather_max: int = 1 << 17, skew_max: float = 1.1,
        finalize: bool = False, smoothing=None, on_iteration_end: Optional[Callable] = None,
        first_iter: int = 1, iter_state=None, species_tallies: str = "full", final_full_read: bool = True,
        before_pcut: Optional[Callable] = None, tcut_print: bool = False, fused_pcuts: bool = True, long_draws: Optional[int] = None,
        long_imult_max: Optional[int] = None) -> RunResult:
    """Run `n_itrs` iterations of all species through all pcuts.

    backend protocol: create/begin_iteration/begin_species/set_fluxes/init_pop/
    run_pcut/new_pcut/export_saved/import_split/pop_size/read_tallies/write_tallies/
    last_kernel_ms (HipBackend in hip_backend.py; tests inject the CPU oracle's).
    gather_max / skew_max: see the module docstring (multi-rank new_pcut).
    finalize: close every iteration as the reference does (src/main_loops.jl:324-391): `ion_finalize`'s dN/dp and
    thermo_calcs on the device-resident histograms (K4, consumers.py) and `iter_finalize` (iter_finalize.py).
    smoothing: an iter_finalize.SmoothingConfig; with smooth_shocks the profile tables of `prob` are replaced after
    every iteration (smooth_grid_par) and uploaded again (mcs_set_grid / mcs_set_cuts) -- BASELINE config[2]'s loop.
    Rank 0 computes the update from the merged tallies and broadcasts the tables, so that every rank transports its
    particles through bit-identical profiles.
    species_tallies: "full" -- every species end hands the whole tally buffer to the host (per_species, on_species_end);
    "light" -- only the part behind the three big histograms (fluxes, escape and coupled spectra, pools, scalars: what
    iter_finalize reads) and the int64 tallies; psd / therm_sf / therm_pf stay on the device, where their consumers run (K4),
    and are fetched once, after the last species of the last iteration (RunResult.tallies_f64 is then complete; with
    final_full_read = False not even then -- run_overlapped fetches every context's buffer once, at the very end).
    tcut_print: replicate the in-place rewrite the reference's `tcut_print` makes at the end of every iteration when time-cut
    tracking is on (src/io.jl:28-45, src/main_loops.jl:383-389): weight_coupled floored, every coupled spectrum normalised
    to a total of 1 and floored -- on the merged tallies, written back to the device (rank 0), so that the next iteration
    accumulates on top of it exactly as the reference does.  Off by default: the tallies then stay plain sums over the
    iterations, which is what the parity fixtures and the overlapped run compare.
    before_pcut(i_iter, i_ion, i_pcut): called before every transport launch (run_overlapped sets the launch geometry there).
    first_iter / iter_state: run iterations first_iter .. first_iter + n_itrs - 1 (the iteration number enters the
    RNG keys and indexes the per-iteration tallies), carrying the iter_finalize state of an earlier call
    (RunResult.iter_state) -- lets a caller step through the loop one iteration at a time.
    """
    import torch

    comm = comm or Comm(False)
    cfg, P = prob.cfg, prob.params
    n_itrs = n_itrs if n_itrs is not None else cfg.num_iterations
    n_pcuts = len(prob.pcuts) if max_pcuts is None else min(max_pcuts, len(prob.pcuts))
    L = backend.layout
    stats: List[PcutStat] = []
    per_species = []
    # rank > 0 keeps only its local partial sums: everything it contributes is a delta
    is_root = comm.rank == 0
    multi = comm.enabled          # (a forced one-rank group runs the multi-rank path too: bench.py MCS_BENCH_FORCE_COMM)
    if long_draws is None:
        long_draws = int(os.environ.get("MCS_LONG_DRAWS", "0")) if not multi else 0
    if long_imult_max is None:
        long_imult_max = int(os.environ.get("MCS_LONG_IMULT_MAX", "8"))
    G_f = G_i = None
    # Live device tensors of the tallies (HIP backend with torch_tallies): the multi-GPU merge
    # then runs in place on the device (RCCL all-reduce, no host round trip).  Otherwise
    # (CPU test backends) the same steps go through read_tallies/write_tallies.
    dev_t = backend.tally_tensors() if (multi and hasattr(backend, "tally_tensors")) else None
    G_pool = None       # merged energy_transfer_pool of the previous species
    local_steps = []    # (i_iter, i_ion, steps this rank's kernels made for that species)
    from .capi import IC as _IC
    i_h, i_r = P.n_grid + _IC["STEPS_HELIX"], P.n_grid + _IC["STEPS_RETRO"]
    # the step counters are never reset: this rank's running total.  A context that has run before (run() called again with
    # first_iter / iter_state, the documented way to step through the loop) starts from what its counters hold now.
    if dev_t is not None:
        steps_seen = int(dev_t[1][i_h].item() + dev_t[1][i_r].item())
    else:
        # (the two counter words only: run_overlapped calls run() once per iteration)
        _i0 = backend.read_counters() if hasattr(backend, "read_counters") else backend.read_tallies()[1]
        steps_seen = int(_i0[i_h] + _i0[i_r])
    iter_finals = []
    empty_launches = []
    if smoothing is not None:
        finalize = True
    if finalize:
        from . import consumers, iter_finalize as itf
        sm = smoothing if smoothing is not None else itf.SmoothingConfig(smooth_shocks=False)
        it_state = iter_state if iter_state is not None else itf.IterState.create(prob, sm, P.n_itrs)

    def tview(t, name):
        o = L.offsets[name]
        return t[o:o + int(np.prod(L.shapes[name]))]

    for i_iter in range(first_iter, first_iter + n_itrs):
        backend.begin_iteration(i_iter)
        if multi and not is_root:
            if dev_t is not None:
                dev_t[0].zero_()
            else:
                f, i = backend.read_tallies()
                f[:] = 0.0
                backend.write_tallies(f, i)
        for i_ion, sp in enumerate(cfg.species, start=1):
            pmax_cutoff = inputs.get_pmax_cutoff(prob.Emax_keV, prob.Emax_per_aa_keV, prob.pmax, sp.aa)
            inj = inputs.init_pop_host(prob, i_ion)
            zz = abs(sp.zz) if cfg.abs_charge else sp.zz
            ewf = 1.0 / cfg.species[-1].density if cfg.species[-1].density != 0 else float("inf")
            backend.begin_species(i_iter, i_ion, sp.aa, zz, pmax_cutoff, sp.density, ewf)
            if is_root:
                backend.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
            if multi:
                if dev_t is not None:
                    if not is_root:   # per-species fills are baselines too
                        for name in ("psd", "esc_psd_up", "esc_psd_down"):
                            tview(dev_t[0], name).zero_()
                    if G_pool is not None:   # ions' donated energy, merged at the previous species end
                        tview(dev_t[0], "energy_recv_pool").copy_(G_pool)
                else:
                    f, i = backend.read_tallies()
                    if not is_root:
                        for name in ("psd", "esc_psd_up", "esc_psd_down"):
                            L.view(f, name)[...] = 0.0
                    if G_pool is not None:
                        L.view(f, "energy_recv_pool")[...] = G_pool
                    backend.write_tallies(f, i)

            n_total = inj.n_pts_use
            # the shard: global index of local particle k = first + k * stride, or gidx[k] after a local split
            first, stride, gidx = comm.rank, comm.world, None
            n_local = (n_total - comm.rank + comm.world - 1) // comm.world if n_total > comm.rank else 0
            if stride == 1:
                backend.init_pop(inj, 0, n_local, n_total)
            else:
                backend.init_pop(inj, first, n_local, n_total, stride)
            p_pcut_hi = inputs.pcut_hi(cfg.EN_PCUT_HI, sp.mass)
            n_use_global = n_total
            # One rank, no per-pcut hook: the whole pcut loop of the species is queued on the device at once -- n_saved, i_mult and
            # the next population's size are decided there (mcs_run_pcuts_fused), one read-back per species instead of one per pcut.
            fused = (fused_pcuts and not multi and before_pcut is None and hasattr(backend, "run_pcuts_fused") and n_pcuts >= 1
                     and os.environ.get("MCS_FUSED_PCUTS", "1") != "0")
            # Long histories told apart (long_draws > 0; MCS_LONG_DRAWS): the next population is ordered non-long before long, which lets a
            # pcut's long histories finish beside the next pcut (mcs_run_pcuts_pipelined; one rank).  A backend without that entry point
            # (the oracle) is told the order (set_long_draws) and runs the ordinary loop: same populations, same streams, same results.
            if long_draws:
                if multi:
                    raise ValueError("long_draws: the pipelined pcut loop and its population order are single-rank (one process per replica)")
                if not hasattr(backend, "run_pcuts_pipelined") and not hasattr(backend, "set_long_draws"):
                    raise ValueError("long_draws: the backend can neither pipeline the pcuts nor order the population by history length")
            pipelined = bool(long_draws) and before_pcut is None and hasattr(backend, "run_pcuts_pipelined") and n_pcuts >= 1 and not getattr(P, "state_fp32", 0)
            if long_draws and not pipelined:
                if not hasattr(backend, "set_long_draws"):
                    raise ValueError("long_draws: this configuration runs the per-pcut loop, and the backend cannot order the population there")
                backend.set_long_draws(int(long_draws))
            i_mult_prev = 0                 # (the rule of mcs_run_pcuts_pipelined for the per-pcut loop: see long_imult_max)
            if pipelined:
                fused = False
                t0 = time.perf_counter()
                targets = [cfg.N_PTS_PCUT if prob.pcuts[ip - 1] < p_pcut_hi else cfg.N_PTS_PCUT_HI for ip in range(1, n_pcuts + 1)]
                n_use_a, n_saved_a, i_mult_a, ms_a, strag_a = backend.run_pcuts_pipelined(1, n_pcuts, targets, int(long_draws), int(long_imult_max))
                wall = (time.perf_counter() - t0) * 1e3
                n_done = n_pcuts
                for ip in range(1, n_pcuts + 1):
                    if int(n_saved_a[ip - 1]) == 0:
                        n_done = ip
                        break
                for ip in range(1, n_done + 1):
                    nu, nsv, im = int(n_use_a[ip - 1]), int(n_saved_a[ip - 1]), int(i_mult_a[ip - 1])
                    stats.append(PcutStat(i_iter, i_ion, ip, nu, nsv, im if nsv > 0 else 0, nu, "-", float(ms_a[ip - 1]), wall / max(n_done, 1)))
                    if verbose and is_root:
                        print(f"[iter {i_iter} ion {i_ion} pcut {ip:2d}] n_use={nu} n_saved={nsv} i_mult={im} kernel={ms_a[ip - 1]:.2f} ms "
                              f"(pipelined: {int(strag_a[ip - 1][0])} long histories exported{', waited' if strag_a[ip - 1][1] else ''})", flush=True)
            if fused:
                t0 = time.perf_counter()
                targets = [cfg.N_PTS_PCUT if prob.pcuts[ip - 1] < p_pcut_hi else cfg.N_PTS_PCUT_HI for ip in range(1, n_pcuts + 1)]
                # (in chunks: a species that ends early -- the thermal electrons in their first pcut -- would otherwise pay ~35 us of
                # empty launches for every remaining pcut; one read-back per chunk of 12 instead of one per pcut)
                chunk = max(1, int(os.environ.get("MCS_FUSED_CHUNK", "12")))
                n_use_a, n_saved_a, i_mult_a, ms_a = [], [], [], []
                for c0 in range(1, n_pcuts + 1, chunk):
                    c1 = min(c0 + chunk - 1, n_pcuts)
                    a_, b_, c_, d_ = backend.run_pcuts_fused(c0, c1, targets[c0 - 1:c1])
                    n_use_a.extend(a_); n_saved_a.extend(b_); i_mult_a.extend(c_); ms_a.extend(d_)
                    if min(b_) == 0:
                        break
                n_done = len(n_use_a)
                wall = (time.perf_counter() - t0) * 1e3
                for ip in range(1, n_done + 1):
                    nu, nsv, im = int(n_use_a[ip - 1]), int(n_saved_a[ip - 1]), int(i_mult_a[ip - 1])
                    last = nsv == 0 or ip == n_pcuts
                    stats.append(PcutStat(i_iter, i_ion, ip, nu, nsv, im if nsv > 0 else 0, nu, "-", float(ms_a[ip - 1]), wall / max(n_done, 1)))
                    if verbose and is_root:
                        print(f"[iter {i_iter} ion {i_ion} pcut {ip:2d}] n_use={nu} n_saved={nsv} i_mult={im} kernel={ms_a[ip - 1]:.2f} ms (fused loop)", flush=True)
                    if last:
                        empty_launches.extend((i_iter, i_ion, jp, float(ms_a[jp - 1])) for jp in range(ip + 1, n_done + 1))
                        break
            for i_pcut in (() if (fused or pipelined) else range(1, n_pcuts + 1)):
                t0 = time.perf_counter()
                if before_pcut is not None:
                    before_pcut(i_iter, i_ion, i_pcut)
                if long_draws and not pipelined:
                    backend.set_long_draws(int(long_draws) if (i_pcut == 1 or long_imult_max <= 0 or i_mult_prev <= long_imult_max) else 0)
                if gidx is not None:
                    n_saved_local = backend.run_pcut_indexed(i_pcut, gidx)
                else:
                    n_saved_local = backend.run_pcut(i_pcut, first, stride)
                gathered = comm.all_gather_ints([n_saved_local, n_local])
                counts = [g[0] for g in gathered]
                n_use_max = max(g[1] for g in gathered)
                n_saved = sum(counts)
                wall = (time.perf_counter() - t0) * 1e3
                # pcut_finalize (src/cuts.jl:100-124)
                i_mult = 0
                if n_saved > 0:
                    n_target = cfg.N_PTS_PCUT if prob.pcuts[i_pcut - 1] < p_pcut_hi else cfg.N_PTS_PCUT_HI
                    i_mult = max(n_target // n_saved, 1)         # new_pcut, src/cuts.jl:42
                i_mult_prev = i_mult
                last = n_saved == 0 or i_pcut == n_pcuts
                local_ok = not multi or (n_saved > gather_max and max(counts) * comm.world <= skew_max * n_saved)
                st = PcutStat(i_iter, i_ion, i_pcut, n_use_global, n_saved, i_mult, n_use_max,
                              "-" if last else ("local" if local_ok else "gather"), backend.last_kernel_ms(), wall)
                stats.append(st)
                if verbose and is_root:
                    print(f"[iter {i_iter} ion {i_ion} pcut {i_pcut:2d}] n_use={n_use_global} (max local {n_use_max}) "
                          f"n_saved={n_saved} i_mult={i_mult} split={st.split} kernel={backend.last_kernel_ms():.2f} ms "
                          f"wall={wall:.1f} ms", flush=True)
                if n_saved == 0:
                    break
                identity = multi and n_saved == n_use_global and i_mult == 1
                n_prev_global = n_use_global                     # (every global index of the pcut just run is below this)
                n_use_global = n_saved * i_mult
                if identity:
                    # everybody was saved and nobody is replicated (the first pcuts of a species): the saved particles' positions
                    # in the global order ARE their indices, so every rank's children keep the global indices their parents had
                    # -- no index column to exchange, no particle to move; the shard description (first / stride / gidx) stands
                    backend.new_pcut(1)
                    stats[-1].split = "identity"
                elif not multi:
                    backend.new_pcut(i_mult)                     # one process: the shard stays 0, 1, 2, ...
                    n_local = n_use_global
                elif local_ok:
                    # every rank splits its own saved particles; the index column alone goes round
                    g_loc = backend.saved_gidx()                                  # ascending, counts[rank] entries
                    cap = max(max(counts), 1)
                    # (padded with the largest integer: every row of the gathered table stays sorted; for each of my saved particles a
                    # searchsorted per peer row counts that rank's saved particles below it.  One row at a time, accumulated in place:
                    # the working set is O(n) -- round 3 searched all rows at once through an expanded [W, n] key matrix and an int64
                    # [W, n] result, 0.5 + 0.8 GB per rank and pcut at config[3]'s 1.25e7 particles per GPU)
                    # (the column travels as int32 while every index fits: half the bytes on the wire and in the search -- 4 B per saved
                    # particle per peer, 32 MB per rank and pcut at 10^6 particles per GPU on 8 GPUs)
                    idt = torch.int32 if n_prev_global < 2 ** 31 - 1 else torch.int64
                    g_key = g_loc.to(idt)
                    pad = torch.full((cap,), torch.iinfo(idt).max, dtype=idt, device=g_loc.device)
                    pad[:g_key.numel()] = g_key
                    g_all = comm.all_gather_rows(pad)                             # [W, cap]
                    pos = torch.zeros(g_key.numel(), dtype=torch.int64, device=g_loc.device)
                    for w in range(g_all.shape[0]):
                        pos += torch.searchsorted(g_all[w], g_key)
                    gidx = (pos[:, None] * i_mult + torch.arange(i_mult, dtype=torch.int64, device=g_loc.device)[None, :]).reshape(-1).contiguous()
                    backend.new_pcut(i_mult)
                    n_local = counts[comm.rank] * i_mult
                else:
                    # all ranks see all parents (sorted by global index); rank r builds elements r, r+W, ... of the split
                    g, f64, meta = backend.export_saved(max(max(counts), 1))
                    g = comm.all_gather_cols(g, counts)
                    f64 = comm.all_gather_cols(f64, counts)
                    meta = comm.all_gather_cols(meta, counts)
                    order = torch.argsort(g, stable=True)
                    f64 = f64.index_select(1, order).contiguous()
                    meta = meta.index_select(0, order).contiguous()
                    first, stride, gidx = comm.rank, comm.world, None
                    n_local = (n_use_global - comm.rank + comm.world - 1) // comm.world if n_use_global > comm.rank else 0
                    backend.import_split(f64, meta, n_saved, i_mult, first, stride, n_local)

            # species end: merge the partial tallies of all ranks (C1)
            if multi and dev_t is not None:
                backend.sync()          # the bound tensors are complete after mcs_sync (it folds the tally replicas in)
                tf, ti = dev_t
                local_steps.append((i_iter, i_ion, int(ti[i_h].item() + ti[i_r].item()) - steps_seen))
                if not is_root:   # every rank carried a full copy of the received-energy pool
                    tview(tf, "energy_recv_pool").zero_()
                comm.all_reduce_sum_(tf)
                comm.all_reduce_sum_(ti)
                G_pool = tview(tf, "energy_transfer_pool").clone()
                last_read = final_full_read and i_iter == first_iter + n_itrs - 1 and i_ion == len(cfg.species)
                if species_tallies == "light" and not last_read:
                    o_small = L.offsets["esc_psd_up"]
                    G_f = np.zeros(L.total)
                    G_f[o_small:] = tf[o_small:].cpu().numpy()
                    G_i = ti.cpu().numpy()
                else:
                    G_f, G_i = tf.cpu().numpy(), ti.cpu().numpy()
                steps_seen = int(G_i[i_h] + G_i[i_r]) if is_root else 0     # rank 0 carries the merged totals on
                if not is_root:
                    tf.zero_(); ti.zero_()
            else:
                last_read = final_full_read and i_iter == first_iter + n_itrs - 1 and i_ion == len(cfg.species)
                light = species_tallies == "light" and not last_read and not multi and hasattr(backend, "read_tallies_light")
                f, i = backend.read_tallies_light() if light else backend.read_tallies()
                local_steps.append((i_iter, i_ion, int(i[i_h] + i[i_r]) - steps_seen))
                steps_seen = int(i[i_h] + i[i_r])
                if multi:
                    if not is_root:
                        L.view(f, "energy_recv_pool")[...] = 0.0
                    tf, ti = torch.from_numpy(f), torch.from_numpy(i)
                    comm.all_reduce_sum_(tf); comm.all_reduce_sum_(ti)
                    G_f, G_i = f.copy(), i.copy()
                    steps_seen = int(G_i[i_h] + G_i[i_r]) if is_root else 0
                    G_pool = L.view(G_f, "energy_transfer_pool").copy()
                    if is_root:
                        backend.write_tallies(G_f, G_i)
                    else:
                        backend.write_tallies(np.zeros_like(G_f), np.zeros_like(G_i))
                else:
                    G_f, G_i = f, i
            per_species.append((i_iter, i_ion, G_f, G_i))      # fresh host arrays: no copy needed
            if on_species_end is not None:
                on_species_end(i_iter, i_ion, G_f, G_i)

        if finalize:
            # ion_finalize of the last species (quirk Q2: only its fluxes and pressures reach iter_finalize) and
            # iter_finalize, on rank 0, whose device buffers hold the merged tallies
            changed = False
            if is_root:
                ion_fin = consumers.ion_finalize(prob, backend, len(cfg.species))
                fin = itf.iter_finalize(prob, it_state, sm, i_iter, G_f, L, ion_fin.P_psd_par, ion_fin.P_psd_perp,
                                        ion_fin.energy_density_psd)
                iter_finals.append((i_iter, fin, ion_fin))
                changed = fin.profile_changed
            if multi and sm.smooth_shocks:
                tabs = torch.from_numpy(np.stack([prob.ux, prob.gam_sf, prob.utot, prob.beta_ef, prob.gam_ef, prob.btot]))
                dev = comm.device if comm.dist.get_backend() == "nccl" else None
                tabs = tabs.to(dev) if dev is not None else tabs
                comm.dist.broadcast(tabs, src=0)
                tabs = tabs.cpu().numpy()
                for k, name in enumerate(("ux", "gam_sf", "utot", "beta_ef", "gam_ef", "btot")):
                    getattr(prob, name)[:] = tabs[k]
                changed = True
            if changed:
                itf.populate_eps_target(prob)        # src/main_loops.jl:76-81, top of the next iteration
                backend.set_grid(prob)
                backend.set_cuts(prob)
        if tcut_print and P.do_tcuts:
            # (after iter_finalize, as at src/main_loops.jl:363-389; G_f is the merged buffer of the last species, which holds the
            # coupled arrays of every species -- they are per-ion slices of one array).  The rewrite is applied to a COPY of the
            # buffer -- the entries already handed out in per_species / on_species_end stay the raw sums -- and on every rank, so
            # that all ranks return the same RunResult; only the root writes the device buffer.
            from . import iter_finalize as _itf
            G_f = G_f.copy()
            wc, sc = L.view(G_f, "weight_coupled"), L.view(G_f, "spectra_coupled")
            _itf.tcut_print(wc, sc, len(prob.tcuts), P.num_psd_mom_bins)
            if is_root:
                backend.write_tally("weight_coupled", wc)
                backend.write_tally("spectra_coupled", sc)
        if on_iteration_end is not None:
            on_iteration_end(i_iter)

    ng = P.n_grid
    from .capi import IC
    return RunResult(G_f, G_i, per_species, stats,
                     int(G_i[ng + IC["STEPS_HELIX"]]), int(G_i[ng + IC["STEPS_RETRO"]]), iter_finals,
                     it_state if finalize else None, local_steps, empty_launches)


# The never-reset tallies of the reference (esc_flux, esc_*_eff, spectra_coupled, spectra_sf / _pf accumulate over
# the iterations of a run; px_esc_feb / energy_esc_feb are indexed by iteration): sums over iterations, hence over contexts.
ACCUMULATED_OVER_ITERATIONS = ("esc_flux", "px_esc_feb", "energy_esc_feb", "esc_energy_eff", "esc_num_eff", "spectra_coupled",
                               "spectra_sf", "spectra_pf")


def run_overlapped(prob: Problem, backends, n_itrs: Optional[int] = None, max_pcuts: Optional[int] = None,
                   on_iteration_end: Optional[Callable] = None, first_iter: int = 1,
                   blocks_per_launch: Optional[int] = None) -> RunResult:
    """The iterations of a run with a FIXED shock profile (smooth-shocks = false -- the stock mc_in.toml, BASELINE
    config[1]) are independent Monte-Carlo realisations: nothing an iteration computes enters the next one's transport
    (src/main_loops.jl:52-121: every tally the transport reads is reset at the top; the RNG keys carry i_iter).  Their
    launches can therefore share the GPU: len(backends) iterations are in flight at a time, each on its own context and
    HIP stream, driven by its own host thread; while one iteration's launch waits for its longest histories (the per-pcut
    tail, 40 % of an iteration at 10^6 particles) the blocks of the other's become resident on the CUs it has freed.
    Per-iteration results are those of run(): same keys, same populations; iter_finalize runs on the host in iteration
    order.  The tallies the reference never resets are sums over iterations and are merged over the contexts at the end.
    blocks_per_launch: workgroups of a K1 launch while iterations overlap; default 2 x #CU / len(backends), i.e. with two
    contexts ONE workgroup per CU each: the two launches are then resident side by side from the start (a CU holds two
    workgroups), each SIMD carries one wave of either, and a wave whose neighbour is in its launch's tail issues at the
    lone-wave rate -- measured 254 ms per iteration against 275 with full-chip launches, which let the other launch in only
    as whole workgroups retire (tools/gpu_concurrent.py).
    Single process only (no communicator): collectives issued from two threads would need an order."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from . import consumers, iter_finalize as itf
    cfg, P = prob.cfg, prob.params
    n_itrs = n_itrs if n_itrs is not None else cfg.num_iterations
    K = len(backends)
    assert K >= 1
    L = backends[0].layout
    sm = itf.SmoothingConfig(smooth_shocks=False)
    st = itf.IterState.create(prob, sm, P.n_itrs)
    locks = [threading.Lock() for _ in backends]
    if blocks_per_launch is None and K > 1 and hasattr(backends[0], "num_cus"):
        blocks_per_launch = max(2 * backends[0].num_cus() // K, 1)

    state = {"busy": 0}
    state_lock = threading.Lock()

    def one(i_iter):
        k = (i_iter - first_iter) % K
        with locks[k]:                       # a context carries one iteration at a time
            be = backends[k]
            with state_lock:
                state["busy"] += 1
            geo = blocks_per_launch and K > 1 and hasattr(be, "set_launch")

            def geometry(*_):
                # decided before EVERY launch: an iteration that is alone on the chip (the last one of an odd count, once its
                # neighbour has finished) gets the automatic full-chip geometry back
                if geo:
                    alone = state["busy"] <= 1
                    be.set_launch(0 if alone else int(blocks_per_launch), 0 if alone else 256)
            try:
                # (long_draws=0: the pcuts are not pipelined here -- the other iterations' launches are what fills this one's tails, and
                # the per-pcut hook needs the per-pcut loop; MCS_LONG_DRAWS does not reach this call)
                res = run(prob, be, None, n_itrs=1, max_pcuts=max_pcuts, first_iter=i_iter, species_tallies="light", final_full_read=False,
                          before_pcut=geometry, long_draws=0)
                ion_fin = consumers.ion_finalize(prob, be, len(cfg.species))     # K4, before the context is reused
            finally:
                with state_lock:
                    state["busy"] -= 1
                if geo:
                    be.set_launch(0, 0)      # back to the automatic geometry, also when the iteration raised
        return k, res, ion_fin

    stats, per_species, iter_finals, local_steps = [], [], [], []
    last = {}                                 # context -> its latest result (running totals of the never-reset tallies)
    from .capi import IC
    ng = P.n_grid
    total = []                                # the step counters are running totals of a context: where each one starts
    for be in backends:
        i0 = be.read_counters() if hasattr(be, "read_counters") else be.read_tallies()[1]
        total.append(int(i0[ng + IC["STEPS_HELIX"]] + i0[ng + IC["STEPS_RETRO"]]))
    with ThreadPoolExecutor(max_workers=K) as pool:
        futs = [pool.submit(one, i) for i in range(first_iter, first_iter + n_itrs)]
        for i_iter, fu in zip(range(first_iter, first_iter + n_itrs), futs):      # consumed in iteration order
            k, res, ion_fin = fu.result()
            fin = itf.iter_finalize(prob, st, sm, i_iter, res.tallies_f64, L, ion_fin.P_psd_par, ion_fin.P_psd_perp,
                                    ion_fin.energy_density_psd)
            local_steps.append((i_iter, len(cfg.species), res.steps_helix + res.steps_retro - total[k]))
            total[k] = res.steps_helix + res.steps_retro
            last[k] = res
            stats.extend(res.stats); per_species.extend(res.per_species); iter_finals.append((i_iter, fin, ion_fin))
            if on_iteration_end is not None:
                on_iteration_end(i_iter)
    # the state after the last iteration: its context's buffer, with the never-reset tallies summed over the contexts
    k_last = (n_itrs - 1) % K
    f, i64 = backends[k_last].read_tallies()              # the only time the three histograms cross to the host
    for k in last:
        if k == k_last:
            continue
        fk, ik = (backends[k].read_tallies_light() if hasattr(backends[k], "read_tallies_light") else backends[k].read_tallies())
        for name in ACCUMULATED_OVER_ITERATIONS:
            L.view(f, name)[...] += L.view(fk, name)
        i64[ng:] += ik[ng:]                   # the event counters are running totals too (num_crossings is per species)
    return RunResult(f, i64, per_species, stats, int(i64[ng + IC["STEPS_HELIX"]]), int(i64[ng + IC["STEPS_RETRO"]]),
                     iter_finals, st, local_steps)

