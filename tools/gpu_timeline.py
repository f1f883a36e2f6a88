"""Per-wave timeline of K1 (libmcs_hip_prof.so): when each wave found the work counter exhausted and when
it ended, relative to the first wave's start.  usage: MCS_HIP_LIB=libmcs_hip_prof.so python tools/gpu_timeline.py N pcut"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ctypes as ct
import _mcs_loader; m = _mcs_loader.load()
from mcs_amd import hip_backend
N = int(sys.argv[1]); PC = [int(x) for x in sys.argv[2].split(",")]
BLOCKS = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prob = m.inputs.build_problem(m.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N))
hb = hip_backend.HipBackend(0, debug_finals=True); hb.create(prob)
if BLOCKS: hb.set_launch(BLOCKS, 256)
hb.begin_iteration(1)
inj = m.inputs.init_pop_host(prob, 1)
hb.begin_species(1, 1, 1.0, 1.0, prob.pmax, 1.0, 1.0)
hb.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
hb.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
buf = np.zeros((8192, 8), dtype=np.uint64)
for ip in range(1, max(PC) + 1):
    ns = hb.run_pcut(ip, 0)
    if ip in PC:
        hb.lib.mcs_prof_waves(buf.ctypes.data_as(ct.c_void_p))
        f = hb.finals(); h = f["helix"].astype(np.int64)
        nw = (BLOCKS or 512) * 4
        w = buf[:nw].astype(np.float64)
        t0 = w[:, 0].min()
        ex = (w[:, 1] - t0) / 100.0; en = (w[:, 2] - t0) / 100.0; st = (w[:, 0] - t0) / 100.0      # microseconds (100 MHz)
        print(f"pcut {ip}: kernel {hb.last_kernel_ms():.2f} ms; histories: mean {h.mean():.0f}, p99 {np.percentile(h,99):.0f}, max {h.max()}, >5000: {(h>5000).sum()}, >9000: {(h>9000).sum()}")
        print(f"   wave start spread {st.max():.0f} us; exhausted at: min {ex.min():.0f} median {np.median(ex):.0f} max {ex.max():.0f} us; lanes live then: mean {w[:,3].mean():.1f}")
        q = np.percentile(en, [10, 50, 90, 99, 100])
        print(f"   wave end at: p10 {q[0]:.0f} p50 {q[1]:.0f} p90 {q[2]:.0f} p99 {q[3]:.0f} max {q[4]:.0f} us")
        # the phase after exhaustion, per wave: time per pass against the mean number of live lanes
        dur = en - ex; ps = np.maximum(w[:, 4], 1); live = w[:, 6] / ps
        print(f"   after exhaustion: passes/wave mean {w[:,4].mean():.0f} max {w[:,4].max():.0f}; us/pass overall {dur.sum()/w[:,4].sum():.3f}; ")
        for lo, hi in ((0, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 65)):
            sel = (live >= lo) & (live < hi) & (w[:, 4] > 50)
            if sel.any():
                print(f"      waves with mean live lanes in [{lo},{hi}): {sel.sum():5d}  us/pass {dur[sel].sum()/w[sel,4].sum():.3f}  passes {w[sel,4].mean():.0f}")
        if ip == PC[0]:     # placement: which waves share a SIMD (HW_REG_HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]; XCC id)
            hw = buf[:nw, 5].astype(np.int64); xcc = buf[:nw, 7].astype(np.int64) & 0xf
            slot = hw & 0xf; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
            cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
            import collections
            byc = collections.defaultdict(list)
            for g in range(nw): byc[int(cuid[g])].append((int(simd[g]), int(slot[g]), g // 4, g % 4, int(en[g]), int(w[g, 4])))
            print(f"   distinct CUs used: {len(byc)}")
            for c in list(sorted(byc))[:6]:
                print(f"      cu {c}: (simd, slot, block, wave, end us, passes after exhaustion) = {sorted(byc[c])}")
            same = sum(1 for g in range(0, nw, 4) if len(set(simd[g:g+4])) == 4 and all(simd[g + i] == (simd[g] + i) % 4 for i in range(4)))
            print(f"   blocks whose waves 0..3 sit on SIMDs s, s+1, s+2, s+3: {same} of {nw//4}; wave w on SIMD w: {sum(1 for g in range(nw) if simd[g] == g % 4)} of {nw}")
            sl = collections.Counter((int(slot[g]) for g in range(nw))); print(f"   wave-slot histogram: {dict(sl)}")
            blkslot = sum(1 for g in range(0, nw, 4) if len(set(slot[g:g+4])) == 1); print(f"   blocks whose 4 waves have the same slot id: {blkslot} of {nw//4}")
            partner = collections.Counter()
            for c, lst in byc.items():
                blocks = sorted(set(t[2] for t in lst))
                if len(blocks) == 2: partner[blocks[1] - blocks[0]] += 1
                else: partner[('n', len(blocks))] += 1
            print(f"   block-id distance of the two blocks sharing a CU: {dict(partner)}")
        last = np.argsort(en)[-8:]
        for i in last:
            print(f"      late wave {i}: end {en[i]:.0f} us, {int(w[i,4])} passes after exhaustion in {dur[i]:.0f} us = {dur[i]/ps[i]:.3f} us/pass, mean live {live[i]:.1f}")
    if ns == 0: break
    hb.new_pcut(max(N // ns, 1))
