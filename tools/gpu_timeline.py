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
hb = hip_backend.HipBackend(0); hb.create(prob)
if BLOCKS: hb.set_launch(BLOCKS, 256)
hb.begin_iteration(1)
inj = m.inputs.init_pop_host(prob, 1)
hb.begin_species(1, 1, 1.0, 1.0, prob.pmax, 1.0, 1.0)
hb.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
hb.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
buf = np.zeros((8192, 4), dtype=np.uint64)
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
    if ns == 0: break
    hb.new_pcut(max(N // ns, 1))
