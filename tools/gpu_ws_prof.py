"""Event profile of the wave-specialised K1 (csrc/libmcs_hip_prof.so: `make prof`; with PROFDEFS="-DMCS_PROF -DMCS_WS_TIMERS" also
s_memtime ticks per phase -- intrusive, ratios).  usage: MCS_HIP_LIB=libmcs_hip_prof.so python tools/gpu_ws_prof.py N NPC [first]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ctypes as ct
import _mcs_loader; m = _mcs_loader.load()
from mcs_amd import hip_backend
N = int(sys.argv[1]); NPC = int(sys.argv[2]); FIRST = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = m.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N)
prob = m.inputs.build_problem(cfg)
hb = hip_backend.HipBackend(0, debug_finals=True); hb.create(prob)
hb.begin_iteration(1)
inj = m.inputs.init_pop_host(prob, 1)
hb.begin_species(1, 1, 1.0, 1.0, prob.pmax, 1.0, 1.0)
hb.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
hb.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
ng = prob.n_grid; IC = m.capi.IC
prof = np.zeros(64, dtype=np.uint64)
hb.lib.mcs_prof_read.argtypes = [ct.c_void_p, ct.c_int]
prev = 0
for ip in range(1, NPC + 1):
    n = hb.pop_size()
    ns = hb.run_pcut(ip, 0)
    i64 = np.zeros(hb.layout.n_i64, dtype=np.int64)
    hb.lib.mcs_read_tallies(hb.h, None, i64.ctypes.data_as(ct.POINTER(ct.c_int64)))
    st = int(i64[ng + IC["STEPS_HELIX"]] + i64[ng + IC["STEPS_RETRO"]]); d = st - prev; prev = st
    ms = hb.last_kernel_ms()
    assert hb.lib.mcs_prof_read(prof.ctypes.data, 1) == 0
    P = prof.astype(np.float64)
    if ip >= FIRST and P[0] > 0:
        T = P[0]
        print(f"pcut {ip:2d} n={n} saved={ns} steps={d} kernel={ms:.2f} ms rate={d/(ms*1e-3):.3e}/s kernel id {hb.last_kernel()}")
        print(f"   wave trips {T:.3e}; lanes holding a particle at the header {P[8]/T:.1f}, of them pending {P[13]/T:.1f}; passes/trip {P[7]/T:.1f}, lanes RUNNING per pass {P[9]/max(P[7],1):.1f}")
        print(f"   per trip: taken from R {P[40]/T:.2f}, fresh {P[41]/T:.3f}, ended {P[10]/T:.3f}, sent on to H {P[47]/T:.3f}")
        print(f"   services/trip {P[12]/T:.3f} ({P[46]/max(P[12],1):.1f} lanes each); E batches {P[42]:.3e} ({P[44]/max(P[42],1):.1f} each), H batches {P[43]:.3e} ({P[45]/max(P[43],1):.1f} each)")
        if P[54] > 0:
            tot = P[50] + P[51] + P[52] + P[53] + P[54]
            print(f"   (timers) ticks per trip: housekeeping {P[50]/T:.0f}, passes {P[54]/T:.0f}, E service {P[51]/T:.0f} ({P[51]/max(P[55],1):.0f} per batch), H service {P[52]/T:.0f} ({P[52]/max(P[56],1):.0f} per batch), in-place / none {P[53]/T:.0f} ({P[53]/max(P[57],1):.0f} per header); total {tot/T:.0f}")
    if ns == 0: break
    hb.new_pcut(max(N // ns, 1))
