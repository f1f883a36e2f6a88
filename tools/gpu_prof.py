"""Event profile of K1 (needs csrc/libmcs_hip_prof.so = the library built with -DMCS_PROF added to
TUNE; run with MCS_HIP_LIB=libmcs_hip_prof.so): per pcut, how often the rare region is entered and why.
(Cycle timers were used during development and removed: an s_memtime read costs ~450 cycles and
serialises the LDS queue, which distorts exactly what it is meant to measure.)
usage: MCS_HIP_LIB=libmcs_hip_prof.so python tools/gpu_prof.py N NPC [first_pcut_to_print]"""
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
    if ip >= FIRST:
        passes = P[0]
        print(f"pcut {ip:2d} n={n} saved={ns} steps={d} kernel={ms:.2f} ms rate={d/(ms*1e-3):.3e}/s")
        print(f"   wave trips {passes:.3e}, live lanes/trip {P[8]/passes:.1f}; common passes {P[7]:.3e} ({P[7]/passes:.1f} per trip), live lanes RUNNING per common pass {P[9]/max(P[7],1):.1f}")
        print(f"   rare region entered in {100*P[12]/passes:.1f} % of passes ({P[13]/max(P[12],1):.1f} lanes each); full path for {P[16]/max(P[12],1):.2f} lanes per entry")
        e = max(P[12], 1)
        print(f"   lanes per entry with: xn switch due {P[22]/e:.2f}, x>=x_up {P[23]/e:.2f}, time event {P[24]/e:.2f}, refresh flags {P[25]/e:.2f}, crossing not plain {P[26]/e:.2f}, near FEB {P[27]/e:.2f}, to be saved {P[28]/e:.2f}, new particle {P[29]/e:.2f}")
        if P[31] > 0: print(f"   (MCS_PROF_TAIL) rare region after exhaustion: {P[31]:.3e} timed entries, {P[30]/P[31]:.0f} s_memtime ticks each (100 MHz x ? -- see tools/ubench)")
        if P[21] > 0: print(f"   (MCS_PROF_TAIL) per full-path group ({P[21]:.3e}): before {P[20]/P[21]:.0f}, slow_post+block1 {P[17]/P[21]:.0f}, slow_pre {P[18]/P[21]:.0f}, particle end {P[19]/max(P[10],1):.0f} per ended group ({P[10]:.3e})")
        if P[32] > 0: print(f"   (MCS_PROF_TAIL) per entry ({P[12]:.3e}): classification {P[32]/P[12]:.0f}, plain_crossing {P[33]/P[12]:.0f}, light handlers {P[34]/P[12]:.0f}")
        if P[48:54].sum() > 0:
            names = ["time cut", "reflect/shock", "zone search + record", "downstream test", "prob_return (+retro walk)", "flags + refresh"]
            print("   (MCS_PROF_TAIL) slow_post sections, ticks per execution (executions): " + "; ".join(f"{nm} {P[40+i]/max(P[48+i],1):.0f} ({P[48+i]:.2e})" for i, nm in enumerate(names)) + f"; retro steps {P[39]:.3e}")
        if P[36] > 0: print(f"   parked particles resumed: {P[37]:.3e} in {P[36]:.3e} refills ({P[37]/P[36]:.1f} each)")
        if P[56:60].sum() > 0: print(f"   (MCS_PROF_TAIL) upward events after exhaustion: grid end crossed {P[56]:.3e}, PRP escape {P[57]:.3e}, PRP return (retro walk) {P[58]:.3e}, beyond x_dt {P[59]:.3e}")
        print(f"   drains/pass {P[3]/passes:.4f}  refills/pass {P[5]/passes:.4f} ({P[6]/max(P[5],1):.1f} lanes each)  particles ended/pass {P[10]/passes:.4f}", flush=True)
    if ns == 0: break
    hb.new_pcut(max(N // ns, 1))
