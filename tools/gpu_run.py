"""Run the HIP path only (no oracle) for N particles over the first NPC pcuts; print per-pcut kernel time and steps."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _mcs_loader; m = _mcs_loader.load()
from mcs_amd import hip_backend
N = int(sys.argv[1]); NPC = int(sys.argv[2])
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 0
MOD = len(sys.argv) > 4 and sys.argv[4] == "mod"      # precursor profile: u_x differs in every upstream zone
FP32 = "fp32" in sys.argv[4:]                         # the fp32-state variant (MCS_F32_LOOP=1: its plain-loop form)
cfg = m.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, state_fp32=FP32)
prob = m.inputs.build_problem(cfg)
if MOD:
    C = m.constants.C
    x = prob.x_grid_cm; up = x < 0
    ux = prob.ux.copy(); ux[up] = prob.ux[1] * (1 - 0.3 * np.exp(x[up] / (50.0 * prob.rg0)))
    prob.ux = ux; prob.utot = np.hypot(prob.ux, prob.uz); prob.gam_sf = 1 / np.sqrt(1 - (prob.utot / C) ** 2)
hb = hip_backend.HipBackend(0, debug_finals=True); hb.create(prob)
if blocks: hb.set_launch(blocks, 256)
hb.begin_iteration(1)
inj = m.inputs.init_pop_host(prob, 1)
hb.begin_species(1, 1, 1.0, 1.0, prob.pmax, 1.0, 1.0)
hb.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
hb.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
ng = prob.n_grid; IC = m.capi.IC
prev = 0; tot_ms = 0; tot_steps = 0; tot_launches = 0
import ctypes as ct
hb.sync(); t_wall0 = time.perf_counter()
for ip in range(1, NPC + 1):
    n = hb.pop_size()
    ns = hb.run_pcut(ip, 0)
    i64 = np.zeros(hb.layout.n_i64, dtype=np.int64)
    hb.lib.mcs_read_tallies(hb.h, None, i64.ctypes.data_as(ct.POINTER(ct.c_int64)))
    st = int(i64[ng + IC["STEPS_HELIX"]] + i64[ng + IC["STEPS_RETRO"]]); d = st - prev; prev = st
    retro = int(i64[ng + IC["STEPS_RETRO"]])
    ms = hb.last_kernel_ms(); tot_ms += ms; tot_steps += d
    tot_launches += hb.last_launches()
    print(f"pcut {ip:2d} n={n} saved={ns} steps={d} kernel={ms:.2f} ms launches={hb.last_launches()} rate={d/(ms*1e-3+1e-12):.3e} steps/s retro_cum={retro}", flush=True)
    if ns == 0: break
    hb.new_pcut(max(N // ns, 1))
hb.sync(); wall = (time.perf_counter() - t_wall0) * 1e3
print(f"TOTAL steps={tot_steps} kernel_ms={tot_ms:.1f} wall_ms={wall:.1f} (incl. one tally read-back per pcut) launches={tot_launches} rate={tot_steps/(tot_ms*1e-3):.3e} steps/s "
      f"MCS_TAIL_BUDGET={os.environ.get('MCS_TAIL_BUDGET', '0')}")
