"""Debug: the first pcuts of N protons through the wave-specialised kernel; prints the error word of a failed launch."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _mcs_loader; m = _mcs_loader.load()
from mcs_amd import hip_backend
for N in [int(x) for x in sys.argv[1:]]:
    cfg = m.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N)
    prob = m.inputs.build_problem(cfg)
    hb = hip_backend.HipBackend(0, debug_finals=True); hb.create(prob)
    hb.begin_iteration(1)
    inj = m.inputs.init_pop_host(prob, 1)
    hb.begin_species(1, 1, 1.0, 1.0, prob.pmax, 1.0, 1.0)
    hb.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
    hb.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
    for ip in range(1, 8):
        try:
            ns = hb.run_pcut(ip, 0)
            print(N, ip, "ok", ns, hb.last_kernel_ms(), flush=True)
        except RuntimeError as e:
            print(N, ip, "FAIL", e, flush=True); break
        if ns == 0: break
        hb.new_pcut(max(N // ns, 1))
    hb.destroy()
