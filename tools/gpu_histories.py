"""Per-particle step counts and saved flags of every pcut of one iteration (BASELINE config[1] shape), for
tools/prefix_pipeline_sim.py.  usage: python tools/gpu_histories.py N out.npz [oracle]
(`oracle`: the CPU oracle instead of the GPU -- small N, to try the simulation without a GPU)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _mcs_loader; m = _mcs_loader.load()

N = int(sys.argv[1]); OUT = sys.argv[2]
USE_ORACLE = len(sys.argv) > 3 and sys.argv[3] == "oracle"
prob = m.inputs.build_problem(m.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N))
if USE_ORACLE:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    be = orc.OracleBackend(m.capi, "det", nthreads=os.cpu_count() or 1)
else:
    from mcs_amd import hip_backend
    be = hip_backend.HipBackend(0, debug_finals=True)
be.create(prob)
be.begin_iteration(1)
inj = m.inputs.init_pop_host(prob, 1)
be.begin_species(1, 1, 1.0, 1.0, prob.pmax, 1.0, 1.0)
be.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
be.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
out = {"n_target": N}
kms, imult = [], []
k = 0
for ip in range(1, len(prob.pcuts) + 1):
    ns = be.run_pcut(ip, 0)
    f = be.finals()
    steps = f["helix"].astype(np.int64) + f["retro"].astype(np.int64)
    out[f"steps_{k}"] = np.minimum(steps, 65535).astype(np.uint16)
    out[f"saved_{k}"] = np.packbits(f["reason"] == 0)
    kms.append(be.last_kernel_ms() if hasattr(be, "last_kernel_ms") else 0.0)
    im = max(N // ns, 1) if ns else 0
    imult.append(im)
    print(f"pcut {ip:2d}: n {len(steps)} saved {ns} i_mult {im} steps {int(steps.sum())} max {int(steps.max())} kernel {kms[-1]:.2f} ms", flush=True)
    k += 1
    if ns == 0:
        break
    be.new_pcut(im)
out["n_pcuts"] = k
out["kernel_ms"] = np.array(kms, dtype=np.float64)
out["i_mult"] = np.array(imult, dtype=np.int64)
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT) >> 20, "MiB")
