"""Where the wall time of an iteration goes besides the transport kernel: wall time per backend call, against the HIP-event
kernel time.  usage: python tools/gpu_host_gap.py [N] [iterations]"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _mcs_loader; m = _mcs_loader.load()
from mcs_amd import hip_backend
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NIT = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = m.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=NIT + 1)
prob = m.inputs.build_problem(cfg)
hb = hip_backend.HipBackend(0); hb.create(prob)
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t; cnt[name] += 1; return r
    setattr(obj, name, g)
for name in ("begin_iteration", "begin_species", "set_fluxes", "init_pop", "run_pcut", "run_pcuts_fused", "new_pcut", "read_tallies_light", "read_tallies", "read_counters",
             "dndp_cr", "thermo_calcs", "set_grid", "set_cuts"):
    wrap(hb, name)
itf = m.iter_finalize
orig_if = itf.iter_finalize
def timed_if(*a, **k):
    t = time.perf_counter(); r = orig_if(*a, **k); acc["iter_finalize(host numpy)"] += time.perf_counter() - t; cnt["iter_finalize(host numpy)"] += 1; return r
itf.iter_finalize = timed_if
sm = itf.SmoothingConfig(smooth_shocks=False)
m.driver.run(prob, hb, None, n_itrs=1, smoothing=sm, species_tallies="light")        # warm
acc.clear(); cnt.clear()
hb.sync(); t0 = time.perf_counter()
res = m.driver.run(prob, hb, None, n_itrs=NIT, smoothing=sm, species_tallies="light", first_iter=2, final_full_read=False)
hb.sync(); wall = (time.perf_counter() - t0) / NIT * 1e3
kern = sum(s.kernel_ms for s in res.stats) / NIT
print(f"N = {N}: wall {wall:.2f} ms per iteration, K1 kernels (HIP events) {kern:.2f} ms, gap {wall - kern:.2f} ms; {len(res.stats)//NIT} pcuts per iteration")
tot = 0.0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"   {k:28s} {v/NIT*1e3:8.3f} ms per iteration in {cnt[k]//NIT:3d} calls"); tot += v
print(f"   {'sum of the above':28s} {tot/NIT*1e3:8.3f} ms;  (run_pcut + run_pcuts_fused) wall - K1 kernel time = {(acc['run_pcut'] + acc['run_pcuts_fused'])/NIT*1e3 - kern:.3f} ms"
      f"   [MCS_FUSED_PCUTS={os.environ.get('MCS_FUSED_PCUTS', '1')}]")
