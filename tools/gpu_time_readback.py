import time, sys, numpy as np
sys.path.insert(0, "tests")
from conftest import mcs, make_problem, hip_backend, start_species
prob = make_problem(100000)
hb = hip_backend(prob)
start_species(hb, prob)
hb.run_pcut(1, 0)
for _ in range(3):
    t = time.perf_counter(); f, i = hb.read_tallies(); dt = time.perf_counter() - t
    print(f"read_tallies: {f.nbytes/1e6:.1f} MB in {dt*1e3:.2f} ms = {f.nbytes/dt/1e9:.1f} GB/s")
from mcs_amd import consumers
t = time.perf_counter(); fin = consumers.ion_finalize(prob, hb, 1); print("ion_finalize (K4 + readback of its results):", (time.perf_counter()-t)*1e3, "ms")
t = time.perf_counter(); fin = consumers.ion_finalize(prob, hb, 1); print("ion_finalize again:", (time.perf_counter()-t)*1e3, "ms")
