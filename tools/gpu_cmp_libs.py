"""Debug aid: run the same problem through two builds of the library (two processes' worth of state in one: the
library path is read at load time, so this script takes the library from MCS_HIP_LIB and dumps the per-particle
finals of every pcut to an .npz; run it twice and diff).  usage: MCS_HIP_LIB=lib.so python tools/gpu_cmp_libs.py out.npz [mod]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import mcs, make_problem, hip_backend, start_species
from test_gpu_parity import modified_profile
N = 900
prob = make_problem(N)
if len(sys.argv) > 2 and sys.argv[2] == "mod": modified_profile(prob)
hb = hip_backend(prob); start_species(hb, prob)
out = {}
for ip in range(1, 10):
    ns = hb.run_pcut(ip, 0)
    f = hb.finals()
    for k, v in f.items(): out[f"p{ip}_{k}"] = v
    if ns == 0: break
    hb.new_pcut(max(N // ns, 1))
np.savez(sys.argv[1], **out)
