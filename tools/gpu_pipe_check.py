"""Parity of the pipelined pcut loop (mcs_run_pcuts_pipelined) with the oracle run in the same population order: per-pcut statistics and
integer tallies bit for bit, fp64 tallies to rounding.  usage: gpu_pipe_check.py [N] [long_draws] [long_imult_max]"""
import sys, time, os
sys.path.insert(0, 'tests')
import numpy as np
from conftest import mcs, make_problem, oracle_backend, hip_backend
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 400
IM = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prob = make_problem(N)
ob = oracle_backend(prob, nthreads=16)
ro = mcs.driver.run(prob, ob, n_itrs=1, long_draws=B, long_imult_max=IM)
hb = hip_backend(prob)
rg = mcs.driver.run(prob, hb, n_itrs=1, long_draws=B, long_imult_max=IM, verbose=True)
so = [(s.n_pts_use, s.n_saved, s.i_mult) for s in ro.stats]
sg = [(s.n_pts_use, s.n_saved, s.i_mult) for s in rg.stats]
print("stats equal:", so == sg, len(so), len(sg))
if so != sg:
    for a, b in zip(so, sg):
        print(a, b, "" if a == b else "  <<<")
print("int tallies equal:", np.array_equal(ro.tallies_i64, rg.tallies_i64))
d = np.abs(ro.tallies_f64 - rg.tallies_f64).max() / np.abs(ro.tallies_f64).max()
print("f64 rel diff:", d, "steps", ro.steps_helix, rg.steps_helix)
ob.destroy(); hb.destroy()
