"""Experiment: K independent iterations on K host threads / K HIP streams vs sequentially."""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _mcs_loader; m = _mcs_loader.load()
from mcs_amd import hip_backend
N = int(sys.argv[1]); K = int(sys.argv[2]); NPC = int(sys.argv[3]) if len(sys.argv) > 3 else 45
BLOCKS = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # workgroups per launch in the concurrent leg (0: the default, 2 per CU)
cfg = m.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=8)
prob = m.inputs.build_problem(cfg)
bes = []
for k in range(K):
    b = hip_backend.HipBackend(0, debug_finals=True); b.create(prob); bes.append(b)
res = [None] * K
def work(k, it):
    # one iteration (i_iter = it) on backend k
    be = bes[k]
    be.begin_iteration(it)
    inj = m.inputs.init_pop_host(prob, 1)
    be.begin_species(it, 1, 1.0, 1.0, prob.pmax, 1.0, 1.0)
    be.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
    be.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
    p_hi = m.inputs.pcut_hi(cfg.EN_PCUT_HI, m.constants.MP)
    for ip in range(1, NPC + 1):
        ns = be.run_pcut(ip, 0)
        if ns == 0: break
        be.new_pcut(max(N // ns, 1))
# warm
work(0, 1)
t = time.time()
for it in range(2, 2 + K): work(0, it)
t_seq = time.time() - t
if BLOCKS:
    for b in bes: b.set_launch(BLOCKS, 256)
t = time.time()
th = [threading.Thread(target=work, args=(k, 2 + k)) for k in range(K)]
[x.start() for x in th]; [x.join() for x in th]
t_con = time.time() - t
print(f"N={N} K={K} blocks={BLOCKS or 512}: sequential {t_seq:.3f} s ({t_seq/K*1e3:.0f} ms/iter), concurrent {t_con:.3f} s ({t_con/K*1e3:.0f} ms/iter), speedup {t_seq/t_con:.2f}")
