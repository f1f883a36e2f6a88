"""First-contact GPU check: math/RNG bit parity and a small transport parity run
against the CPU oracle, plus a coarse timing.  (Superseded by tests/ -m gpu.)"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import _mcs_loader; m = _mcs_loader.load()
from mcs_amd import hip_backend
import orc, ctypes as ct

def bits(a): return np.ascontiguousarray(a).view(np.int64)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
NPC = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = m.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N)
prob = m.inputs.build_problem(cfg)
hb = hip_backend.HipBackend(0, debug_finals=True); hb.create(prob)
ob = orc.OracleBackend(m.capi, "det", nthreads=8); ob.create(prob)

# 1. math
rng = np.random.default_rng(0); n = 1_000_000
dp = ct.POINTER(ct.c_double)
def oev(fn, a, b=None):
    a = np.ascontiguousarray(a); b = a if b is None else np.ascontiguousarray(b); out = np.zeros_like(a)
    ob.lib.orc_eval_fn(m.capi.FN[fn], len(a), a.ctypes.data_as(dp), b.ctypes.data_as(dp), out.ctypes.data_as(dp)); return out
cases = {"sin": (rng.uniform(-10, 10, n), None), "cos": (rng.uniform(-10, 10, n), None), "asin": (rng.uniform(-1, 1, n), None),
         "acos": (rng.uniform(-1, 1, n), None), "atan2": (rng.normal(size=n), rng.normal(size=n)),
         "log10": (10 ** rng.uniform(-30, 30, n), None), "mod2pi": (rng.uniform(-20, 20, n), None),
         "sqrt": (10 ** rng.uniform(-40, 40, n), None), "div": (rng.normal(size=n), rng.normal(size=n)),
         "hypot1": (10 ** rng.uniform(-6, 10, n), None),
         "uniform": (np.floor(rng.uniform(0, 2**40, n)), np.floor(rng.uniform(0, 30000, n)))}
for k, (a, b) in cases.items():
    g = hb.eval_fn(k, a, b); o = oev(k, a, b)
    print(f"math {k:7s} mismatches {(bits(g) != bits(o)).sum()} / {n}", flush=True)

# 2. transport parity
for be in (hb, ob):
    be.begin_iteration(1)
    inj = m.inputs.init_pop_host(prob, 1)
    be.begin_species(1, 1, 1.0, 1.0, prob.pmax, 1.0, 1.0)
    be.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
    be.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
pg, po = hb.get_population(), ob.get_population()
for f in pg.fields():
    a, b = getattr(pg, f), getattr(po, f)
    print("init_pop", f, "equal" if np.array_equal(a.view(np.uint8), b.view(np.uint8)) else "DIFF", flush=True)
for ip in range(1, NPC + 1):
    t = time.time(); nsg = hb.run_pcut(ip, 0); tg = time.time() - t
    t = time.time(); nso = ob.run_pcut(ip, 0); to = time.time() - t
    fg, fo = hb.finals(), ob.finals()
    sg, lg = hb.get_saved(); so, lo = ob.get_saved()
    ok = {k: int((fg[k] != fo[k]).sum()) if fg[k].dtype != np.float64 else int((bits(fg[k]) != bits(fo[k])).sum()) for k in fg}
    sv = {f: int((getattr(sg, f).view(np.uint8) != getattr(so, f).view(np.uint8)).sum()) for f in sg.fields()}
    steps = int(fo['helix'].astype(np.int64).sum() + fo['retro'].astype(np.int64).sum())
    print(f"pcut {ip}: n={hb.pop_size()} saved gpu/orc {nsg}/{nso} lsave diff {(lg!=lo).sum()} finals mismatches {ok} saved-field byte diffs {sum(sv.values())} "
          f"steps {steps} gpu kernel {hb.last_kernel_ms():.2f} ms ({steps/(hb.last_kernel_ms()*1e-3+1e-12):.3e} steps/s) oracle {to*1e3:.0f} ms", flush=True)
    if nso == 0: break
    im = max(N // nso, 1)
    ng_, no_ = hb.new_pcut(im), ob.new_pcut(im)
    pg, po = hb.get_population(), ob.get_population()
    d = sum(int((getattr(pg, f).view(np.uint8) != getattr(po, f).view(np.uint8)).sum()) for f in pg.fields())
    print(f"   new_pcut i_mult={im}: n_new gpu/orc {ng_}/{no_} byte diffs {d}", flush=True)
Tg, Ig = hb.read_tallies(); To, Io = ob.read_tallies()
L = hb.layout
for name in L.offsets:
    a, b = L.view(Tg, name), L.view(To, name)
    scale = np.max(np.abs(b)) + 1e-300
    print(f"tally {name:22s} max|diff|/max|ref| = {np.max(np.abs(a-b))/scale:.3e}  nonzero ref {int((np.abs(b)>1e-90).sum())}")
print("i64 equal:", np.array_equal(Ig, Io), {k: (int(Ig[L.n_grid+v]), int(Io[L.n_grid+v])) for k, v in m.capi.IC.items()})
