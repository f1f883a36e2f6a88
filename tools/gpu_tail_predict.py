"""Can the long histories of a late pcut be told in advance?  In pcuts 20-34 of BASELINE config[1] the mean history is 150-300 steps and the
launch waits 4-5 ms for a thousand histories of 4000-10000 steps (frac 0.15).  The population there is a few thousand parents x hundreds of
identical copies: if the long histories come from particular parents (states), starting those first hides the tail behind the bulk -- pure
scheduling, same results.  This tool records, per late pcut, the state every particle starts from and its step count, and prints how P(long)
depends on the parent and on its state.  usage: python tools/gpu_tail_predict.py [N] [first_pcut] [long_steps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from conftest import mcs, make_problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P0 = int(sys.argv[2]) if len(sys.argv) > 2 else 20
LONG = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
prob = make_problem(N)
from mcs_amd import hip_backend
be = hip_backend.HipBackend(0, debug_finals=True)
be.create(prob)
be.begin_iteration(1)
inj = mcs.inputs.init_pop_host(prob, 1)
be.begin_species(1, 1, 1.0, 1.0, prob.pmax, 1.0, 1.0)
be.set_fluxes(inj.pxx_flux, inj.pxz_flux, inj.energy_flux)
be.init_pop(inj, 0, inj.n_pts_use, inj.n_pts_use)
im_prev = 1
rg0 = prob.rg0
for ip in range(1, len(prob.pcuts) + 1):
    pop = be.get_population() if ip >= P0 else None
    ns = be.run_pcut(ip, 0)
    if ip >= P0:
        f = be.finals()
        steps = f["helix"].astype(np.int64) + f["retro"].astype(np.int64)
        n = len(steps)
        lng = steps >= LONG
        parent = np.arange(n) // im_prev
        npar = parent.max() + 1
        per_par = np.bincount(parent, weights=lng, minlength=npar)
        order = np.argsort(-per_par)
        cum = np.cumsum(per_par[order]) / max(lng.sum(), 1)
        k50, k90 = int(np.searchsorted(cum, 0.5)) + 1, int(np.searchsorted(cum, 0.9)) + 1
        # what a uniform draw of the same number of long histories over the parents would give
        rng = np.random.default_rng(1)
        rnd = np.bincount(rng.integers(0, npar, int(lng.sum())), minlength=npar)
        cr = np.cumsum(np.sort(rnd)[::-1]) / max(lng.sum(), 1)
        r50, r90 = int(np.searchsorted(cr, 0.5)) + 1, int(np.searchsorted(cr, 0.9)) + 1
        print(f"pcut {ip:2d}: n {n} parents {npar} (x{im_prev}) mean steps {steps.mean():7.1f} max {steps.max():6d} long(>={LONG}) {int(lng.sum()):6d} = {100 * lng.mean():.3f} % | "
              f"parents holding 50 % / 90 % of the long ones: {k50} / {k90} ({100 * k50 / npar:.1f} % / {100 * k90 / npar:.1f} %; uniform: {r50} / {r90})", flush=True)
        x = pop.x_PT_cm / rg0
        mu = pop.pb_pf / pop.ptot_pf
        for name, v in (("x / rg0", x), ("mu = pb / p", mu), ("acctime", pop.acctime_sec)):
            qs = np.quantile(v, np.linspace(0, 1, 9))
            row = []
            for a, b in zip(qs[:-1], qs[1:]):
                sel = (v >= a) & (v <= b)
                row.append(f"[{a:9.3g},{b:9.3g}] {100 * lng[sel].mean():6.3f}%")
            print(f"        P(long) by octile of {name:12s}: " + "  ".join(row), flush=True)
        # list-scheduling model of the launch: 2048 waves x 64 lanes take the next particle of the queue when they are free; one time unit
        # per step.  Queue in index order (today) against descending mu (parents sorted, copies adjacent) and against the clairvoyant order.
        import heapq
        def makespan(order):
            L = 131072
            st = steps[order]
            if len(st) <= L:
                return int(st.max())
            h = st[:L].tolist(); heapq.heapify(h)
            for v in st[L:].tolist():
                heapq.heapreplace(h, h[0] + v)
            return int(max(h))
        ideal = max(int(steps.max()), int(steps.sum() // 131072))
        m_idx = makespan(np.arange(n))
        pmu = mu[::im_prev][:npar] if im_prev > 1 else mu
        par_sorted = np.argsort(-pmu, kind="stable")
        order_mu = (par_sorted[:, None] * im_prev + np.arange(im_prev)[None, :]).reshape(-1)
        order_mu = order_mu[order_mu < n]
        m_mu = makespan(order_mu)
        m_lpt = makespan(np.argsort(-steps, kind="stable"))
        print(f"        makespan in steps (131072 lanes): index order {m_idx}, descending mu {m_mu}, longest first {m_lpt}, lower bound {ideal}  -> mu order saves {100 * (1 - m_mu / m_idx):.1f} %", flush=True)
        dn = (pop.downstream == 1)
        print(f"        P(long | downstream flag) {100 * lng[dn].mean() if dn.any() else 0:.3f} %  ({dn.mean() * 100:.1f} % of the particles);  P(long | not) {100 * lng[~dn].mean() if (~dn).any() else 0:.3f} %", flush=True)
    if ns == 0:
        break
    im_prev = max(N // ns, 1)
    be.new_pcut(im_prev)
