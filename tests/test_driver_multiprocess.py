"""The multi-GPU path of the host driver, rehearsed on CPU: world_size 2 and 3 over gloo
with the CPU oracle as backend must reproduce the single-process run (contiguous particle
shards, global RNG keys, all-gather of n_saved, one sum-all-reduce of the tallies)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, mcs, orc


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _launch(world, out, N, npc, extra=()):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), out, "oracle", str(N), str(npc), *extra],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    try:
        for p in procs:
            o, _ = p.communicate(timeout=300)
            assert p.returncode == 0, o.decode()[-3000:]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def _single(N, npc, two=False):
    kw = {}
    if two:
        kw = dict(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1)], energy_transfer_frac=0.1)
    cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=2, **kw)
    prob = mcs.inputs.build_problem(cfg)
    be = orc.OracleBackend(mcs.capi, "det", 1)
    be.create(prob)
    return prob, mcs.driver.run(prob, be, None, n_itrs=2, max_pcuts=npc)


@pytest.mark.parametrize("world,two", [(2, False), (3, False), (2, True)])
def test_sharded_run_equals_single_process(tmp_path, world, two):
    N, npc = 150, 7
    out = str(tmp_path / f"w{world}.npz")
    _launch(world, out, N, npc, ("2",) if two else ())
    prob, ref = _single(N, npc, two)
    got = np.load(out)
    stats_ref = np.array([[s.i_iter, s.i_ion, s.i_pcut, s.n_pts_use, s.n_saved, s.i_mult] for s in ref.stats])
    assert np.array_equal(got["stats"], stats_ref)              # same populations at every pcut
    assert np.array_equal(got["i"], ref.tallies_i64)            # same steps, crossings, exits
    L = mcs.capi.Layout(prob.params)
    from conftest import assert_tallies_close
    assert_tallies_close(L, got["f"], ref.tallies_f64, rtol=1e-12)
