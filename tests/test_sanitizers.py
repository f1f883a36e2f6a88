"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool:
sanitizers run on the CPU build only).  The transport restatement, the tally consumers and the iter_finalize twin are
the same sources the parity tests trust; a two-species run (protons + electrons: radiative losses, energy transfer,
injection probability < 1, amplified downstream field) through two iterations with the profile update must finish
without a single report."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SCRIPT = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/oracle"); sys.path.insert(0, %(root)r + "/tests")
import _mcs_loader; mcs = _mcs_loader.load()
import orc
ME_MP = mcs.constants.ME / mcs.constants.MP
N = 200
cfg = mcs.inputs.Config(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=2,
                        species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(ME_MP, -1.0, 1e6, 1.0)],
                        energy_transfer_frac=0.1, radiation_losses=True, INJFR=[0.7, 1.0], b_field_turbulence=1.0)
prob = mcs.inputs.build_problem(cfg)
be = orc.OracleBackend(mcs.capi, "san", nthreads=2); be.create(prob)
res = mcs.driver.run(prob, be, None, n_itrs=2, smoothing=mcs.iter_finalize.SmoothingConfig(True))
print("SAN_OK", res.steps_helix + res.steps_retro, len(res.iter_finals))
'''


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_is_clean_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "san"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=asan + " " + ubsan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "SAN_OK" in r.stdout, out[-3000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-3000:]
