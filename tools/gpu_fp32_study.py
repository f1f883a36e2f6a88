"""Tolerance study of the fp32-state variant of K1 (BASELINE config[4]) against the fp64 path on the GPU.

For each case (config[1]: 1e6 protons; mixed species: p + He + e- with radiative losses and ion -> electron energy transfer)
three full iterations are run through driver.run: fp64 (iteration 1 seeds), fp64 with the seeds of iteration 2 (an
independent Monte-Carlo realisation: the noise floor any comparison has to be read against) and the fp32-state kernel
(iteration 1 seeds).  Compared: the downstream dN/dp (sum over angle bins of psd, zones shock+3 and shock+10) -- L-inf
and RMS deviation of log10 dN/dp over the power-law range, the fitted slope next to the Keshet & Waxman index -- the
escaping spectrum, the fluxes, the population sizes per pcut and the rate.
usage: python tools/gpu_fp32_study.py [N] [mixed]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import mcs
from mcs_amd import hip_backend
from test_physics import keshet_waxman_slope

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
MIXED = "mixed" in sys.argv[2:]
ME_MP = mcs.constants.ME / mcs.constants.MP


def run(fp32, first_iter):
    kw = dict(N_PTS_INJ=N, N_PTS_PCUT=N, N_PTS_PCUT_HI=N, num_iterations=2, state_fp32=fp32)
    if MIXED:
        kw.update(species=[mcs.inputs.Species(1.0, 1.0, 1e6, 1.0), mcs.inputs.Species(4.0, 2.0, 1e6, 0.1),
                           mcs.inputs.Species(ME_MP, -1.0, 1e6, 1.2)],
                  energy_transfer_frac=0.1, radiation_losses=True)
    prob = mcs.inputs.build_problem(mcs.inputs.Config(**kw))
    be = hip_backend.HipBackend(0); be.create(prob)
    t0 = time.perf_counter()
    res = mcs.driver.run(prob, be, None, n_itrs=1, first_iter=first_iter)
    dt = time.perf_counter() - t0
    kms = sum(s.kernel_ms for s in res.stats)
    be.destroy()
    return prob, res, dt, kms


def spectra(prob, L, T, zone):
    P = prob.params
    dn = L.view(T, "psd")[zone - 1].sum(axis=0)
    mb = prob.psd_mom_bounds
    k = np.arange(1, P.num_psd_mom_bins)
    dndp = dn[k] / (10.0 ** mb[k + 1] - 10.0 ** mb[k])
    pc = 10.0 ** (0.5 * (mb[k] + mb[k + 1]))
    return pc, dndp


def compare(name, prob, L, Ta, Tb, species_idx):
    P = prob.params
    out = []
    for zone in (P.i_shock + 3, P.i_shock + 10):
        pc, a = spectra(prob, L, Ta, zone); _, b = spectra(prob, L, Tb, zone)
        sel = (pc > 30.0) & (pc < 3.0e4) & (a > 0) & (b > 0)
        d = np.log10(a[sel]) - np.log10(b[sel])
        sa = np.polyfit(np.log10(pc[sel]), np.log10(a[sel]), 1)[0]; sb = np.polyfit(np.log10(pc[sel]), np.log10(b[sel]), 1)[0]
        out.append(f"zone shock+{zone - P.i_shock}: Linf |dlog10 dN/dp| = {np.max(np.abs(d)):.4f}, rms = {np.sqrt(np.mean(d * d)):.4f}, "
                   f"slopes {sa:.4f} vs {sb:.4f}")
    ea, eb = L.view(Ta, "esc_psd_down").sum(axis=0), L.view(Tb, "esc_psd_down").sum(axis=0)
    sel = (ea > 1e-90) & (eb > 1e-90)
    d = np.log10(ea[sel]) - np.log10(eb[sel])
    out.append(f"downstream escape spectrum (sum over angle): Linf |dlog10| = {np.max(np.abs(d)):.4f} over {int(sel.sum())} bins, "
               f"total {ea.sum() / eb.sum() - 1:+.3e}")
    for fl in ("pxx_flux", "energy_flux"):
        x, y = L.view(Ta, fl), L.view(Tb, fl)
        out.append(f"{fl}: max |a - b| / max |b| = {np.max(np.abs(x - y)) / np.max(np.abs(y)):.3e}")
    print(f"  [{name}]"); [print("     " + o) for o in out]


print(f"== fp32-state tolerance study, N = {N}, {'p + He + e- (radiative losses, ion -> electron energy transfer)' if MIXED else 'protons (config[1])'}")
p64, r64, t64, k64 = run(False, 1)
_, r64b, _, _ = run(False, 2)
p32, r32, t32, k32 = run(True, 1)
L = mcs.capi.Layout(p64.params)
print(f"   Keshet & Waxman dN/dp index: {keshet_waxman_slope(p64.params):.4f}")
n_sp = len(p64.cfg.species)
for isp in range(n_sp):
    Ta, Tb, Tc = r64.per_species[isp][2], r64b.per_species[isp][2], r32.per_species[isp][2]
    print(f" species {isp + 1}:")
    compare("fp64 seeds 2 vs fp64 seeds 1 (Monte-Carlo noise floor)", p64, L, Tb, Ta, isp)
    compare("fp32 state vs fp64 (same seeds)", p64, L, Tc, Ta, isp)
s64 = {(s.i_ion, s.i_pcut): s.n_saved for s in r64.stats}; s32 = {(s.i_ion, s.i_pcut): s.n_saved for s in r32.stats}
s64b = {(s.i_ion, s.i_pcut): s.n_saved for s in r64b.stats}
thr = max(N // 50, 1000)
dev = max(abs(v - s32.get(k, 0)) / v for k, v in s64.items() if v > thr)
devb = max(abs(v - s64b.get(k, 0)) / v for k, v in s64.items() if v > thr)
print(f" saved particles per pcut (pcuts with > {thr} saved): largest relative difference fp32 vs fp64 {dev:.3e} (fp64 seeds 2 vs seeds 1: {devb:.3e}); "
      f"pcuts reached {len(s64)} (fp64) {len(s64b)} (fp64 seeds 2) {len(s32)} (fp32)")
st64, st32 = r64.steps_helix + r64.steps_retro, r32.steps_helix + r32.steps_retro
print(f" steps: fp64 {st64:.4e} in {k64:.0f} ms of kernels ({st64 / k64 / 1e-3:.3e} steps/s); fp32 {st32:.4e} in {k32:.0f} ms ({st32 / k32 / 1e-3:.3e} steps/s)")
ng, IC = p64.params.n_grid, mcs.capi.IC
for nm in ("HELIX_CAP", "PPERP_CLAMP", "PSP_CLAMP", "ZONE_FAIL", "RETRO_CAP", "REASON1", "REASON2", "REASON3", "REASON4"):
    print(f"   counter {nm}: fp64 {int(r64.tallies_i64[ng + IC[nm]])}  fp64 seeds 2 {int(r64b.tallies_i64[ng + IC[nm]])}  fp32 {int(r32.tallies_i64[ng + IC[nm]])}")
