#!/usr/bin/env python3
"""Fail the build when the register allocation of the transport kernels is not what the sources and DESIGN.md say.

`make -C montecarloscattering.jl_amd/csrc` compiles mcs_transport.hip with -Rpass-analysis=kernel-resource-usage and
keeps the remarks in csrc/mcs_transport.resources.txt; this script parses them and compares with LIMITS below.  Why a
hard check: a build of K1 that spilled differently once miscompiled the l_save byte store (ROCm 7.2; csrc/Makefile), and
the kernel sits at the edge of the register file -- an innocent edit moves spills into the common pass.
usage: python tools/check_resources.py [resources.txt]   (exit code 1 on violation)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, "montecarloscattering.jl_amd", "csrc", "mcs_transport.resources.txt")

# kernel -> (max VGPRs, max VGPR spills, max scratch bytes/lane, required occupancy [waves/SIMD], max LDS bytes/block)
LIMITS = {
    # round 4: suspend / resume is a kernel of its own (mcs_k_transport_sliced), the shipping kernels are back at <= 76 B/lane
    "mcs_k_transport_plain": dict(vgprs=256, vgpr_spill=56, scratch=76, occupancy=2, lds=81920),
    "mcs_k_transport": dict(vgprs=256, vgpr_spill=56, scratch=76, occupancy=2, lds=81920),
    "mcs_k_transport_lossy": dict(vgprs=256, vgpr_spill=56, scratch=76, occupancy=2, lds=81920),
    "mcs_k_transport_plain_etf": dict(vgprs=256, vgpr_spill=56, scratch=76, occupancy=2, lds=81920),
    "mcs_k_transport_sliced": dict(vgprs=256, vgpr_spill=64, scratch=112, occupancy=2, lds=81920),
    # (+ the tail loop, round 4: 70 spills; they stay in the import / export code of the sliced form)
    "mcs_k_transport_plain_sliced": dict(vgprs=256, vgpr_spill=72, scratch=112, occupancy=2, lds=81920),
    "mcs_k_transport_lossy_sliced": dict(vgprs=256, vgpr_spill=64, scratch=112, occupancy=2, lds=81920),
    "mcs_k_transport_plain_etf_sliced": dict(vgprs=256, vgpr_spill=72, scratch=112, occupancy=2, lds=81920),
    # the wave-specialised kernels (512-thread blocks, one per CU: up to 128 KB of LDS with the particle pool)
    "mcs_k_transport_ws": dict(vgprs=256, vgpr_spill=24, scratch=76, occupancy=2, lds=131072),
    "mcs_k_transport_ws_etf": dict(vgprs=256, vgpr_spill=24, scratch=76, occupancy=2, lds=131072),
    # round 4: the retro walk is inlined in the fp32 kernels -- the 192-208 B/lane of rounds 2-3 were the frame of that one call
    # (the 34-word lane state and the RNG stream passed by reference), not spills
    "mcs_k_transport_f32": dict(vgprs=168, vgpr_spill=32, scratch=96, occupancy=3, lds=54613),
    "mcs_k_transport_f32_lossy": dict(vgprs=168, vgpr_spill=32, scratch=96, occupancy=3, lds=54613),
    "mcs_k_transport_f32_loop": dict(vgprs=168, vgpr_spill=0, scratch=0, occupancy=3, lds=40960),
    "mcs_k_transport_f32_loop_exact": dict(vgprs=168, vgpr_spill=0, scratch=16, occupancy=3, lds=40960),
}


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("sgprs", r"TotalSGPRs: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else DEFAULT
    if not os.path.exists(path):
        print(f"check_resources: {path} not found -- build the library first (make -C montecarloscattering.jl_amd/csrc)")
        return 1
    got = parse(path)
    bad = 0
    for k, lim in LIMITS.items():
        r = got.get(k)
        if r is None:
            print(f"check_resources: kernel {k} not found in {path}"); bad += 1; continue
        line = (f"{k}: {r.get('vgprs')} VGPRs, {r.get('vgpr_spill')} VGPR spills, {r.get('sgpr_spill')} SGPR spills (to VGPR lanes), "
                f"{r.get('scratch')} B/lane scratch, {r.get('occupancy')} waves/SIMD, {r.get('lds')} B LDS/block")
        errs = []
        if r.get("vgprs", 0) > lim["vgprs"]: errs.append("VGPRs")
        if r.get("vgpr_spill", 0) > lim["vgpr_spill"]: errs.append("VGPR spills")
        if r.get("scratch", 0) > lim["scratch"]: errs.append("scratch")
        if r.get("occupancy", 0) != lim["occupancy"]: errs.append("occupancy")
        if r.get("lds", 0) > lim["lds"]: errs.append("LDS")
        print(("FAIL " if errs else "ok   ") + line + (f"   -- over the documented limit: {', '.join(errs)} {lim}" if errs else ""))
        bad += bool(errs)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
