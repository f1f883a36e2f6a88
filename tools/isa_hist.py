#!/usr/bin/env python3
"""Instruction histogram of the transport kernel's common pass, from the gfx950 assembly.

    make -C montecarloscattering.jl_amd/csrc asm        # writes /tmp/mcs_transport-hip-amdgcn-amd-amdhsa-gfx950.s
    python tools/isa_hist.py [file.s] [kernel]

The particle loop is the largest backward branch of the kernel.  Its common pass is the chain of basic blocks a
wave runs through when no housekeeping is due and no lane has rare work: from the loop header, the housekeeping
branch and the rare-region branch are taken over their (out-of-line or skipped) bodies, everything else falls
through.  The script walks that chain -- at a conditional branch it follows the edge that SKIPS code (forward
target) when the branch guards a skippable body (s_cbranch_execz / scc / vccz over a region), which is what
happens when nothing is due -- and classifies every instruction on it.  Cycle floor: every VALU/SALU issue of a
lone wave costs ~4 cycles whatever the dependency (profiles/r01_ubench_issue_latency.txt: fp64 FMA 4.1, MUL/ADD
5.2, RCP 16), an untaken conditional region ~36.
"""
import collections
import os
import re
import sys


def parse(path, kernel):
    """Instructions of the kernel in layout order, each with the label that precedes it (or None)."""
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    out, label = [], None
    for l in lines[start + 1:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            label = (m.group(1), m.group(2)); continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        t = t.split(";")[0].strip()
        if t:
            out.append((t, label)); label = None
    return out


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        if re.search(r"_f64", op):
            if op.startswith(("v_rcp", "v_rsq", "v_sqrt")):
                return "valu_f64_trans"
            if op.startswith("v_fma"):
                return "valu_f64_fma"
            if op.startswith("v_cmp") or op.startswith("v_cmpx"):
                return "valu_f64_cmp"
            return "valu_f64_other"
        if op.startswith(("v_cmp", "v_cmpx")):
            return "valu_int_cmp"
        if op.startswith("v_mad_u64_u32"):
            return "valu_mad_u64"
        if op.startswith(("v_cndmask", "v_mov", "v_readlane", "v_writelane", "v_readfirstlane", "v_accvgpr")):
            return "valu_move_select"
        if op.startswith("v_cvt"):
            return "valu_cvt"
        return "valu_int_other"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op == "s_nop":
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem_" + op.split("_")[0]
    return "other"


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/mcs_transport-hip-amdgcn-amd-amdhsa-gfx950.s"
    kernel = sys.argv[2] if len(sys.argv) > 2 else "mcs_k_transport_plain"
    ins = parse(path, kernel)
    # the particle loop: the depth-1 loop header with child loops (the rare code's loops); its common pass is the
    # fall-through chain from the header -- conditional branches guard out-of-line rare code (MCS_UNLIKELY) or exec-mask
    # regions that are entered, so they fall through; an unconditional s_branch is followed -- until the header comes again
    label_at = {lab[0]: i for i, (t, lab) in enumerate(ins) if lab}
    headers = [i for i, (t, lab) in enumerate(ins) if lab and "This Loop Header: Depth=1" in lab[1]]
    best = None
    for h in headers:
        name = ins[h][1][0]
        body, i, seen = [], h, set()
        while i < len(ins) and i not in seen:
            seen.add(i)
            t, lab = ins[i]
            if lab and lab[0] == name and body:
                break
            body.append((t, lab))
            m = re.match(r"s_branch\s+(\S+)", t)
            if m:
                i = label_at[m.group(1)]
                continue
            i += 1
        n64 = sum(1 for t, _ in body if "_f64" in t.split()[0])
        if best is None or n64 > best[0]:
            best = (n64, name, body)
    _, name, body = best
    hist = collections.Counter(classify(t) for t, _ in body)
    total = len(body)
    # the loop trip holds MCS_PASSES_PER_ITER common passes behind one header (mcs_transport.hip)
    passes = 1
    try:
        src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "montecarloscattering.jl_amd", "csrc", "mcs_transport.hip")).read()
        passes = int(re.search(r"#define MCS_PASSES_PER_ITER (\d+)", src).group(1))
    except Exception:
        pass
    regions = [t for t, _ in body if t.startswith("s_cbranch")]
    inline = sum(1 for k in range(len(body)) if body[k][0].startswith("s_cbranch_execz") and
                 any(lab and lab[0] == body[k][0].split()[1] for _, lab in body[k:]))
    print(f"{kernel}: particle loop header {name}; one trip of the loop (header + {passes} common pass{'es' if passes > 1 else ''}, nothing rare due) = "
          f"{total} instructions laid out up to the back edge; counts below are per TRIP, the floors per PASS")
    for k in sorted(hist, key=lambda k: -hist[k]):
        print(f"  {k:20s} {hist[k]:5d}")
    valu = sum(v for k, v in hist.items() if k.startswith("valu"))
    salu = hist["salu"] + hist["s_nop"] + hist["s_waitcnt"]
    other = total - valu - salu - hist["branch"]
    print(f"VALU {valu}  (fp64 {sum(v for k, v in hist.items() if k.startswith('valu_f64'))}, "
          f"Philox mad_u64 {hist['valu_mad_u64']}), SALU+nop+wait {salu}, branches {hist['branch']} "
          f"({inline} in-line conditional regions, the rest guard out-of-line rare code or close the loop), LDS/VMEM {other}")
    print("branches:", "; ".join(regions))
    issue = 4.4
    lone = ((valu + salu + other) * issue + hist["branch"] * 36) / passes
    print(f"cycle floor of one pass, lone wave : (({valu} + {salu} + {other}) x {issue} + {hist['branch']} x 36) / {passes} = {lone:.0f} cycles "
          f"= {lone / 2.4e3:.2f} us at 2.4 GHz  -> a 10^4-pass history (helix cap) = {lone / 2.4e3 * 1e4 / 1e3:.1f} ms")
    two = valu * 4.0 / passes
    print(f"VALU-issue floor (2 waves per SIMD, everything else overlapped): {valu} x 4 / {passes} = {two:.0f} cycles per wave-pass "
          f"-> {1024 * 2.4e9 / two * 64 / 1e10:.1f}e10 steps/s chip-wide with 64 live lanes; x 400 flop = "
          f"{1024 * 2.4e9 / two * 64 * 400 / 78.6e12:.2f} of the fp64 VALU peak")


if __name__ == "__main__":
    main()
