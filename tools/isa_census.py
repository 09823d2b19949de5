#!/usr/bin/env python3
"""Static instruction census of the gfx950 code of one or more kernels (cross-compiled, no GPU needed).

    python tools/isa_census.py [--asm FILE] [--src des_dev.hip|des_dev2d.hip] [-D...] PATTERN [PATTERN ...]

Compiles the translation unit to assembly (hipcc -S --cuda-device-only, the library's own flags; or reads --asm), finds the
kernels whose demangled name contains PATTERN and prints per kernel the instruction counts by class -- fp64 arithmetic
(add / mul / fma), the division / reciprocal / sqrt sequences (v_div_scale, v_div_fmas, v_div_fixup, v_rcp, v_rsq, v_sqrt),
other VALU (moves, selects, compares, integer and 64-bit address arithmetic), LDS, vector memory (loads / stores / LDS-DMA),
scalar -- for the whole kernel and for every LOOP of it (a backward branch to a label), innermost first.  Static counts:
a loop's body is counted once.  The dynamic per-wavefront figures come from the SQ counters (tools/summarize_counters.py);
this tool says what the instructions ARE.  Variants (-DDES_EXP_...) give a differential census: build with a piece of the
element code left out and the difference is that piece's share.

With --vmem-after-lds-dma it also checks the invariant of the pipelined stress update (passes/e2.hpp): on the straight-line
path from the last LDS-DMA request of the tile loop to the loop's back edge there are at least N vector-memory instructions
(the `s_waitcnt vmcnt(12)` at the top of the next tile relies on at least 13 younger stores).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dynearthsol_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value", "-Wno-pass-failed",
         "-Wno-unused-command-line-argument", "-I" + os.path.join(ROOT, "include")]

DIV = ("v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rcp_f32", "v_rcp_iflag_f32")
F64 = ("v_add_f64", "v_mul_f64", "v_fma_f64", "v_max_f64", "v_min_f64", "v_ldexp_f64", "v_frexp", "v_fract_f64", "v_trunc_f64",
       "v_floor_f64", "v_ceil_f64", "v_rndne_f64", "v_cvt_f64", "v_cmp_", "v_cmpx_")


def classify(op):
    if op.startswith("v_"):
        if op.startswith(DIV):
            return "div/rcp/sqrt"
        if op.startswith(("v_add_f64", "v_mul_f64", "v_fma_f64")):
            return "fp64 add/mul/fma"
        if "f64" in op:
            return "fp64 other"
        if op.startswith(("v_cmp", "v_cndmask")):
            return "cmp/select"
        if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_swap")):
            return "move"
        return "int/addr VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_load_lds", "buffer_load")) and "lds" in op:
        return "LDS-DMA"
    if op.startswith(("global_load", "flat_load", "buffer_load", "scratch_load")):
        return "vmem load"
    if op.startswith(("global_store", "flat_store", "buffer_store", "scratch_store", "global_atomic", "flat_atomic")):
        return "vmem store/atomic"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_barrier")):
        return "branch/barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "scalar load"
    if op.startswith("s_"):
        return "scalar ALU"
    return "other"


ORDER = ["fp64 add/mul/fma", "fp64 other", "div/rcp/sqrt", "cmp/select", "move", "int/addr VALU", "LDS", "LDS-DMA", "vmem load",
         "vmem store/atomic", "scalar load", "scalar ALU", "s_waitcnt", "branch/barrier", "other"]


def build_asm(src, defs):
    out = tempfile.NamedTemporaryFile(prefix="des_isa_", suffix=".s", delete=False).name
    cmd = ["hipcc"] + FLAGS + defs + ["-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
    if r.returncode != 0:
        sys.exit(r.stderr[-3000:])
    return out


def kernels(path):
    """{mangled name: [(label or None, opcode, text)]} for every function of the assembly file"""
    funcs, cur, name = {}, None, None
    for line in open(path):
        s = line.strip()
        m = re.match(r"^(_Z[\w$.]+):", s)
        if m and not s.startswith(".L"):
            name = m.group(1)
            cur = funcs.setdefault(name, [])
            continue
        if cur is None or not s or s.startswith((";", "//")):
            continue
        if s.startswith(".Lfunc_end") or s.startswith(".section") or s.startswith(".amdhsa_kernel"):
            if s.startswith(".Lfunc_end"):
                cur = None
            continue
        m = re.match(r"^(\.LBB[\w]+):", s)
        if m:
            cur.append((m.group(1), None, s))
            continue
        if s.startswith("."):
            continue
        op = s.split()[0]
        cur.append((None, op, s))
    return funcs


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def count(items):
    c = {}
    for lab, op, _ in items:
        if op:
            k = classify(op)
            c[k] = c.get(k, 0) + 1
    return c


def loops(items):
    """[(start index, end index)] of every backward branch target..branch, innermost (shortest) first"""
    pos = {lab: i for i, (lab, op, _) in enumerate(items) if lab}
    found = []
    for i, (lab, op, text) in enumerate(items):
        if op and op.startswith(("s_cbranch", "s_branch")):
            tgt = text.split()[-1]
            if tgt in pos and pos[tgt] < i:
                found.append((pos[tgt], i))
    return sorted(set(found), key=lambda ab: ab[1] - ab[0])


def show(title, c):
    valu = sum(v for k, v in c.items() if k in ORDER[:6])
    total = sum(c.values())
    print("  %-34s total %5d | VALU %5d" % (title, total, valu))
    print("      " + "  ".join("%s %d" % (k, c[k]) for k in ORDER if c.get(k)))


def vmem_after_last_dma(items):
    """for the loop that holds LDS-DMA requests: vector-memory instructions between its last LDS-DMA request and its back edge"""
    best = None
    for a, b in loops(items):
        body = items[a:b + 1]
        dma = [i for i, (lab, op, _) in enumerate(body) if op and classify(op) == "LDS-DMA"]
        if not dma:
            continue
        n = sum(1 for lab, op, _ in body[dma[-1] + 1:] if op and classify(op) in ("vmem load", "vmem store/atomic"))
        best = n if best is None else min(best, n)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("patterns", nargs="+")
    ap.add_argument("--asm", default=None)
    ap.add_argument("--src", default="des_dev.hip")
    ap.add_argument("--loops", type=int, default=4, help="how many of the kernel's largest loops to print")
    ap.add_argument("--vmem-after-lds-dma", action="store_true")
    ap.add_argument("-D", dest="defs", action="append", default=[])
    a = ap.parse_args()
    path = a.asm or build_asm(a.src, ["-D" + d for d in a.defs])
    funcs = kernels(path)
    names = demangle(list(funcs))
    for mangled, items in funcs.items():
        dn = names[mangled].replace("des_hip::", "").replace("void ", "")
        short = dn.split("(")[0]
        if not any(p in short for p in a.patterns):
            continue
        print(short)
        show("whole kernel (static)", count(items))
        ls = loops(items)
        for a0, b0 in sorted(ls, key=lambda ab: ab[0] - ab[1])[:a.loops]:
            show("loop %s (%d lines)" % (items[a0][0], b0 - a0), count(items[a0:b0 + 1]))
        if a.vmem_after_lds_dma:
            print("  vmem instructions behind the last LDS-DMA request of the tile loop: %s" % vmem_after_last_dma(items))
    if not a.asm:
        os.unlink(path)


if __name__ == "__main__":
    main()
