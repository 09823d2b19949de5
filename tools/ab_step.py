#!/usr/bin/env python3
"""Interleaved A/B of engine builds / switches on ONE device in ONE allocation (boxes of the pool differ by ~4 %):

    python tools/ab_step.py [--rounds 2] [--resolution R] label[:lib.so][:ENV=VAL,ENV=VAL] ...      (on the MI355X box)

Every variant runs tools/time_kernels.py (the headline 1M-tet mesh; 200 timed steps + per-kernel HIP events) in a
process of its own, `rounds` times, round-robin; prints per variant the step time and the patch / stress passes in us.
`lib.so` is a build of tools/build_variant.sh (empty: the in-tree library); timing only -- results are not checked."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    rounds, extra, limit = 2, [], 150
    while args and args[0].startswith("--"):
        if args[0] == "--rounds":
            rounds = int(args[1])
        elif args[0] == "--limit":
            limit = int(args[1])
        else:
            extra += args[:2]
        args = args[2:]
    variants = []
    for spec in args:
        parts = spec.split(":")
        label = parts[0]
        lib = parts[1] if len(parts) > 1 and parts[1] else None
        env = dict(kv.split("=", 1) for kv in parts[2].split(",")) if len(parts) > 2 and parts[2] else {}
        variants.append((label, lib, env))
    res = {v[0]: [] for v in variants}
    for _ in range(rounds):
        for label, lib, env in variants:
            e = dict(os.environ, **env)
            if lib:
                e["DES_HIP_LIB"] = os.path.join(ROOT, lib)
            else:
                e.pop("DES_HIP_LIB", None)
            try:
                # (a variant that hangs must not take the whole call with it: its process is killed after `limit` seconds)
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_kernels.py")] + extra, env=e, capture_output=True, text=True, timeout=limit)
            except subprocess.TimeoutExpired:
                res[label].append("TIMED OUT after %d s" % limit)
                print("%s: TIMED OUT" % label, flush=True)
                continue
            line = [l for l in out.stdout.splitlines() if "us per step" in l]
            if out.returncode or not line:
                res[label].append("FAILED: " + (out.stderr.strip().splitlines() or ["?"])[-1][:200])
                continue
            res[label].append(line[-1].split(": ", 1)[1])
            print("%s: %s" % (label, res[label][-1]), flush=True)
    print("---- by variant")
    for label, _, _ in variants:
        for r in res[label]:
            print("%-28s %s" % (label, r))


if __name__ == "__main__":
    main()
