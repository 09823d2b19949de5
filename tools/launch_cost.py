#!/usr/bin/env python3
"""What does a small launch cost the step?  Times the bench step with one of the O(surface) / O(1)
launches left out at a time (DES_EXP_SKIP, engine/launch.hpp -- the results of those runs are wrong,
only their timing is used): the difference is what fusing that launch away could gain at most.
Meshes: the 1M-tet headline mesh and the 137k-tet strong-scaling shard size.

  python tools/launch_cost.py            (on the MI355X box; ~1 min)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(skip, extra):
    env = dict(os.environ)
    env.pop("DES_EXP_SKIP", None)
    if skip:
        env["DES_EXP_SKIP"] = skip
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "400", "--warmup", "40", "--cpu-steps", "0", "--no-large-series", "--no-elide-compare",
                          "--no-profile", "--no-ceiling"] + extra, capture_output=True, text=True, env=env, check=True)
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])["config"]["hip_event_ms_per_step"] * 1e3


for name, extra in (("1,001,310 tets (TetGen)", []), ("137,200 tets (regular)", ["--resolution", "1428.5714285714287"])):
    base = [run(None, extra) for _ in range(2)]
    print("%s: %.1f / %.1f us per step with every launch" % (name, base[0], base[1]))
    for skip in ("e2r", "s3", "s2", "dt", "e2r,s3,s2,dt"):
        t = run(skip, extra)
        print("   without %-14s %.1f us  (%+.1f)" % (skip, t, t - min(base)), flush=True)
