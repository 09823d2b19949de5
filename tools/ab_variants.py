#!/usr/bin/env python3
"""A/B of kernel builds on ONE device in ONE allocation: for each library given on the
command line run bench.py's per-kernel timing twice, interleaved, and print µs per call."""
import json, os, subprocess, sys
libs = sys.argv[1:]
res = {}
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, DES_HIP_LIB=os.path.abspath(lib))
        out = subprocess.check_output([sys.executable, "bench.py", "--steps", "60", "--warmup", "10", "--cpu-steps", "0", "--no-large-series", "--no-elide-compare"], env=env)
        r = json.loads(out.decode().strip().splitlines()[-1])
        res.setdefault(lib, []).append((r["ms_per_step"], r["config"]["kernel_ms_per_call"]))
names = ["E1_geom_rotate_strainrate", "N1_mass_temperature_dvoldt", "E2_update_stress", "N2_nmd_gather", "E3_nmd_force", "N3_force_velocity_coord"]
for lib, runs in res.items():
    print(os.path.basename(lib), " ms/step:", " ".join("%.4f" % r[0] for r in runs),
          " | " + "  ".join("%s %s" % (n.split("_")[0], "/".join("%.1f" % (1e3 * r[1][n]) for r in runs)) for n in names))
