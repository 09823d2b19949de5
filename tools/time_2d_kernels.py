#!/usr/bin/env python3
"""Per-kernel times of the 2-D step from a rocprofv3 kernel trace, without result checks (for A/B builds selected with
DES_HIP_LIB): python tools/time_2d_kernels.py [resolution_m]   (on the MI355X box)"""
import csv, glob, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = sys.argv[1] if len(sys.argv) > 1 else "250"
d = tempfile.mkdtemp(prefix="p2d", dir="/tmp")
env = dict(os.environ, TMPDIR="/tmp")
out = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                      os.path.join(ROOT, "tools", "time_2d.py"), res], capture_output=True, text=True, cwd="/tmp", env=env)
print([l for l in out.stdout.splitlines() if "triangles" in l])
f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(os.environ.get("TOP", "8"))]:
    print("  %-50s %6s calls %9.2f us" % (r["Name"].replace("des2d::(anonymous namespace)::", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
