#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (--2d: the 2-D engine's translation unit)
args = sys.argv[1:]
src = os.path.join(root, "dynearthsol_amd", "csrc", "des_dev2d.hip" if "--2d" in args else "des_dev.hip")
args = [a for a in args if a != "--2d"]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value",
       "-I" + os.path.join(root, "include"), "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + args
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: +([^:]+): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k in ("Function Name", "Name"):
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
        cur = {"name": name.replace("void ", "").replace("des_hip::", "").replace("des2d::", "")}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print("%-48s %5s %5s %7s %6s %5s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "occ"))
for r in rows:
    print("%-48s %5s %5s %7s %6s %5s" % (r["name"][:48], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
                                          r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
