#!/usr/bin/env python3
"""rocprofv3 --pmc passes over tools/time_2d.py (the 2-D engine), one pass per counter group, per-launch averages by kernel:

    python tools/pmc_2d.py OUT.txt RESOLUTION "SQ_WAVES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY" ...     (on the MI355X box)
"""
import collections, csv, glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path, res, groups = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
base = os.path.join(ROOT, "gpurun_out")
log = []
for g in groups:
    d = tempfile.mkdtemp(prefix="pmc2d_", dir=base if os.path.isdir(base) else None)
    cmd = ["rocprofv3", "--pmc"] + g.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.join(ROOT, "tools", "time_2d.py"), res]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", DES2D_TIME_STEPS="40"))
    if r.returncode:
        log.append("# group '%s' failed: %s" % (g, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "?"))
        continue
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            m = re.search(r"(k2p?_\w+)", row["Kernel_Name"])
            if m:
                acc[row["Counter_Name"]][m.group(1)].append(float(row["Counter_Value"]))
kern = [k for k in ("k2p_temp_dvoldt", "k2_stress", "k2_node_avg", "k2p_force", "k2_node_final", "k2_rotate_vol", "k2p_mass") if any(acc[c][k] for c in acc)]
with open(out_path, "w") as f:
    f.write("# rocprofv3 --pmc passes over 'tools/time_2d.py %s', per-launch averages by kernel (tools/pmc_2d.py)\n" % res)
    for l in log:
        f.write(l + "\n")
    f.write("%-30s" % "counter" + "".join("%17s" % k for k in kern) + "\n")
    for c in sorted(acc):
        f.write("%-30s" % c + "".join("%17.5g" % (sum(acc[c][k]) / len(acc[c][k])) if acc[c][k] else "%17s" % "-" for k in kern) + "\n")
print(open(out_path).read())
