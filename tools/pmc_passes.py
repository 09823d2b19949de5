#!/usr/bin/env python3
"""rocprofv3 --pmc passes over bench.py, one pass per counter group (`--kernel-trace` only beside it), and the
per-launch average of every counter by kernel:

    python tools/pmc_passes.py OUT.txt "SQ_WAVES SQ_INSTS_VALU" "MemUnitBusy" ... [-- bench args]     (on the MI355X box)
"""
import collections
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.measure_traffic import short  # noqa: E402

NAMES = {"E1_geom_rotate_strainrate": "E1", "EN1_mass_temperature_dvoldt": "EN1", "E2_update_stress": "E2",
         "E2G_geom_rotate_update_stress": "E2G", "EN2_nmd_gather": "EN2", "EN3_force_nodes": "EN3", "k_s2": "S2", "k_s3_finalize": "S3"}


def main():
    args = sys.argv[1:]
    bench_args = []
    if "--" in args:
        k = args.index("--")
        args, bench_args = args[:k], args[k + 1:]
    out_path, groups = args[0], args[1:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    base = os.path.join(ROOT, "gpurun_out")
    log = []
    for g in groups:
        d = tempfile.mkdtemp(prefix="pmcp_", dir=base if os.path.isdir(base) else None)
        cmd = ["rocprofv3", "--pmc"] + g.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2", "--cpu-steps", "0", "--no-large-series", "--no-elide-compare", "--no-profile", "--no-ceiling"] + bench_args
        r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        if r.returncode:
            log.append("# group '%s' failed: %s" % (g, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "?"))
            continue
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                k = NAMES.get(short(row["Kernel_Name"]))
                if k:
                    acc[row["Counter_Name"]][k].append(float(row["Counter_Value"]))
        sys.stderr.write("group done: %s\n" % g); sys.stderr.flush()
    kern = [k for k in ("EN1", "E2G", "EN2", "EN3", "E1", "E2", "S2", "S3") if any(acc[c][k] for c in acc)]
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --pmc passes over 'bench.py --steps 20 --warmup 2 %s', per-launch averages by kernel (tools/pmc_passes.py)\n" % " ".join(bench_args))
        for l in log:
            f.write(l + "\n")
        f.write("%-36s" % "counter" + "".join("%13s" % k for k in kern) + "\n")
        for c in sorted(acc):
            f.write("%-36s" % c + "".join("%13.5g" % (sum(acc[c][k]) / len(acc[c][k])) if acc[c][k] else "%13s" % "-" for k in kern) + "\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main()
