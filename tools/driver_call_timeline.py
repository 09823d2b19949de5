#!/usr/bin/env python3
"""What the driver's `bench.py --steps 20 --warmup 5` call is made of on the device: every dispatch of the TIMED 20-step call
(the last call of the run), from a rocprofv3 kernel trace -- per kernel its launches, total time and the gaps in front of it,
and the call's steps one by one (time from one force pass to the next).  Shows what the first / last / compute_dt steps of a
call cost beside the plain ones.

    python tools/driver_call_timeline.py [steps warmup] > profiles/r04_x_driver_call_timeline.txt      (on the MI355X box)
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


def main():
    steps = sys.argv[1] if len(sys.argv) > 1 else "20"
    warm = sys.argv[2] if len(sys.argv) > 2 else "5"
    base = os.path.join(ROOT, "gpurun_out")
    d = tempfile.mkdtemp(prefix="dc_", dir=base if os.path.isdir(base) else None)
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--steps", steps, "--warmup", warm, "--cpu-steps", "0", "--no-large-series", "--no-elide-compare", "--no-profile", "--no-ceiling"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    if out.returncode:
        sys.exit("rocprofv3 failed:\n" + out.stderr[-2000:])
    rows = []
    for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    force = [i for i, r in enumerate(rows) if r[2].startswith("EN3_force_nodes") or r[2].startswith("N3_")]
    n = int(steps)
    first = force[-n]
    # the call starts behind the last dispatch of the call before: walk back from its first force pass to the largest gap
    lo = first
    while lo > 0 and rows[lo][0] - rows[lo - 1][1] < 20000 and first - lo < 12:
        lo -= 1
    win = rows[lo:]
    span = (win[-1][1] - win[0][0]) / 1e3
    print("# %s" % " ".join(cmd[8:]))
    print("# timed call: %d dispatches, %.1f us on the device = %.2f us per step" % (len(win), span, span / n))
    dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
    prev = None
    for s, e, k in win:
        dur[k] += e - s
        cnt[k] += 1
        if prev is not None: gap[k] += s - prev
        prev = e
    print("%-40s %6s %10s %10s %10s" % ("kernel", "calls", "us/call", "total us", "gaps us"))
    for k in sorted(cnt, key=lambda k: -(dur[k] + gap[k])):
        print("%-40s %6d %10.2f %10.1f %10.1f" % (k, cnt[k], dur[k] / cnt[k] / 1e3, dur[k] / 1e3, gap[k] / 1e3))
    print("step by step (from the end of one force pass to the end of the next):")
    ends = [win[0][0]] + [rows[i][1] for i in force[-n:]] + [win[-1][1]]
    for j in range(1, len(ends)):
        a, b = ends[j - 1], ends[j]
        ks = [r[2].split("<")[0] for r in win if r[0] >= a and r[1] <= b]
        label = "step %2d" % j if j <= n else "tail   "
        print("  %s %8.1f us   %s" % (label, (b - a) / 1e3, " ".join(ks)))
    for l in out.stdout.splitlines():
        if l.startswith("{"):
            import json
            print("# bench line under the profiler: %.4f ms per step" % json.loads(l)["ms_per_step"])


if __name__ == "__main__":
    main()
