#!/usr/bin/env python3
"""Where a step's time goes on the device: per kernel the mean duration AND the mean gap to the
kernel before it (end -> start), from a rocprofv3 kernel trace of bench.py.  The gaps are what a kernel
boundary costs in this stream (launch + cache write-back / invalidate); durations + gaps = the step.

  python tools/step_timeline.py [bench args] > profiles/r03_x_timeline_137k.txt        (on the MI355X box)

Only the dispatches of the timed region's steady state are used: the last `--steps` x (launches per step) of
the trace would include the end-of-call classic step, so the middle half of the trace is taken."""
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
    bench_args = sys.argv[1:]
    base = os.path.join(ROOT, "gpurun_out")
    d = tempfile.mkdtemp(prefix="tl_", dir=base if os.path.isdir(base) else None)
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--cpu-steps", "0", "--no-large-series", "--no-elide-compare", "--no-profile", "--no-ceiling"] + bench_args
    out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    if out.returncode:
        sys.exit("rocprofv3 failed:\n" + out.stderr[-2000:])
    rows = []
    for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    n = len(rows)
    mid = rows[n // 4: 3 * n // 4]
    dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
    prev_end = None
    for s, e, k in mid:
        if prev_end is not None:
            dur[k] += e - s
            gap[k] += s - prev_end
            cnt[k] += 1
        prev_end = e
    span = mid[-1][1] - mid[0][1]
    steps = cnt.get("EN3_force_nodes", 0) or cnt.get("N3_force_velocity_coord", 1)
    print("# %s" % " ".join(cmd[8:]))
    print("# %d dispatches in the window, %d steps, %.2f us per step (window span / steps)" % (len(mid), steps, span / steps / 1e3))
    print("%-36s %8s %10s %10s %12s" % ("kernel", "calls", "us/call", "gap us", "us per step"))
    tot_d = tot_g = 0.0
    for k in sorted(cnt, key=lambda k: -(dur[k] + gap[k])):
        print("%-36s %8d %10.2f %10.2f %12.2f" % (k, cnt[k], dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3, (dur[k] + gap[k]) / steps / 1e3))
        tot_d += dur[k]; tot_g += gap[k]
    print("%-36s %8s %10.2f %10.2f %12.2f" % ("sum per step", "", tot_d / steps / 1e3, tot_g / steps / 1e3, (tot_d + tot_g) / steps / 1e3))
    for l in out.stdout.splitlines():
        if l.startswith("{"):
            import json
            print("# bench line under the profiler: %.4f ms per step" % json.loads(l)["ms_per_step"])


if __name__ == "__main__":
    main()
