#!/usr/bin/env python3
"""The device timeline of a few steps of the 2-D middle shard with its exchange on RCCL (tools/time_2d_shard.py --only ...),
in-order and overlapped: every dispatch of two steady-state steps with its queue, start (us from the first) and duration,
from a rocprofv3 kernel trace.  Shows what runs beside what, and what a cross-stream dependency costs.

    python tools/timeline_2d_overlap.py > profiles/r04_x_2d_overlap_timeline.txt        (on the MI355X box)
"""
import csv
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"des2d::|des_hip::|desk::", "", name)
    return name.split("(")[0][:52]


def run(mode):
    base = os.path.join(ROOT, "gpurun_out")
    d = tempfile.mkdtemp(prefix="tl2_", dir=base if os.path.isdir(base) else None)
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.join(ROOT, "tools", "time_2d_shard.py"), "60", "--only", mode]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    if out.returncode:
        sys.exit("rocprofv3 failed:\n" + out.stderr[-3000:])
    rows = []
    for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    rows.sort()
    line = [l for l in out.stdout.splitlines() if l.startswith(mode)]
    print("== %s   (%s under the profiler)" % (mode, line[0] if line else "?"))
    # two steps out of the middle: from one K2P_force to the third one after it
    idx = [i for i, r in enumerate(rows) if r[2].startswith("k2p_force")]
    a = idx[len(idx) // 2]
    b = idx[len(idx) // 2 + 2]
    t0 = rows[a][0]
    queues = sorted({r[3] for r in rows[a:b]})
    print("%-52s %6s %10s %9s" % ("dispatch", "queue", "start us", "us"))
    for s, e, k, q in rows[a:b]:
        print("%-52s %6s %10.2f %9.2f" % (k, queues.index(q), (s - t0) / 1e3, (e - s) / 1e3))
    print("two steps: %.1f us" % ((rows[b][0] - t0) / 1e3))


if __name__ == "__main__":
    for mode in ("inorder", "overlapped"):
        run(mode)
