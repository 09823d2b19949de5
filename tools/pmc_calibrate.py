#!/usr/bin/env python3
"""Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on the engine's own access shapes.

MI355X_MICROARCH.md calibrates the counters for 16 B-per-lane streaming only (FETCH_SIZE reports
half the bytes there).  The passes of this engine issue 8 B-per-lane plane loads, 32-B record gathers
and 8-B gathers, so the correction is measured here on kernels with a KNOWN byte count
(des_dev_access_bench) instead of being assumed: two rocprofv3 passes (--pmc FETCH_SIZE, --pmc
WRITE_SIZE; they cannot share one) over a child that runs the four patterns, then counter / bytes.
Writes profiles/r02_pmc_calibration.json; tools/measure_traffic.py applies the factors.

  python tools/pmc_calibrate.py          (on the MI355X box)
"""
import collections
import csv
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITEMS = 64 * 1024 * 1024          # per launch: 0.5 - 2 GiB of source data, far beyond L2 + MALL
READ = {0: 16, 1: 8, 2: 36, 3: 12}      # bytes read per item (index + payload), written per item
WRITE = {0: 16, 1: 8, 2: 8, 3: 8}
NAMES = {0: "stream 16 B/lane", 1: "stream 8 B/lane", 2: "gather 32-B records (+4-B index)", 3: "gather 8 B (+4-B index)"}


def child():
    sys.path.insert(0, ROOT)
    import dynearthsol_amd as des
    lib = des.load_hip_lib()
    lib.des_dev_access_bench.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_double)]
    for pat in range(4):
        ms = C.c_double(0)
        rc = lib.des_dev_access_bench(0, pat, ITEMS, 5, C.byref(ms))
        assert rc == 0, rc
        print("pattern %d: %.3f ms per launch = %.0f GB/s" % (pat, ms.value, (READ[pat] + WRITE[pat]) * ITEMS / ms.value / 1e6), flush=True)


def counters(counter):
    d = tempfile.mkdtemp(prefix="pmc_cal_", dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                    sys.executable, os.path.abspath(__file__), "--child"], check=True, stdout=subprocess.DEVNULL, cwd="/tmp",
                   env=dict(os.environ, TMPDIR="/tmp"))
    out = collections.defaultdict(list)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and "k_access<" in r["Kernel_Name"]:
                pat = int(r["Kernel_Name"].split("k_access<")[1].split(">")[0])
                out[pat].append(float(r["Counter_Value"]))
    return out


def main():
    if "--child" in sys.argv:
        return child()
    f, w = counters("FETCH_SIZE"), counters("WRITE_SIZE")
    res = {"items_per_launch": ITEMS, "unit_note": "counter values are KiB (x1024 = bytes)", "patterns": {}}
    for pat in range(4):
        fv = sum(f[pat]) / len(f[pat]) * 1024
        wv = sum(w[pat]) / len(w[pat]) * 1024
        res["patterns"][NAMES[pat]] = {
            "bytes_read": READ[pat] * ITEMS, "bytes_written": WRITE[pat] * ITEMS, "FETCH_SIZE_bytes": fv, "WRITE_SIZE_bytes": wv,
            "fetch_counter_per_byte_read": fv / (READ[pat] * ITEMS), "write_counter_per_byte_written": wv / (WRITE[pat] * ITEMS)}
        print("%-36s FETCH_SIZE / bytes read = %.3f   WRITE_SIZE / bytes written = %.3f"
              % (NAMES[pat], fv / (READ[pat] * ITEMS), wv / (WRITE[pat] * ITEMS)))
    # (on the GPU box only gpurun_out/ travels back: DES_PROFILE_OUT=gpurun_out/<dir>, then copy to profiles/)
    with open(os.path.join(os.environ.get("DES_PROFILE_OUT", os.path.join(ROOT, "profiles")), "r02_pmc_calibration.json"), "w") as fp:
        json.dump(res, fp, indent=1)


if __name__ == "__main__":
    main()
