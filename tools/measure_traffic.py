#!/usr/bin/env python3
"""HBM traffic per launch of every pass from the rocprofv3 PMC counters, for bench.py's
`roofline.traffic`: two separate passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`: they do not fit
one pass, MI355X_MICROARCH.md) over `python3 bench.py --steps 20 --warmup 2 --cpu-steps 0
--no-profile --no-ceiling [bench args]`, averaged per kernel.

Counter -> bytes: both counters are in KiB.  WRITE_SIZE counts the bytes exactly; FETCH_SIZE counts
HALF of the bytes moved, for 8 B per lane streams as for the 16 B per lane ones the guide documents
(tools/pmc_calibrate.py, profiles/r02_pmc_calibration.json: 0.500 for both; a random 32-B or 8-B
gather shows 128 B moved per access, i.e. whole lines, in the same unit) -- so
traffic = (FETCH_SIZE / 0.5 + WRITE_SIZE) x 1024 for every pass of this engine.

  python tools/measure_traffic.py [bench args]      (on the MI355X box)   -> profiles/r05_pmc_traffic.json (with --ndims 2: r05_pmc_traffic_2d.json)
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """'void des_hip::E2_update_stress<desk::MathOcml, 1>(args)' -> 'E2_update_stress'"""
    k = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
    for ns in ("des_hip::", "des2d::", "(anonymous namespace)::"):
        if k.startswith(ns):
            k = k[len(ns):]
    base = k.split("<")[0]
    # the 2-D engine's launches under the names its own accounting (and bench.py --ndims 2) uses
    two_d = {"k2_stress": "K2_stress", "k2p_temp_dvoldt": "K2P_temp_dvoldt", "k2p_force": "K2P_force", "k2p_mass": "K2P_mass",
             "k2_node_avg": "K2_node_avg", "k2_node_avg_extent": "K2_node_avg", "k2_rotate_vol": "K2_rotate_vol",
             "k2_surf_commit": "K2_surf_commit", "k2_surf_edv_cse_all": "K2_surf_edv_cse_all"}
    if base in two_d:
        return two_d[base]
    if base == "E2_update_stress_pipe":
        return "E2G_geom_rotate_update_stress"          # the pipelined form of E2<GEO> (round 4): the same pass, the same profile id
    if base == "E2_update_stress":
        targs = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]      # <M, DEFER, GEO, RH>
        if len(targs) >= 3 and targs[2] == "1":
            return "E2G_geom_rotate_update_stress"      # the GEO variant: bench.py's name for it (profile id K_E2G)
    return base


def counters(counter, bench_args):
    base = os.path.join(ROOT, "gpurun_out")
    d = tempfile.mkdtemp(prefix="pmc_", dir=base if os.path.isdir(base) else None)
    out = subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                          sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2", "--cpu-steps", "0",
                          "--no-profile", "--no-ceiling", "--no-large-series", "--no-elide-compare"] + bench_args, capture_output=True, text=True, cwd="/tmp",
                         env=dict(os.environ, TMPDIR="/tmp"))
    if out.returncode:
        sys.exit("rocprofv3 failed:\n" + out.stderr[-2000:])
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    vals = collections.defaultdict(list)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return line, vals


def main():
    bench_args = sys.argv[1:]
    cal = {"fetch": 0.5, "write": 1.0, "source": "MI355X_MICROARCH.md (16 B/lane streaming)"}
    cpath = os.path.join(ROOT, "profiles", "r02_pmc_calibration.json")
    if os.path.exists(cpath):
        # the streaming rows give the unit of the counter (every 128-B request is tallied as 64 B); the
        # gather rows show the same unit on whole-line fetches (a random 32-B or 8-B read moves 128 B)
        pats = {k: v for k, v in json.load(open(cpath))["patterns"].items() if k.startswith("stream")}
        fs = [v["fetch_counter_per_byte_read"] for v in pats.values()]
        ws = [v["write_counter_per_byte_written"] for v in pats.values()]
        cal = {"fetch": sum(fs) / len(fs), "write": sum(ws) / len(ws), "fetch_range": [min(fs), max(fs)],
               "write_range": [min(ws), max(ws)], "source": "profiles/r02_pmc_calibration.json (streaming rows)"}
    line, f = counters("FETCH_SIZE", bench_args)
    _, w = counters("WRITE_SIZE", bench_args)
    res = {"workload": {"nelem": line["config"]["nelem"], "nnode": line["config"]["nnode"]},
           "workload_text": line["config"]["workload"], "bench_args": bench_args, "calibration": cal}
    for k in sorted(f):
        if k not in w or not k[:2] in ("E1", "E2", "E3", "N1", "N2", "N3", "EN", "S2", "S3", "K2"):
            continue
        fv, wv = sum(f[k]) / len(f[k]), sum(w[k]) / len(w[k])
        res[k] = {"fetch_size_kib_raw": fv, "write_size_kib_raw": wv, "launches": len(f[k]),
                  "traffic_bytes_per_launch": (fv / cal["fetch"] + wv / cal["write"]) * 1024}
        print("%-28s FETCH %.1f MiB raw, WRITE %.1f MiB -> traffic %.1f MB/launch" % (k, fv / 1024, wv / 1024, res[k]["traffic_bytes_per_launch"] / 1e6))
    # (on the GPU box only gpurun_out/ travels back: DES_PROFILE_OUT=gpurun_out/<dir>, then copy to profiles/)
    json.dump(res, open(os.path.join(os.environ.get("DES_PROFILE_OUT", os.path.join(ROOT, "profiles")), os.environ.get("DES_TRAFFIC_NAME", "r05_pmc_traffic_2d.json" if "--ndims" in bench_args else "r05_pmc_traffic.json")), "w"), indent=1)


if __name__ == "__main__":
    main()
