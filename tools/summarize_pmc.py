#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes (separate FETCH_SIZE and WRITE_SIZE runs, as
MI355X_MICROARCH.md prescribes) into profiles/pmc_traffic.json: HBM-side bytes per launch of
each pass, with the gfx950 correction (FETCH_SIZE counts 64 B per 128-B request: x2; both
counters are in KiB).  usage: summarize_pmc.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, sys

NAMES = {"E1_geom_rotate_strainrate<3>": "E1_geom_rotate_strainrate",
         "N1_mass_temperature_dvoldt<1, 1>": "N1_mass_temperature_dvoldt",
         "E2_update_stress<desk::MathOcml, 1>": "E2_update_stress", "E2_return_mapping<desk::MathOcml>": "E2_return_mapping",
         "N2_nmd_gather": "N2_nmd_gather",
         "E3_nmd_force": "E3_nmd_force", "N3_force_velocity_coord": "N3_force_velocity_coord"}


def short(name):
    """'void des_hip::E2_update_stress<desk::MathOcml, 1>(args)' -> 'E2_update_stress<desk::MathOcml, 1>'"""
    k = name.split("(")[0].replace("void ", "").strip()
    for ns in ("des_hip::", "(anonymous namespace)::"):
        if k.startswith(ns):
            k = k[len(ns):]
    return k


def agg(d, counter):
    out = collections.defaultdict(list)
    for path in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            out[k].append(float(r["Counter_Value"]))
    return out


f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
res = {}
for k, name in NAMES.items():
    if k in f and k in w:
        fv, wv = sum(f[k]) / len(f[k]), sum(w[k]) / len(w[k])
        res[name] = {"fetch_size_kib_raw": fv, "write_size_kib": wv, "launches": len(f[k]),
                     "traffic_bytes_per_launch": (2 * fv + wv) * 1024}
json.dump(res, open(sys.argv[3], "w"), indent=1)
for k, v in res.items():
    print("%-28s traffic %.1f MB/launch" % (k, v["traffic_bytes_per_launch"] / 1e6))
