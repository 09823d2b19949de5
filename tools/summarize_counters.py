#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes: summarize_counters.py <dir> [<dir> ...]"""
import collections, csv, glob, sys
def short(name):
    k = name.split("(")[0].replace("void ", "").strip()
    for ns in ("des_hip::", "(anonymous namespace)::"):
        if k.startswith(ns):
            k = k[len(ns):]
    return k.split("<")[0]


NAMES = {"E1_geom_rotate_strainrate": "E1", "N1_mass_temperature_dvoldt": "N1", "EN1_mass_temperature_dvoldt": "EN1",
         "E2_update_stress": "E2", "E2_return_mapping": "E2R", "N2_nmd_gather": "N2", "EN2_nmd_gather": "EN2",
         "E3_nmd_force": "E3", "N3_force_velocity_coord": "N3", "EN3_force_nodes": "EN3", "k_s2": "S2", "k_s3_finalize": "S3"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k in NAMES:
                acc[r["Counter_Name"]][NAMES[k]].append(float(r["Counter_Value"]))
kern = [k for k in ("E1", "N1", "EN1", "E2", "N2", "EN2", "E3", "N3", "EN3", "S2", "S3") if any(acc[c][k] for c in acc)]
print("%-32s" % "counter" + "".join("%12s" % k for k in kern))
for c in sorted(acc):
    print("%-32s" % c + "".join("%12.4g" % (sum(acc[c][k]) / len(acc[c][k])) if acc[c][k] else "%12s" % "-" for k in kern))
