#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes: summarize_counters.py <dir> [<dir> ...]"""
import collections, csv, glob, sys
NAMES = {"E1_geom_rotate_strainrate<3>": "E1", "N1_mass_temperature_dvoldt<1, 1>": "N1", "E2_update_stress<desk::MathOcml, 1>": "E2", "E2_update_stress<desk::MathOcml, 0>": "E2 (one pass)", "E2_return_mapping<desk::MathOcml>": "E2R",
         "N2_nmd_gather": "N2", "E3_nmd_force": "E3", "N3_force_velocity_coord": "N3", "k_s2": "S2", "k_s3_finalize": "S3"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if k.startswith("des_hip::"):
                k = k[len("des_hip::"):]
            if k in NAMES:
                acc[r["Counter_Name"]][NAMES[k]].append(float(r["Counter_Value"]))
kern = ["E1", "N1", "E2", "N2", "E3", "N3", "S2", "S3"]
print("%-32s" % "counter" + "".join("%12s" % k for k in kern))
for c in sorted(acc):
    print("%-32s" % c + "".join("%12.4g" % (sum(acc[c][k]) / len(acc[c][k])) if acc[c][k] else "%12s" % "-" for k in kern))
