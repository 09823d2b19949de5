import json, subprocess, sys
for rh in ["elastic", "maxwell", "elasto-plastic", "elasto-visco-plastic"]:
    out = subprocess.check_output([sys.executable, "bench.py", "--steps", "60", "--warmup", "10", "--cpu-steps", "0", "--no-large-series", "--no-elide-compare", "--rheology", rh])
    r = json.loads(out.decode().strip().splitlines()[-1])
    k = r["config"]["kernel_ms_per_call"]
    print("%-22s ms/step %.4f  E2 %.1f us  E1 %.1f  E3 %.1f" % (rh, r["ms_per_step"], 1e3*k["E2_update_stress"], 1e3*k["E1_geom_rotate_strainrate"], 1e3*k["E3_nmd_force"]))
