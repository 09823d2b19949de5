#!/usr/bin/env python3
"""time des_dev_compute_dt (E1<DT> + finalize) for variant libraries"""
import os, subprocess, sys, json
code = '''
import sys, time
sys.path.insert(0, ".")
import bench, dynearthsol_amd as des
host = des.Host(cfg_text=bench.BENCH_CFG.format(res=repr(400e3/560), xlen=repr(400e3)))
dev = des.DeviceEngine(host); dev.init_from_host(host)
for _ in range(5): dev.compute_dt()
dev.sync(); t = time.perf_counter()
for _ in range(50): dev.compute_dt()
dev.sync(); print("%.1f us per compute_dt" % ((time.perf_counter() - t) / 50 * 1e6))
'''
for lib in sys.argv[1:]:
    out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, DES_HIP_LIB=os.path.abspath(lib)))
    print(os.path.basename(lib), out.decode().strip().splitlines()[-1])
