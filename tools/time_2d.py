#!/usr/bin/env python3
"""Step time of the 2-D (triangle) engine on the regular 2-D mesh: python tools/time_2d.py [resolution_m ...]
(400 km x 100 km box, elasto-visco-plastic, thermal + NMD + surface diffusion)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cfgs
import dynearthsol_amd as des

for res in [float(a) for a in sys.argv[1:]] or [1000.0, 500.0, 250.0]:
    host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, lx=400e3, lz=100e3, res=res)), ndims=2)
    dev = des.DeviceEngine(host)
    dev.init_from_host(host)
    nst = int(os.environ.get("DES2D_TIME_STEPS", "400"))
    dev.step(40)
    dev.timer_start()
    dev.step(nst, want_scalars=False)
    ms = dev.timer_stop() / nst
    print("%8.0f m  %8d triangles  %.4f ms/step  %.3e element-steps/s  %.0f GB/s of the engine's own %d B/elem + %d B/node"
          % (res, host.nelem, ms, host.nelem / (ms * 1e-3), dev._lib.des_dev_algorithmic_bytes_per_step(dev._h) / (ms * 1e-3) / 1e9, 8 * 118, 8 * 60))
    dev.close()
