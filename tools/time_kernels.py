#!/usr/bin/env python3
"""Per-kernel HIP-event times of the headline step WITHOUT looking at the results (no status / NaN check): for A/B
builds whose results are wrong on purpose (timing experiments: -DDES_EXP_*), selected with DES_HIP_LIB.

    DES_HIP_LIB=build/variants/x.so python tools/time_kernels.py [--resolution R]      (on the MI355X box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402
import dynearthsol_amd as des                          # noqa: E402


def main():
    res = None
    if "--resolution" in sys.argv:
        res = float(sys.argv[sys.argv.index("--resolution") + 1])
    if res is None:
        host = des.Host(cfg_text=bench.BENCH_CFG.format(res="460.0", xlen="400e3"),
                        overrides="mesh.meshing_option = 2\nmesh.meshing_elem_shape = 0\n", mesh_file=des.reference_mesh("test-3d-big-460"))
    else:
        host = des.Host(cfg_text=bench.BENCH_CFG.format(res=repr(res), xlen="400e3"))
    dev = des.DeviceEngine(host)
    dev.init_from_host(host)
    dev.step(20, want_scalars=False)
    dev.sync()
    dev.timer_start()
    dev.step(200, want_scalars=False)
    ms = dev.timer_stop() / 200
    dev.profile_enable(True)
    dev.step(20, want_scalars=False)
    k = {n: 1e3 * t / c for n, t, c in dev.profile_read()}
    print("%s: %.1f us per step; %s" % (os.path.basename(os.environ.get("DES_HIP_LIB", "default")), 1e3 * ms,
                                        "  ".join("%s %.1f" % (n.split("_")[0], k[n]) for n in sorted(k) if n[:2] in ("EN", "E2"))))


if __name__ == "__main__":
    main()
