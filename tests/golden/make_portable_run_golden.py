#!/usr/bin/env python3
"""Writes tests/golden/portable_run_hashes.json: SHA-256 of the state fields after whole runs of
the CPU oracle with the portable libm (des_oracle_set_libm(1)).  With that libm nothing on the
path depends on the platform's C library, so these are known answers for the WHOLE step: the
oracle must reproduce them on any CPU / compiler, the HIP engine (DES_LIBM=portable) on the GPU
(tests/test_known_answers.py).

  python tests/golden/make_portable_run_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import cfgs                                                    # noqa: E402
import dynearthsol_amd as des                                  # noqa: E402
from oracle_binding import OracleEngine, portable_libm         # noqa: E402

FIELDS = ("COORD", "VEL", "TEMPERATURE", "STRESS", "STRAIN", "PLSTRAIN", "VISCOSITY", "DHACC")


def cases():
    g = os.path.join(HERE, "%s.desmesh")
    yield "evp_200", des.Host(cfg_text=cfgs.make(**cfgs.EVP)), 200
    yield "evp_two_materials_100", des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, nmat=2))), 100
    yield "yield_heavy_150", des.Host(cfg_text=cfgs.make(**cfgs.YIELD)), 150
    yield "maxwell_100", des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, rheol="maxwell"))), 100
    yield "oblique_rift_3d_1000", des.Host(cfg_text=cfgs.OBLIQUE, mesh_file=g % "oblique-rift-3d"), 1000
    yield "test_3d_evp_300", des.Host(cfg_text=cfgs.TEST3D, mesh_file=g % "test-3d",
                                      overrides="mat.rheology_type = elasto-visco-plastic\nmat.min_viscosity = 1e19\nbc.mantle_temperature = 1573\n"), 300
    yield "equ_tiny_200", des.Host(cfg_text=cfgs.make_equ()), 200


def digest(engine):
    sc = engine.step(0)
    out = {"steps": int(sc.steps), "time": float(sc.time).hex(), "dt": float(sc.dt).hex()}
    for f in FIELDS:
        out[f] = hashlib.sha256(engine.download(f).tobytes()).hexdigest()
    return out


def main():
    res = {}
    with portable_libm():
        for name, host, nsteps in cases():
            o = OracleEngine(host)
            o.init_from_host(host)
            o.step(nsteps)
            res[name] = digest(o)
            print(name, res[name]["STRESS"][:16])
    with open(os.path.join(HERE, "portable_run_hashes.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
