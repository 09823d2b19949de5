#!/usr/bin/env python3
"""Regenerate tests/golden/conjugate-faults-3d.desmesh: the mesh the reference builds for
examples/conjugate-faults-3d.cfg (meshing_option = 1: new_mesh_uniform_resolution) with its
vendored TetGen (`make -C oracle ref` -> oracle/_ref/tetmesh --uniform), finished by the host
library as create_new_mesh does.  4,313 nodes / 20,334 tets.  Dev-time tool (needs /root/reference)."""
import os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfgs
import dynearthsol_amd as des

subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
raw = os.path.join(tempfile.mkdtemp(), "raw.desmesh")
subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "tetmesh"), "--uniform", "200e3", "100e3", "50e3", "5e3", raw])
h = des.Host(cfg_text=cfgs.CONJUGATE, mesh_file=raw)
assert (h.nnode, h.nelem) == (4313, 20334)
h.save_mesh(os.path.join(HERE, "conjugate-faults-3d.desmesh"))
print("wrote conjugate-faults-3d.desmesh:", h.nnode, "nodes", h.nelem, "tets")
