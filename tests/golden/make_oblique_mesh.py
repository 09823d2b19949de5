#!/usr/bin/env python3
"""Regenerate tests/golden/oblique-rift-3d.desmesh: the mesh the reference builds for
examples/oblique-rift-3d.cfg (meshing_option = 2) with its vendored TetGen (`make -C oracle ref`
-> oracle/_ref/tetmesh), finished by the host library as create_new_mesh does.  676 nodes /
2,991 tets, the counts SURVEY.md 8d records.  Dev-time tool (needs /root/reference)."""
import os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfgs
import dynearthsol_amd as des

subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
raw = os.path.join(tempfile.mkdtemp(), "raw.desmesh")
subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "tetmesh"), "200e3", "100e3", "50e3", "5e3", "30",
                       "0.3", "0.7", "0.3", "0.7", "0.7", "1.0", raw])
h = des.Host(cfg_text=cfgs.OBLIQUE, mesh_file=raw)
assert (h.nnode, h.nelem) == (676, 2991)
h.save_mesh(os.path.join(HERE, "oblique-rift-3d.desmesh"))
print("wrote oblique-rift-3d.desmesh:", h.nnode, "nodes", h.nelem, "tets")
