#!/usr/bin/env python3
"""Regenerate tests/golden/test-3d.desmesh: the mesh the reference builds for
benchmarks-cores/test-3d.cfg (meshing_option = 2).  The reference's vendored TetGen is compiled
from /root/reference/tetgen where it lies (`make -C oracle ref` -> oracle/_ref/tetmesh, driver
oracle/ref_tetmesh/tetmesh_driver.cpp restating new_mesh_refined_zone's input, mesh.cxx:1642-1845);
the host library then discards internal segments and renumbers as create_new_mesh does
(mesh.cxx:3499-3502) and writes the finished mesh.  Result: 3,018 nodes / 13,850 tets, the
counts SURVEY.md Appendix A records for the reference.  Dev-time tool (needs /root/reference)."""
import os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfgs
import dynearthsol_amd as des

subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
raw = os.path.join(tempfile.mkdtemp(), "raw.desmesh")
subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "tetmesh"), "100e3", "10e3", "10e3", "1e3", "10",
                       "0.3", "0.7", "0.0", "1.0", "0.0", "1.0", raw])
h = des.Host(cfg_text=cfgs.TEST3D, mesh_file=raw)
assert (h.nnode, h.nelem) == (3018, 13850)
h.save_mesh(os.path.join(HERE, "test-3d.desmesh"))
print("wrote test-3d.desmesh:", h.nnode, "nodes", h.nelem, "tets")
