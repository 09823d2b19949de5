#!/usr/bin/env python3
"""Regenerate tests/golden/test-tiny.desmesh and test-topo.desmesh: the meshes the reference's 2-D build
makes for benchmarks-cores/test-tiny.cfg (meshing_option = 91, benchmarks-cores/cube.poly) and
benchmarks-cores/test-topo.cfg (topo.poly: a box with a 10-km relief on top, mesh.resolution = 5e3).  The reference's
vendored Triangle is compiled from /root/reference/triangle where it lies (`make -C oracle ref` ->
oracle/_ref/trimesh, driver oracle/ref_trimesh/trimesh_driver.cpp restating new_mesh_from_polyfile's
input and triangulate_polygon's switches, mesh.cxx:1872-2251, 688-770); the host library then
discards internal segments and renumbers as create_new_mesh does (mesh.cxx:3499-3502) and writes the
finished mesh: 97 nodes / 164 triangles; 361 / 653 for test-topo.  Dev-time tool (needs /root/reference)."""
import os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfgs
import dynearthsol_amd as des

subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
raw = os.path.join(tempfile.mkdtemp(), "raw.desmesh")
# meshing_option, mesh.resolution, mesh.min_angle, mat.num_materials of the .cfg
subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "trimesh"), "--poly",
                       "/root/reference/benchmarks-cores/cube.poly", "91", "1e4", "30", "8", raw])
h = des.Host(cfg_text=cfgs.TEST_TINY, mesh_file=raw, ndims=2)
assert (h.nnode, h.nelem) == (97, 164)
h.save_mesh(os.path.join(HERE, "test-tiny.desmesh"))
print("wrote test-tiny.desmesh:", h.nnode, "nodes", h.nelem, "triangles")

raw = os.path.join(tempfile.mkdtemp(), "raw.desmesh")
subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "trimesh"), "--poly",
                       "/root/reference/benchmarks-cores/topo.poly", "91", "5e3", "30", "8", raw])
h = des.Host(cfg_text=cfgs.TEST_TINY, overrides=cfgs.TEST_TOPO_OVERRIDES, mesh_file=raw, ndims=2)
assert (h.nnode, h.nelem) == (361, 653)
h.save_mesh(os.path.join(HERE, "test-topo.desmesh"))
print("wrote test-topo.desmesh:", h.nnode, "nodes", h.nelem, "triangles")
