"""bench.py's output contract at N = 1 (the driver parses this line): one JSON line with the
metric of BASELINE.json, the roofline of the dominant kernel and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

import dynearthsol_amd as des

pytestmark = pytest.mark.gpu


def test_single_gpu_line_has_every_field_of_the_contract():
    out = subprocess.run([sys.executable, os.path.join(des.REPO_ROOT, "bench.py"), "--steps", "20", "--warmup", "3",
                          "--resolution", "2000", "--cpu-steps", "5", "--series-resolution", "2500", "--series-steps", "8"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    base = json.load(open(os.path.join(des.REPO_ROOT, "BASELINE.json")))
    assert r["metric"] == "explicit time-steps/sec x #elements" and r["unit"] == "element-steps/s"
    assert base["metric"].startswith("explicit time-steps/sec")
    assert (r["n_gpus"], r["steps"], r["warmup"]) == (1, 20, 3)
    # the default for N > 1 is strong scaling on the fixed mesh (BASELINE configs[3]); the N = 1 line says so too
    assert r["higher_is_better"] is True and r["scaling"] == "strong" and r["vs_baseline"] is None
    assert r["dtype"] == "f64" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] == pytest.approx(r["config"]["nelem"] * 20 / (r["ms_per_step"] * 20 * 1e-3), rel=1e-6)
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"]) and 0 < roof["frac"] < 1
    assert roof["traffic"] is None and roof["traffic_source"] is None     # PMC traffic is only quoted for the workload it was measured on
    # measured device-copy ceiling: between a third of and the nominal peak
    assert 2500 < roof["ceiling_GBs"] < 8000 and roof["frac_of_ceiling"] == pytest.approx(roof["achieved"] / roof["ceiling_GBs"])
    # ... and what a bytes-only kernel in the dominant launch's memory shape reaches from HBM (des_dev_plane_ceiling): below the
    # plain copy, above what the pass itself reaches
    if roof["kernel"] == "E2G_geom_rotate_update_stress":
        assert 2500 < roof["shape_ceiling_GBs"] <= 1.05 * roof["ceiling_GBs"]
        assert roof["frac_of_shape_ceiling"] == pytest.approx(roof["achieved"] / roof["shape_ceiling_GBs"]) and roof["frac_of_shape_ceiling"] < 1.0
    else:           # (a mesh this small may have another dominant launch: the shape is the stress update's)
        assert roof["shape_ceiling_GBs"] is None
    assert "regular 5-tet mesher" in r["config"]["workload"]
    cpu = r["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "element-steps/s" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    assert r["config"]["nan_entries"] == 0 and r["config"]["status"] == 0
    # the path that was measured: every engine switch set in the environment (none here), both store policies, and
    # the second series on a larger mesh of the same box
    c = r["config"]
    assert c["engine_switches"] == "" and c["interior_step_store_elision"] is True
    assert c["ms_per_step_every_step_stores_every_field"] > 0
    big = c["large_mesh_series"]
    assert big["nelem"] == 160 * 8 * 4 * 5 and big["steps"] == 8 and big["value"] > 0 and big["status"] == 0
    assert "real_traffic_bytes_per_plain_step" not in c           # PMC traffic is only quoted for the workload it was measured on


def test_engine_switches_are_recorded_in_the_line():
    """a stray DES_PATCH=0 changes the measured path: the line must say so (des_dev_config_string)"""
    env = dict(os.environ, DES_PATCH="0", DES_E2_ELIDE="0")
    out = subprocess.run([sys.executable, os.path.join(des.REPO_ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--resolution", "2000",
                          "--cpu-steps", "0", "--no-large-series", "--no-ceiling"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][0])
    sw = r["config"]["engine_switches"].split()
    assert "DES_PATCH=0" in sw and "DES_E2_ELIDE=0" in sw
    assert r["config"]["interior_step_store_elision"] is False and r["config"]["ms_per_step_every_step_stores_every_field"] is None
    assert r["config"]["large_mesh_series"] is None


def test_two_d_line_has_roofline_and_cpu_baseline():
    """bench.py --ndims 2: the tri-mesh line through the same entry point -- roofline of the 2-D engine's dominant launch on
    its own minimum bytes (HIP events inside the engine), CPU baseline from the 2-D oracle's OpenMP build"""
    out = subprocess.run([sys.executable, os.path.join(des.REPO_ROOT, "bench.py"), "--ndims", "2", "--steps", "20", "--warmup", "3",
                          "--resolution", "1000", "--cpu-steps", "5"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    c = r["config"]
    assert r["metric"] == "explicit time-steps/sec x #elements" and r["dtype"] == "f64" and c["ndims"] == 2
    assert c["nelem"] == 2 * 400 * 100 and "triangle" in c["workload"] and c["nan_entries"] == 0 and c["status"] == 0
    roof = r["roofline"]
    assert roof["kernel"] == "K2_stress" and roof["bound"] == "hbm" and 0 < roof["frac"] < 1
    # (168 B per triangle on interior steps -- the pass forms the strain rate itself since round 5 --, 224 on the call's last one,
    #  less the 16 B of moduli a one-material model does not read)
    assert roof["algorithmic_bytes_per_launch"] == pytest.approx(((168 * 19 + 224) / 20 - 16) * c["nelem"] + 48 * c["nnode"])
    assert set(c["kernel_ms_per_call"]) >= {"K2P_temp_dvoldt", "K2_stress", "K2P_force", "K2P_mass"}
    cpu = r["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and "triangle" in cpu["sample"]
    assert c["ms_per_step_every_step_stores_every_field"] > 0 and c["large_mesh_series"] is None


def test_default_workload_is_the_reference_tetgen_mesh():
    """No flags: the headline mesh of SURVEY.md 8(d), named in config.workload."""
    if des.reference_mesh("test-3d-big-460") is None:
        pytest.skip("data/test-3d-big-460.desmesh.xz is missing")
    out = subprocess.run([sys.executable, os.path.join(des.REPO_ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--cpu-steps", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][0])
    assert (r["config"]["nelem"], r["config"]["nnode"]) == (1001310, 185637)
    assert "TetGen mesh at mesh.resolution = 460 m" in r["config"]["workload"] and r["roofline"]["frac"] > 0.3
    assert r["cpu_baseline"] is None and r["config"]["nan_entries"] == 0
    # ... with the large-mesh series of the same box beside it (N = 1 point of the second strong-scaling curve), and what the
    # plain fused step really moves (PMC traffic of its four launches) beside the contract-bytes figure
    assert r["config"]["large_mesh_series"]["nelem"] == 8780800 and r["config"]["large_mesh_series"]["value"] > 2e9
    assert 0.2 < r["config"]["real_traffic_frac_of_hbm_peak"] < r["config"]["whole_step_frac_of_hbm_peak"]
