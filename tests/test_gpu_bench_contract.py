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
                          "--resolution", "2000", "--cpu-steps", "5"], capture_output=True, text=True, timeout=600)
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
    assert "regular 5-tet mesher" in r["config"]["workload"]
    cpu = r["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "element-steps/s" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    assert r["config"]["nan_entries"] == 0 and r["config"]["status"] == 0


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
