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
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["dtype"] == "f64" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] == pytest.approx(r["config"]["nelem"] * 20 / (r["ms_per_step"] * 20 * 1e-3), rel=1e-6)
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"]) and 0 < roof["frac"] < 1
    assert roof["traffic"] is None                       # PMC traffic is only quoted for the default workload
    cpu = r["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "element-steps/s" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    assert r["config"]["nan_entries"] == 0 and r["config"]["status"] == 0
