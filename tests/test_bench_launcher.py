"""`python bench.py --gpus N` starts its own ranks (bench.py: launch_ranks): the plain command of the bench contract must
work for N > 1 without a launcher around it.  Here, without a GPU: the launcher spawns torch.distributed.run as a child,
both ranks meet over gloo, the device path refuses loudly (no CPU fallback), and the launcher hands the failure on --
non-zero exit code, no result line.  The same command WITH a GPU prints its line: tests/test_gpu_decomp.py."""
import os
import subprocess
import sys

import dynearthsol_amd as des


def _device_count():
    import torch
    return torch.cuda.device_count()


def test_plain_command_starts_its_own_ranks_and_fails_loudly_without_a_gpu():
    if _device_count() > 0:
        import pytest
        pytest.skip("a GPU is visible: the GPU suite runs the same command to a result line")
    env = dict(os.environ, DES_BENCH_BACKEND="gloo", DES_BENCH_TRANSPORT="host")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(des.REPO_ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                          "--no-large-series", "--resolution", "4000", "--cpu-steps", "0"],
                         capture_output=True, text=True, timeout=280, env=env, cwd=des.REPO_ROOT)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], "a result line without a device"
    # both ranks were started and got as far as the engine: the refusal is the device path's own, once per rank
    assert "[rank0]" in out.stderr and "[rank1]" in out.stderr
    assert out.stderr.count("the device path has no CPU fallback") >= 2


def test_rank_count_and_gpus_flag_must_agree():
    """under a launcher (WORLD_SIZE set) --gpus has to name the same count: no silent N = 1 line for an N = 4 command"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(des.REPO_ROOT, "bench.py"), "--gpus", "4", "--steps", "2"],
                         capture_output=True, text=True, timeout=120, env=env, cwd=des.REPO_ROOT)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr and not out.stdout.strip()


def test_launcher_keeps_the_refusal_codes(tmp_path):
    """torch.distributed.run exits 1 whatever its ranks returned; the ranks leave 3 (no RCCL communicator) / 4 (self-check
    failed) in DES_BENCH_RC_FILE and launch_ranks returns that"""
    sys.path.insert(0, des.REPO_ROOT)
    import bench
    rc_file = tmp_path / "rc"
    os.environ["DES_BENCH_RC_FILE"] = str(rc_file)
    try:
        bench._set_exit_code(4)
    finally:
        del os.environ["DES_BENCH_RC_FILE"]
    assert rc_file.read_text().strip() == "4"
