"""The drop-in executable and des_run over the HIP engine: frames of a device run against frames
of the same loop over the CPU oracle (bit for bit for elasto-plastic physics)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from dynearthsol_amd import driver
from oracle_binding import OracleEngine
from test_driver_output import oracle_api, read_frame, in_tmp  # noqa: F401

pytestmark = pytest.mark.gpu

EXE = os.path.join(des.REPO_ROOT, "dynearthsol_amd", "bin", "dynearthsol3d-hip")
OV = ("sim.max_steps = 40\nsim.output_step_interval = 20\nsim.is_outputting_averaged_fields = yes\n"
      "mesh.quality_check_step_interval = 10\nsim.checkpoint_frame_interval = 2\n")


def test_executable_writes_the_frames_the_oracle_loop_writes(in_tmp):
    # overrides go into the file: the executable takes the reference's single argument
    with open("model.cfg", "w") as f:
        f.write(cfgs.apply_overrides(cfgs.make(**cfgs.EP), OV + "sim.modelname = gpu\n"))
    out = subprocess.run([EXE, "model.cfg"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Output # 2" in out.stdout and "Ending simulation." in out.stdout
    host = des.Host(cfg_path="model.cfg", overrides="sim.modelname = cpu\n")
    st = driver.run(host, api=oracle_api())
    assert st.frames == 3 and st.checkpoints == 2
    for frame in (0, 1, 2):
        a, b = read_frame("gpu.save.%06d" % frame), read_frame("cpu.save.%06d" % frame)
        assert sorted(a) == sorted(b)
        for name in a:
            if name == "walltime_sec":
                continue
            assert np.array_equal(a[name], b[name]), (frame, name)
    a, b = read_frame("gpu.chkpt.000002"), read_frame("cpu.chkpt.000002")
    for name in a:
        assert np.array_equal(a[name], b[name]), name
    ia, ib = np.loadtxt("gpu.info").reshape(-1, 8), np.loadtxt("cpu.info").reshape(-1, 8)
    assert np.array_equal(np.delete(ia, 4, axis=1), np.delete(ib, 4, axis=1))


def test_python_run_binds_the_hip_engine(in_tmp):
    host = des.Host(cfg_text=cfgs.make(**cfgs.EVP), overrides=OV + "sim.modelname = py\n")
    st = driver.run(host)
    assert (st.steps, st.frames, st.exit_code) == (40, 3, 0) and st.compute_seconds > 0
    fr = read_frame("py.save.000002")
    assert fr["steps"].view(np.int32)[0] == 40


def test_mesh_quality_reductions_match_the_oracle():
    # elasto-plastic below yield: device and oracle states are bit-identical, so must be the reductions
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP))
    dev, ora = des.DeviceEngine(host), OracleEngine(host)
    dev.init_from_host(host); ora.init_from_host(host)
    dev.step(30); ora.step(30)
    res = []
    for eng, prefix in ((dev, "des_dev"), (ora, "des_oracle")):
        q = driver.DesQuality()
        f = getattr(eng._lib, prefix + "_mesh_quality")
        f.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.POINTER(driver.DesQuality)]
        vol = eng.download("VOLUME")
        zmin = -host.params.zlength
        # thresholds chosen so that every branch reports something
        assert f(eng._h, float(vol.mean()), zmin, 1e-3, C.byref(q)) == 0
        res.append((q.small_elem, q.bottom_node, q.worst_elem, q.worst_quality))
    assert res[0] == res[1]
    assert res[0][0] >= 0 and res[0][2] >= 0 and 0 < res[0][3] < 1


def test_executable_restarts_from_its_own_checkpoint(in_tmp):
    base = ("sim.max_steps = 40\nsim.output_step_interval = 20\nsim.checkpoint_frame_interval = 1\n"
            "mesh.quality_check_step_interval = 10\nsim.is_outputting_averaged_fields = yes\n")
    kw = dict(cfgs.EVP, nmat=2)
    open("a.cfg", "w").write(cfgs.apply_overrides(cfgs.make(**kw), base + "sim.modelname = a\n"))
    open("b.cfg", "w").write(cfgs.apply_overrides(cfgs.make(**kw), base + "sim.modelname = b\nsim.is_restarting = yes\n"
                                                  "sim.restarting_from_modelname = a\nsim.restarting_from_frame = 1\n"))
    for cfg in ("a.cfg", "b.cfg"):
        out = subprocess.run([EXE, cfg, "--quiet"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
    a, b = read_frame("a.save.000002"), read_frame("b.save.000002")
    for name in a:
        if name != "walltime_sec":
            assert np.array_equal(a[name], b[name]), name


def test_distributed_run_with_one_rank_equals_the_plain_run(in_tmp):
    """dynearthsol_amd.distributed on a world of one (all the 1-GPU box can hold): the collective
    engine table over the HIP engine, RCCL communicator attached, writes the frames of driver.run."""
    import torch.distributed as dist
    from dynearthsol_amd.distributed import run_distributed
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 90))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        kw = dict(cfgs.EVP, nmat=2)
        st = run_distributed(des.Host(cfg_text=cfgs.make(**kw), overrides=OV + "sim.modelname = dist\n"), dist)
        assert (st.steps, st.frames, st.exit_code) == (40, 3, 0)
    finally:
        dist.destroy_process_group()
    driver.run(des.Host(cfg_text=cfgs.make(**kw), overrides=OV + "sim.modelname = plain\n"))
    for frame in (0, 1, 2):
        a, b = read_frame("dist.save.%06d" % frame), read_frame("plain.save.%06d" % frame)
        for name in a:
            if name != "walltime_sec":
                assert np.array_equal(a[name], b[name]), (frame, name)


def test_reference_tiny_3d_benchmark_through_the_executable(in_tmp):
    """benchmarks-cores/test-3d-tiny.cfg (= test-3d.cfg with 4 steps, a frame every step, a
    quality check every 2nd) on the reference's TetGen mesh, through `dynearthsol3d-hip cfg --mesh`:
    every frame and checkpoint equals the oracle loop's to the bit."""
    mesh = os.path.join(des.REPO_ROOT, "tests", "golden", "test-3d.desmesh")
    tiny = ("sim.max_steps = 4\nsim.output_step_interval = 1\nsim.checkpoint_frame_interval = 2\n"
            "mesh.quality_check_step_interval = 2\n")
    text = cfgs.apply_overrides(cfgs.TEST3D, tiny + "sim.modelname = gpu\n")
    text = "\n".join(l for l in text.splitlines() if not l.startswith(("max_time_in_yr", "output_time_interval_in_yr")))
    open("tiny.cfg", "w").write(text + "\n")
    out = subprocess.run([EXE, "tiny.cfg", "--mesh", mesh], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    host = des.Host(cfg_path="tiny.cfg", overrides="sim.modelname = cpu\n", mesh_file=mesh)
    assert (host.nnode, host.nelem) == (3018, 13850)
    st = driver.run(host, api=oracle_api())
    assert (st.steps, st.frames) == (4, 5)
    names = sorted(f for f in os.listdir(in_tmp) if f.startswith("gpu.") and not f.endswith(".info"))
    assert names == ["gpu.chkpt.%06d" % i for i in (0, 2, 4)] + ["gpu.save.%06d" % i for i in range(5)]
    for name in names:
        a, b = read_frame(name), read_frame(name.replace("gpu.", "cpu."))
        for k in a:
            if k != "walltime_sec":
                assert np.array_equal(a[k], b[k]), (name, k)


def test_phase_changes_on_the_device_loop_match_the_oracle_loop(in_tmp):
    """phase_changes on the host's markers every 10 steps, counts re-uploaded (dirty
    flag -> k_props) and compute_dt repeated with the new material mix -- the HIP engine under
    des_run against the CPU oracle under the same loop: the same markers change at the same steps,
    the frames agree (bit for bit where no libm is involved; the creep law's pow / exp are glibc's
    bits on the device too)."""
    from test_markers import PHASE_KW, PHASE_OV
    text = cfgs.make(**PHASE_KW)
    hd = des.Host(cfg_text=text, overrides=PHASE_OV + "sim.modelname = gpu\n")
    ho = des.Host(cfg_text=text, overrides=PHASE_OV + "sim.modelname = cpu\n")
    sd = driver.run(hd)
    so = driver.run(ho, api=oracle_api())
    assert sd.phase_changed_markers == so.phase_changed_markers > 0
    assert (sd.steps, sd.frames, sd.dt) == (so.steps, so.frames, so.dt)
    for frame in range(4):
        a, b = read_frame("gpu.save.%06d" % frame), read_frame("cpu.save.%06d" % frame)
        assert sorted(a) == sorted(b)
        for name in a:
            if name == "walltime_sec":
                continue
            if name.startswith("markerset") or name in ("material", "connectivity", "bcflag") or a[name].size % 8:
                assert np.array_equal(a[name], b[name]), (frame, name)
            else:
                x, y = a[name].view(np.float64), b[name].view(np.float64)
                assert np.abs(x - y).max() <= 1e-10 * max(np.abs(y).max(), 1e-300), (frame, name)


def test_remeshing_round_trip_on_the_device_matches_the_oracle_loop(in_tmp):
    """The same round trip (tests/test_driver_output.py) with the HIP engine: the device run stops at
    the same step for the same reason, hands the same state to the remesher, and after the restart on
    a NEW engine continues to the same frames as the oracle loop, bit for bit (elasto-plastic)."""
    from test_driver_output import remesh_round_trip
    sd = remesh_round_trip("gpu", None)
    so = remesh_round_trip("cpu", oracle_api())
    assert [(s.steps, s.remesh_needed, s.exit_code, s.last_frame) for s in sd] == \
           [(s.steps, s.remesh_needed, s.exit_code, s.last_frame) for s in so] == [(300, 2, 31, 4), (700, 0, 0, 9)]
    for frame in range(10):
        a, b = read_frame("gpu.save.%06d" % frame), read_frame("cpu.save.%06d" % frame)
        for name in a:
            if name != "walltime_sec":
                assert np.array_equal(a[name], b[name]), (frame, name)


def test_executable_takes_a_remesher_command(in_tmp):
    import sys
    from test_driver_output import REMESH_OV, FLATTENER
    text = "\n".join(l for l in cfgs.TEST3D.splitlines()
                     if not l.startswith(("max_time_in_yr", "output_time_interval_in_yr")))
    with open("model.cfg", "w") as f:
        f.write(cfgs.apply_overrides(text, REMESH_OV + "sim.modelname = exe\n"))
    mesh = os.path.join(des.REPO_ROOT, "tests", "golden", "test-3d.desmesh")
    out = subprocess.run([EXE, "model.cfg", "--mesh", mesh, "--remesher", "%s %s" % (sys.executable, FLATTENER)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Remeshing:" in out.stdout and "bottom_flattener" in out.stdout and out.stdout.count("Ending simulation.") == 1
    info = np.loadtxt("exe.info").reshape(-1, 8)
    assert info[-1, 1] == 700 and len(info) == 10


# ---- the remeshing round trip with a mesh whose node and element counts CHANGE, on the device -----------------
# tools/remesh_tool.py (a test tool: the reference's TetGen on the deformed box + nearest-neighbour remap, 13,850 -> ~7,700 tets)
# as the remesher: the restart runs on a NEW engine -- new allocations, new internal order, new patch lists and LDS sizes, new
# residual blocks -- which is where an allocation or patch-list bug of a changing mesh would sit.
def _remesh_pieces():
    import sys
    from test_distributed_run import REMESH_TOOL, REMESH_OV, _remesh_text
    if not os.access(os.path.join(des.REPO_ROOT, "oracle", "_ref", "tetmesh"), os.X_OK):
        pytest.skip("oracle/_ref/tetmesh is missing (make -C oracle ref)")
    mesh = os.path.join(des.REPO_ROOT, "tests", "golden", "test-3d.desmesh")
    cmd = "%s %s --resolution 1500" % (sys.executable, REMESH_TOOL)
    return _remesh_text(), REMESH_OV, mesh, cmd


def test_remeshing_round_trip_with_a_changing_element_count_on_the_device(in_tmp):
    text, ov, mesh, cmd = _remesh_pieces()
    make_host = lambda extra: des.Host(cfg_text=text, overrides=ov.replace("rtd", "gpu") + (extra or ""), mesh_file=None if extra else mesh)
    stats = driver.run_with_remesher(make_host, cmd)
    assert [(s.steps, s.remesh_needed, s.exit_code) for s in stats] == [(300, 2, 31), (400, 0, 0)]
    info = np.loadtxt("gpu.info").reshape(-1, 8)
    assert info[:, 1].tolist() == [0, 100, 200, 300, 300, 300, 400]
    assert info[4, 5:7].tolist() == [3018, 13850] and info[5, 6] != 13850 and info[5, 6] > 5000           # another mesh, another size
    ne = int(info[5, 6])
    # up to the remesh: the oracle loop writes the same frames (and stops for the same reason)
    st = driver.run(des.Host(cfg_text=text, overrides=ov.replace("rtd", "cpu"), mesh_file=mesh), api=oracle_api())
    assert (st.steps, st.remesh_needed) == (300, 2)
    for frame in range(5):
        a, b = read_frame("gpu.save.%06d" % frame), read_frame("cpu.save.%06d" % frame)
        for name in a:
            if name != "walltime_sec":
                assert np.array_equal(a[name], b[name]), (frame, name)
    # after it: the oracle loop restarted from the SAME pair (the one the device run's remesher left) takes the same 100 steps
    host = des.Host(cfg_text=text, overrides=ov.replace("rtd", "cpr")
                    + "sim.is_restarting = yes\nsim.restarting_from_modelname = gpu\nsim.restarting_from_frame = 5\n")
    assert host.nelem == ne
    st = driver.run(host, api=oracle_api())
    assert (st.steps, st.exit_code) == (400, 0)
    a, b = read_frame("gpu.save.000006"), read_frame("cpr.save.000006")
    for name in a:
        if name != "walltime_sec":
            assert np.array_equal(a[name], b[name]), name


def _worker_remesh_device(rank, world, port, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    from dynearthsol_amd.decomp import PhasedStepper, TorchComm
    from dynearthsol_amd.distributed import run_distributed_with_remesher
    from test_distributed_run import REMESH_TOOL, REMESH_OV, _remesh_text
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(out_dir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mesh = os.path.join(here, "golden", "test-3d.desmesh")
    comm = TorchComm(dist)
    make_host = lambda extra: des.Host(cfg_text=_remesh_text(), overrides=REMESH_OV + (extra or ""), mesh_file=None if extra else mesh)
    # one DEVICE engine per rank, both on this box's one GPU (RCCL refuses that: the two-phase step, ghost state over gloo)
    stats = run_distributed_with_remesher(make_host, "%s %s --resolution 1500" % (sys.executable, REMESH_TOOL), dist,
                                          engine_factory=lambda part: des.DeviceEngine(part, device=0),
                                          stepper=lambda e, p: PhasedStepper(e, p, comm))
    assert [(s.steps, s.remesh_needed, s.exit_code) for s in stats] == [(300, 2, 31), (400, 0, 0)]
    dist.barrier()
    dist.destroy_process_group()


def test_remeshing_round_trip_on_two_device_ranks(tmp_path):
    """... and on a decomposed mesh: two device engines (two processes on the one GPU), rank 0 remeshes, both restart on a new
    PARTITION of the new mesh -- new slabs, ghost regions, patch lists, residual blocks on the device; the 100 steps after the
    remesh equal those of ONE device engine restarted from the same pair, bit for bit."""
    import torch.multiprocessing as mp
    _remesh_pieces()
    from test_distributed_run import REMESH_OV, _remesh_text
    mp.spawn(_worker_remesh_device, args=(2, 33400 + os.getpid() % 500, str(tmp_path)), nprocs=2, join=True)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        info = np.loadtxt("rtd.info").reshape(-1, 8)
        assert info[:, 1].tolist() == [0, 100, 200, 300, 300, 300, 400] and info[5, 6] != 13850
        host = des.Host(cfg_text=_remesh_text(), overrides=REMESH_OV.replace("rtd", "rts")
                        + "sim.is_restarting = yes\nsim.restarting_from_modelname = rtd\nsim.restarting_from_frame = 5\n")
        assert host.nelem == int(info[5, 6])
        st = driver.run(host)
        assert (st.steps, st.exit_code) == (400, 0)
        a, b = read_frame("rtd.save.000006"), read_frame("rts.save.000006")
        for k in ("coordinate", "velocity", "temperature", "stress", "strain", "plastic strain"):
            assert np.array_equal(a[k], b[k]), k
    finally:
        os.chdir(cwd)
