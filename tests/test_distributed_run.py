"""The whole program on several ranks (dynearthsol_amd/distributed.py): every rank runs des_run()
over a collective engine table; rank 0 writes the frames.  Here over gloo with the CPU oracle as
the engine of each rank: the frames must equal those of a one-rank run to the bit."""
import os
import sys

import numpy as np

import cfgs
import dynearthsol_amd as des

OV = ("sim.max_steps = 30\nsim.output_step_interval = 10\nsim.checkpoint_frame_interval = 2\n"
      "mesh.quality_check_step_interval = 10\nsim.is_outputting_averaged_fields = yes\n")


def _worker(rank, world, port, out_dir):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    from dynearthsol_amd.decomp import PhasedStepper, TorchComm
    from dynearthsol_amd.distributed import run_distributed
    from oracle_binding import OracleEngine
    from test_decomp_cpu import _LocalMeshHost
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(out_dir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, nmat=2)), overrides=OV + "sim.modelname = multi\n")
    comm = TorchComm(dist)
    st = run_distributed(host, dist, engine_factory=lambda part: OracleEngine(_LocalMeshHost(part)),
                         stepper=lambda e, p: PhasedStepper(e, p, comm))
    assert (st.steps, st.frames, st.exit_code) == (30, 4, 0)
    dist.barrier()
    dist.destroy_process_group()


def test_three_ranks_write_the_frames_of_one_rank(tmp_path):
    import torch.multiprocessing as mp
    from dynearthsol_amd import driver
    from test_driver_output import oracle_api, read_frame
    world = 3
    port = 29300 + os.getpid() % 500
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, nmat=2)), overrides=OV + "sim.modelname = single\n")
        driver.run(host, api=oracle_api())
        files = sorted(f for f in os.listdir(tmp_path) if f.startswith("multi."))
        assert files == ["multi.chkpt.000000", "multi.chkpt.000002", "multi.info", "multi.save.000000",
                         "multi.save.000001", "multi.save.000002", "multi.save.000003"]
        for name in files:
            if name.endswith(".info"):
                a, b = np.loadtxt(name).reshape(-1, 8), np.loadtxt(name.replace("multi", "single")).reshape(-1, 8)
                assert np.array_equal(np.delete(a, 4, axis=1), np.delete(b, 4, axis=1))
                continue
            a, b = read_frame(name), read_frame(name.replace("multi", "single"))
            assert sorted(a) == sorted(b)
            for k in a:
                if k != "walltime_sec":
                    assert np.array_equal(a[k], b[k]), (name, k)
    finally:
        os.chdir(cwd)


# ---- the 2-D program (ndims = 2: two-component vectors, three-component tensors, stressyy, the wall-extent reduction) ------
OV2D = ("sim.max_steps = 30\nsim.output_step_interval = 10\nsim.checkpoint_frame_interval = 2\n"
        "mesh.quality_check_step_interval = 10\nsim.is_outputting_averaged_fields = yes\n")
KW2D = dict(cfgs.EVP, nmat=2, res=1e3)


def _worker_2d(rank, world, port, out_dir):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    from dynearthsol_amd.decomp import PhasedStepper, TorchComm
    from dynearthsol_amd.distributed import run_distributed
    from oracle_binding import OracleEngine
    from test_decomp_cpu import _LocalMeshHost
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(out_dir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    host = des.Host(cfg_text=cfgs.make(**KW2D), overrides=OV2D + "sim.modelname = multi\n", ndims=2)
    comm = TorchComm(dist)
    st = run_distributed(host, dist, engine_factory=lambda part: OracleEngine(_LocalMeshHost(part)),
                         stepper=lambda e, p: PhasedStepper(e, p, comm))
    assert (st.steps, st.frames, st.exit_code) == (30, 4, 0)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_write_the_2d_frames_of_one_rank(tmp_path):
    import torch.multiprocessing as mp
    from dynearthsol_amd import driver
    from test_driver_output import oracle_api, read_frame
    world = 2
    port = 30300 + os.getpid() % 500
    mp.spawn(_worker_2d, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        host = des.Host(cfg_text=cfgs.make(**KW2D), overrides=OV2D + "sim.modelname = single\n", ndims=2)
        driver.run(host, api=oracle_api(2))
        files = sorted(f for f in os.listdir(tmp_path) if f.startswith("multi."))
        assert files == ["multi.chkpt.000000", "multi.chkpt.000002", "multi.info", "multi.save.000000",
                         "multi.save.000001", "multi.save.000002", "multi.save.000003"]
        for name in files:
            if name.endswith(".info"):
                a, b = np.loadtxt(name).reshape(-1, 8), np.loadtxt(name.replace("multi", "single")).reshape(-1, 8)
                assert np.array_equal(np.delete(a, 4, axis=1), np.delete(b, 4, axis=1))
                continue
            a, b = read_frame(name, ndims=2), read_frame(name.replace("multi", "single"), ndims=2)
            assert sorted(a) == sorted(b)
            for k in a:
                if k != "walltime_sec":
                    assert np.array_equal(a[k], b[k]), (name, k)
    finally:
        os.chdir(cwd)


# ---- the remeshing round trip on two ranks, with a remesher that changes the node and element counts ------
REMESH_TOOL = os.path.join(des.REPO_ROOT, "tools", "remesh_tool.py")
REMESH_OV = ("sim.max_steps = 400\nmesh.quality_check_step_interval = 300\nmesh.max_boundary_distortion = 0.00039\n"
             "sim.modelname = rtd\n")


def _remesh_text():
    return "\n".join(l for l in cfgs.TEST3D.splitlines() if not l.startswith(("max_time_in_yr", "output_time_interval_in_yr")))


def _worker_remesh(rank, world, port, out_dir):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    from dynearthsol_amd.decomp import PhasedStepper, TorchComm
    from dynearthsol_amd.distributed import run_distributed_with_remesher
    from oracle_binding import OracleEngine
    from test_decomp_cpu import _LocalMeshHost
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(out_dir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mesh = os.path.join(here, "golden", "test-3d.desmesh")
    comm = TorchComm(dist)

    def make_host(extra):
        return des.Host(cfg_text=_remesh_text(), overrides=REMESH_OV + (extra or ""), mesh_file=None if extra else mesh)
    stats = run_distributed_with_remesher(make_host, "%s %s --resolution 1500" % (sys.executable, REMESH_TOOL), dist,
                                          engine_factory=lambda part: OracleEngine(_LocalMeshHost(part)),
                                          stepper=lambda e, p: PhasedStepper(e, p, comm))
    assert [(s.steps, s.remesh_needed, s.exit_code) for s in stats] == [(300, 2, 31), (400, 0, 0)]
    dist.barrier()
    dist.destroy_process_group()


def _worker_remesh_fails(rank, world, port, out_dir):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    from dynearthsol_amd.decomp import PhasedStepper, TorchComm
    from dynearthsol_amd.distributed import run_distributed_with_remesher
    from oracle_binding import OracleEngine
    from test_decomp_cpu import _LocalMeshHost
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(out_dir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mesh = os.path.join(here, "golden", "test-3d.desmesh")
    comm = TorchComm(dist)
    make_host = lambda extra: des.Host(cfg_text=_remesh_text(), overrides=REMESH_OV + (extra or ""), mesh_file=None if extra else mesh)
    try:
        run_distributed_with_remesher(make_host, "false", dist, engine_factory=lambda part: OracleEngine(_LocalMeshHost(part)),
                                      stepper=lambda e, p: PhasedStepper(e, p, comm))
        raise AssertionError("rank %d: a failed remesher went unnoticed" % rank)
    except des.DesError as e:
        # EVERY rank learns of rank 0's failure at once (no rank left in a barrier until the backend's timeout)
        assert e.code == 21 and "remesher failed on rank 0" in str(e) and "CalledProcessError" in str(e), str(e)
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_remesher_ends_every_rank_with_the_reason(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker_remesh_fails, args=(2, 32900 + os.getpid() % 500, str(tmp_path)), nprocs=2, join=True)


def test_remeshing_round_trip_on_two_ranks_with_a_changing_element_count(tmp_path):
    """benchmarks-cores/test-3d-remesh.cfg's situation (a displaced bottom node trips bad_mesh_quality at step 300) on two
    ranks: rank 0 runs tools/remesh_tool.py -- the reference's TetGen on the deformed box + nearest-neighbour remap, 13,850 ->
    ~7,700 tets --, both ranks restart from the pair it leaves: a new partition of the new mesh, clock and frame numbering
    continued.  The run after the remesh must be the one a single rank makes from the same pair, bit for bit."""
    import pytest
    import torch.multiprocessing as mp
    from dynearthsol_amd import driver
    from test_driver_output import oracle_api, read_frame, as_f64
    if not os.access(os.path.join(des.REPO_ROOT, "oracle", "_ref", "tetmesh"), os.X_OK):
        pytest.skip("oracle/_ref/tetmesh is missing (make -C oracle ref)")
    world = 2
    port = 32300 + os.getpid() % 500
    mp.spawn(_worker_remesh, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        info = np.loadtxt("rtd.info").reshape(-1, 8)
        assert info[:, 0].tolist() == [0, 1, 2, 3, 4, 5, 6] and info[:, 1].tolist() == [0, 100, 200, 300, 300, 300, 400]
        assert info[4, 5:7].tolist() == [3018, 13850] and info[5, 6] != 13850 and info[5, 6] > 5000      # a new mesh of another size
        new = read_frame("rtd.save.000005")
        ne = int(info[5, 6])
        assert as_f64(new["stress"], ne, 6).shape == (ne, 6) and np.isfinite(as_f64(new["stress"], ne, 6)).all()
        # lithostatic state carried over: the mean vertical stress stays where it was
        old = read_frame("rtd.save.000004")
        szz_old, szz_new = as_f64(old["stress"], 13850, 6)[:, 2].mean(), as_f64(new["stress"], ne, 6)[:, 2].mean()
        assert abs(szz_new - szz_old) < 0.05 * abs(szz_old)
        # one rank restarting from the same pair takes the same 100 steps: the frame of step 400 equal to the bit
        host = des.Host(cfg_text=_remesh_text(), overrides=REMESH_OV.replace("rtd", "rts")
                        + "sim.is_restarting = yes\nsim.restarting_from_modelname = rtd\nsim.restarting_from_frame = 5\n")
        assert host.nelem == ne
        st = driver.run(host, api=oracle_api())
        assert (st.steps, st.exit_code) == (400, 0)
        a, b = read_frame("rtd.save.000006"), read_frame("rts.save.000006")
        for k in ("coordinate", "velocity", "temperature", "stress", "strain", "plastic strain"):
            assert np.array_equal(a[k], b[k]), k
    finally:
        os.chdir(cwd)
