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
