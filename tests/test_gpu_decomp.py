"""Domain decomposition on the device: several engines on ONE GPU, each on its slab of the
mesh, halo values moved by the test harness through des_dev_halo_pack/unpack between the five
phases of des_dev_phase -- against one undecomposed engine stepping with des_dev_step.
The RCCL transport of a real multi-GPU run uses the same lists, pack/unpack kernels and phase
order (des_dev_step), only the copy in the middle differs."""
import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from dynearthsol_amd.decomp import Partition, PhasedStepper, LoopbackComm, init_rank, run_loopback, assemble

pytestmark = pytest.mark.gpu

NODE_FIELDS = (("COORD", 3), ("VEL", 3), ("TEMPERATURE", 1), ("MASS", 1), ("VOLUME_N", 1), ("FORCE", 3), ("DHACC", 1))
ELEM_FIELDS = (("STRESS", 6), ("STRAIN", 6), ("STRAIN_RATE", 6), ("PLSTRAIN", 1), ("VISCOSITY", 1), ("VOLUME", 1),
               ("VOLUME_OLD", 1), ("DPRESSURE", 1))


class _NoReduce:
    def reduce_dt(self, engine, recompute):
        return None


@pytest.mark.parametrize("name,kw,nranks,overrides", [
    ("ep_2", cfgs.EP, 2, None),
    ("ep_3_nosurf", cfgs.EP, 3, "control.surface_process_option = 0\n"),
    ("evp_4", cfgs.EVP, 4, None),
    ("evp_2mat_3", dict(cfgs.EVP, nmat=2), 3, None),
    ("ep_2_nonmd", cfgs.EP, 2, "control.is_using_mixed_stress = no\n"),
])
def test_decomposed_engines_match_one_engine_bit_for_bit(name, kw, nranks, overrides):
    host = des.Host(cfg_text=cfgs.make(**kw), overrides=overrides)
    ref = des.DeviceEngine(host)
    dt_ref = ref.init_from_host(host)
    parts = [Partition(host, nranks, r) for r in range(nranks)]
    engines = [des.DeviceEngine(p) for p in parts]
    steppers = [PhasedStepper(e, p, None) for e, p in zip(engines, parts)]
    comm = LoopbackComm(steppers)
    for e, p in zip(engines, parts):
        init_rank(e, p, _NoReduce())
    assert all(d == dt_ref for d in comm.reduce_dt_all(recompute=True))
    nsteps = 24
    ref.step(nsteps)
    run_loopback(steppers, nsteps)
    for f, c in NODE_FIELDS:
        got = assemble(parts, [e.download(f) for e in engines], c, host.nnode, "node")
        assert np.array_equal(got, ref.download(f)), f
    for f, c in ELEM_FIELDS:
        got = assemble(parts, [e.download(f) for e in engines], c, host.nelem, "elem")
        assert np.array_equal(got, ref.download(f)), f
    sref = ref.step(0)
    for e in engines:
        sc = e.step(0)
        assert (sc.dt, sc.time, sc.steps) == (sref.dt, sref.time, sref.steps)


def test_single_rank_communicator_is_a_no_op():
    """des_dev_step with a 1-rank RCCL communicator attached (the N=1 case of the multi-GPU
    bench) gives the same bits as without."""
    import torch.distributed as dist
    import os
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        host = des.Host(cfg_text=cfgs.make(**cfgs.EVP))
        a, b = des.DeviceEngine(host), des.DeviceEngine(Partition(host, 1, 0))
        a.init_from_host(host)
        part = b._host
        b.set_halo(part)
        b.comm_init(dist, 0, 1)
        b.init_from_host(host)
        a.step(20); b.step(20)
        for f in ("COORD", "VEL", "STRESS", "TEMPERATURE"):
            assert np.array_equal(a.download(f), b.download(f)), f
    finally:
        dist.destroy_process_group()
