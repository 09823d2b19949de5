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


def test_rccl_send_recv_path_moves_halo_values():
    """The 1-GPU box cannot hold two RCCL ranks, so the ncclSend/ncclRecv path of
    des_dev_exchange is driven with the rank as its own neighbour: the values of the `send`
    nodes must arrive, for every exchange kind, at the `recv` nodes -- through pack kernel,
    grouped RCCL p2p on the engine's stream and unpack kernel."""
    import ctypes as C
    import os
    import types
    import torch.distributed as dist
    from dynearthsol_amd._structs import DesHalo
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29900 + os.getpid() % 90))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        host = des.Host(cfg_text=cfgs.make(**cfgs.EVP))
        eng = des.DeviceEngine(host)
        nn = host.nnode
        rng = np.random.default_rng(5)
        perm = rng.permutation(nn).astype(np.int32)
        k = nn // 3
        send, recv = np.sort(perm[:k]), np.sort(perm[k:2 * k])
        # two "neighbours", both rank 0, to cover the grouped loop
        cut = k // 2
        arrs = dict(nbr=np.zeros(2, np.int32), sp=np.array([0, cut, k], np.int32), rp=np.array([0, cut, k], np.int32),
                    send=send, recv=recv)
        pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        halo = DesHalo(0, nn, 2, pi(arrs["nbr"]), pi(arrs["sp"]), pi(send), pi(arrs["rp"]), pi(recv))
        eng.set_halo(types.SimpleNamespace(halo=halo, owned=(0, nn), host=host))
        eng.comm_init(dist, 0, 1)
        eng.init_from_host(host)
        vel = rng.standard_normal((3, nn)); eng.upload("VEL", vel)
        tem = rng.standard_normal(nn); eng.upload("TEMPERATURE", tem)
        ntmp = rng.standard_normal(nn); eng.upload("NTMP", ntmp)
        coord = eng.download("COORD").reshape(3, nn).copy()
        eng.exchange(2)          # {vx,vy,vz,x,y,z}
        eng.exchange(0)          # {T, ntmp}
        eng.sync()
        v2 = eng.download("VEL").reshape(3, nn); c2 = eng.download("COORD").reshape(3, nn)
        t2 = eng.download("TEMPERATURE"); n2 = eng.download("NTMP")
        assert np.array_equal(v2[:, recv], vel[:, send]) and np.array_equal(c2[:, recv], coord[:, send])
        assert np.array_equal(t2[recv], tem[send]) and np.array_equal(n2[recv], ntmp[send])
        untouched = np.setdiff1d(np.arange(nn), recv)
        assert np.array_equal(v2[:, untouched], vel[:, untouched]) and np.array_equal(t2[untouched], tem[untouched])
        ntmp3 = rng.standard_normal(nn); eng.upload("NTMP", ntmp3)
        eng.exchange(1); eng.sync()
        assert np.array_equal(eng.download("NTMP")[recv], ntmp3[send])
    finally:
        dist.destroy_process_group()


def test_overlapped_exchange_schedule_is_bit_identical_to_the_serial_one():
    """des_dev_step hides the exchanges after phases 0 and 1 behind the elements that touch no halo
    node (second stream + events; boundary element ranges afterwards).  Same arithmetic, different
    schedule: overlap on and off must agree to the bit.  (One GPU: the rank is its own neighbour;
    the halo values it receives are those of other nodes, so after the coordinate exchange the
    boundary elements degenerate -- the comparison is on raw bits, NaNs included.)"""
    import ctypes as C
    import os
    import types
    import torch.distributed as dist
    from dynearthsol_amd._structs import DesHalo
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29990 - os.getpid() % 90))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        kw = dict(cfgs.EVP, lx=60e3, ly=10e3, lz=8e3, res=1e3)
        results = []
        for overlap in (0, 1):
            host = des.Host(cfg_text=cfgs.make(**kw))
            nn, ne = host.nnode, host.nelem
            o0, o1 = 200, nn - 300                                  # "halo" = the first 200 and last 300 nodes
            from dynearthsol_amd._structs import DesMesh
            m = DesMesh.from_buffer_copy(host.mesh)                 # the engine lays its data out around
            m.owned_begin, m.owned_end = o0, o1                     # the owned range it is told at create
            eng = des.DeviceEngine(types.SimpleNamespace(params=host.params, mesh=m))
            recv = np.concatenate([np.arange(0, o0), np.arange(o1, nn)]).astype(np.int32)
            rng = np.random.default_rng(3)
            send = np.sort(rng.choice(np.arange(o0, o1), size=len(recv), replace=False)).astype(np.int32)
            cut = len(recv) // 2
            arrs = [np.zeros(2, np.int32), np.array([0, cut, len(recv)], np.int32)]
            pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
            halo = DesHalo(o0, o1, 2, pi(arrs[0]), pi(arrs[1]), pi(send), pi(arrs[1]), pi(recv))
            eng.set_halo(types.SimpleNamespace(halo=halo, owned=(o0, o1), host=host))
            eng.comm_init(dist, 0, 1)
            eng.init_from_host(host)
            assert eng.set_overlap(overlap) == overlap
            snaps = []
            for n in (1, 3):
                eng.step(n)
                snaps.append([eng.download(f).view(np.uint64).copy() for f in ("STRESS", "VEL", "COORD", "TEMPERATURE", "NTMP", "FORCE")])
            results.append(snaps)
            conn = host.array("connectivity").reshape(4, ne)
            touches_halo = ((conn < o0) | (conn >= o1)).any(axis=0)
            assert touches_halo[:256].any() and touches_halo[-256:].any() and not touches_halo[ne // 2]
        # away from the (degenerate) boundary elements the fields are regular numbers
        assert np.isfinite(results[0][0][0].view(np.float64)).mean() > 0.9
        assert np.isfinite(results[0][1][1].view(np.float64)).mean() > 0.5
        for a, b in zip(results[0], results[1]):
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
    finally:
        dist.destroy_process_group()
