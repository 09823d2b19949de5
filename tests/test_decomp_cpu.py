"""Domain decomposition on the CPU: the slab partition with its four-layer ghost region (host
library), the exchange lists and the two-phase step with ONE exchange, checked with the oracle
against an undecomposed run.
(a) several ranks in one process (loopback); (b) world_size-2 gloo over torch.distributed."""
import os
import sys

import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from dynearthsol_amd.decomp import Partition, PhasedStepper, init_rank, run_loopback, LoopbackComm, TorchComm, assemble
from oracle_binding import OracleEngine

NODE_FIELDS = (("COORD", 3), ("VEL", 3), ("TEMPERATURE", 1), ("MASS", 1), ("VOLUME_N", 1), ("FORCE", 3))
ELEM_FIELDS = (("STRESS", 6), ("STRAIN", 6), ("STRAIN_RATE", 6), ("PLSTRAIN", 1), ("VISCOSITY", 1), ("VOLUME", 1))


class _LocalMeshHost:
    """Duck-typed host for OracleEngine(host): the engine only needs .params and .mesh."""
    def __init__(self, part):
        self.params, self.mesh, self._keep = part.params, part.mesh, part
        self.ndims = getattr(part, "ndims", 3)


def build(kw, nranks, overrides=None):
    host = des.Host(cfg_text=cfgs.make(**kw), overrides=overrides)
    parts = [Partition(host, nranks, r) for r in range(nranks)]
    return host, parts


def test_partition_covers_the_mesh_exactly():
    host, parts = build(cfgs.EP, 3)
    owned = np.concatenate([p.l2g_node[p.owned[0]:p.owned[1]] for p in parts])
    assert np.array_equal(np.sort(owned), np.arange(host.nnode))            # every node owned once
    eowned = np.concatenate([p.l2g_elem[p.elem_owned] for p in parts])
    assert np.array_equal(np.sort(eowned), np.arange(host.nelem))           # every element owned once
    conn = host.array("connectivity").reshape(4, -1)
    for p in parts:
        a, b = p.node_ranges[p.rank], p.node_ranges[p.rank + 1]
        assert p.halo.nlayers == 4
        # the local mesh: the supports of the owned nodes, grown by three more element layers
        reach = (conn >= a) & (conn < b)
        held = reach.any(axis=0)
        for _ in range(3):
            nodes = np.unique(conn[:, held])
            held = np.isin(conn, nodes).any(axis=0)
        assert np.array_equal(np.nonzero(held)[0], p.l2g_elem)
        assert np.array_equal(np.unique(conn[:, held]), p.l2g_node)
        assert np.all(np.diff(p.l2g_node) > 0) and np.all(np.diff(p.l2g_elem) > 0)   # global order kept
        # send list of r towards q == recv list of q from r, as GLOBAL ids, nodes and elements
        for q, sidx, seidx in zip(p.nbr_rank, p.send_idx, p.esend_idx):
            other = parts[q]
            k = other.nbr_rank.index(p.rank)
            assert np.array_equal(p.l2g_node[sidx], other.l2g_node[other.recv_idx[k]])
            assert np.array_equal(p.l2g_elem[seidx], other.l2g_elem[other.erecv_idx[k]])
            assert p.elem_owned[seidx].all() and ((sidx >= p.owned[0]) & (sidx < p.owned[1])).all()
        ghosts = np.concatenate(p.recv_idx) if p.recv_idx else np.zeros(0, int)
        non_owned = np.setdiff1d(np.arange(p.nnode), np.arange(p.owned[0], p.owned[1]))
        assert np.array_equal(np.sort(ghosts), non_owned)
    # load balance of the owned work within 15 %
    w = [int(p.elem_owned.sum()) for p in parts]
    assert max(w) < 1.15 * min(w) + 200


@pytest.mark.parametrize("name,kw,nranks", [
    ("ep_2", cfgs.EP, 2), ("ep_3", cfgs.EP, 3), ("evp_4", cfgs.EVP, 4), ("yield_2", cfgs.YIELD, 2),
    ("evp_2mat_3", dict(cfgs.EVP, nmat=2), 3),
])
def test_decomposed_oracle_is_bit_identical_to_one_rank(name, kw, nranks):
    host, parts = build(kw, nranks)
    ref = OracleEngine(host)
    dt_ref = ref.init_from_host(host)
    engines = [OracleEngine(_LocalMeshHost(p)) for p in parts]
    steppers = [PhasedStepper(e, p, None) for e, p in zip(engines, parts)]
    comm = LoopbackComm(steppers)
    for e, p in zip(engines, parts):
        e.set_halo(p)
    # init(): same order as EngineBase.init_from_host, then the first compute_dt across ranks
    class _C:                                         # reduce_dt is collective: do it for all at once below
        def reduce_dt(self, engine, recompute): return None
    for e, p in zip(engines, parts):
        init_rank(e, p, _C())
    dts = comm.reduce_dt_all(recompute=True)
    assert all(d == dt_ref for d in dts)
    nsteps = 30
    ref.step(nsteps)
    run_loopback(steppers, nsteps)
    for f, c in NODE_FIELDS:
        got = assemble(parts, [e.download(f) for e in engines], c, host.nnode, "node")
        assert np.array_equal(got, ref.download(f)), f
    for f, c in ELEM_FIELDS:
        got = assemble(parts, [e.download(f) for e in engines], c, host.nelem, "elem")
        assert np.array_equal(got, ref.download(f)), f
    assert all(e.step(0).dt == ref.step(0).dt for e in engines)


@pytest.mark.parametrize("seed", range(6))
def test_decomposed_oracle_random_option_combinations(seed):
    from test_gpu_parity import _random_overrides
    ov = _random_overrides(np.random.default_rng(3000 + seed))
    kw = dict(cfgs.EVP, rheol=["elasto-plastic", "elasto-visco-plastic"][seed % 2], nmat=1 + seed % 3 % 2, lx=60e3)
    host = des.Host(cfg_text=cfgs.make(**kw), overrides=ov)
    nranks = 2 + seed % 3
    parts = [Partition(host, nranks, r) for r in range(nranks)]
    ref = OracleEngine(host)
    dt_ref = ref.init_from_host(host)
    engines = [OracleEngine(_LocalMeshHost(p)) for p in parts]
    steppers = [PhasedStepper(e, p, None) for e, p in zip(engines, parts)]
    comm = LoopbackComm(steppers)

    class _C:
        def reduce_dt(self, engine, recompute): return None
    for e, p in zip(engines, parts):
        init_rank(e, p, _C())
    assert all(d == dt_ref for d in comm.reduce_dt_all(recompute=True))
    ref.step(15)
    run_loopback(steppers, 15)
    for f, c in NODE_FIELDS:
        assert np.array_equal(assemble(parts, [e.download(f) for e in engines], c, host.nnode, "node"), ref.download(f)), (f, ov)
    for f, c in ELEM_FIELDS:
        assert np.array_equal(assemble(parts, [e.download(f) for e in engines], c, host.nelem, "elem"), ref.download(f)), (f, ov)


PT_OV = "control.has_PT = yes\ncontrol.PT_max_iter = 40\ncontrol.PT_relative_tolerance = %s\ncontrol.has_moving_mesh = %s\n"


@pytest.mark.parametrize("ndims,nranks,tol,moving", [(3, 2, "1e-2", "yes"), (3, 3, "1e-4", "yes"), (3, 4, "1e-4", "no"), (2, 3, "1e-3", "yes")])
def test_pseudo_transient_loop_on_a_decomposed_mesh(ndims, nranks, tol, moving):
    """control.has_PT (dynearthsol.cxx:803-864) on N ranks: the ghost region refreshed before every iteration, the residual
    put together across ranks in global block order (des_params.h: DES_RES_BLOCK), so that every rank -- and a run on one
    rank -- takes the same decision.  Same iteration counts, same bits as the single-domain oracle."""
    kw = dict(cfgs.EP, res=1e3) if ndims == 2 else cfgs.EP
    host = des.Host(cfg_text=cfgs.make(**kw), overrides=PT_OV % (tol, moving), **({"ndims": 2} if ndims == 2 else {}))
    parts = [Partition(host, nranks, r) for r in range(nranks)]
    ref = OracleEngine(host)
    dt_ref = ref.init_from_host(host)
    engines = [OracleEngine(_LocalMeshHost(p)) for p in parts]
    steppers = [PhasedStepper(e, p, None) for e, p in zip(engines, parts)]
    comm = LoopbackComm(steppers)
    from dynearthsol_amd.decomp import init_rank_mesh, init_rank_fields
    for e, p in zip(engines, parts):
        init_rank_mesh(e, p)
    if ndims == 2:
        comm.reduce_wall_all()
    for e, p in zip(engines, parts):
        init_rank_fields(e, p)
    assert all(d == dt_ref for d in comm.reduce_dt_all(recompute=True))
    total = 0
    for n in (1, 4, 8):                                  # crosses step 10 (compute_dt)
        so = ref.step(n)
        run_loopback(steppers, n)
        assert all(st.n_pt_iterations == so.n_pt_iterations > 0 for st in steppers)
        assert all(e.step(0).l2_residual == so.l2_residual for e in engines)      # one association, whatever the partition
        total += so.n_pt_iterations
        nf, ef = (NODE_FIELDS_2D, ELEM_FIELDS_2D) if ndims == 2 else (NODE_FIELDS, ELEM_FIELDS)
        for f, c in nf:
            assert np.array_equal(assemble(parts, [e.download(f) for e in engines], c, host.nnode, "node"), ref.download(f)), f
        for f, c in ef:
            assert np.array_equal(assemble(parts, [e.download(f) for e in engines], c, host.nelem, "elem"), ref.download(f)), f
    assert total >= 13


def _gloo_worker(rank, world, port, nsteps, out_dir, overrides=None):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    host = des.Host(cfg_text=cfgs.make(**(cfgs.EP if overrides else cfgs.EVP)), overrides=overrides)
    part = Partition(host, world, rank)
    eng = OracleEngine(_LocalMeshHost(part))
    comm = TorchComm(dist)
    dt = init_rank(eng, part, comm)
    stepper = PhasedStepper(eng, part, comm)
    stepper.step(nsteps)
    o0, o1 = part.owned
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), dt0=dt, dt=eng.step(0).dt, n_pt=stepper.n_pt_iterations,
             l2=eng.step(0).l2_residual,
             nodes=part.l2g_node[o0:o1], elems=part.l2g_elem[part.elem_owned],
             vel=eng.download("VEL").reshape(3, -1)[:, o0:o1],
             T=eng.download("TEMPERATURE")[o0:o1], stress=eng.download("STRESS").reshape(6, -1)[:, part.elem_owned])
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_gloo_match_one_rank(tmp_path):
    import torch.multiprocessing as mp
    nsteps, world = 25, 2
    port = 29500 + os.getpid() % 2000
    mp.spawn(_gloo_worker, args=(world, port, nsteps, str(tmp_path)), nprocs=world, join=True)
    host = des.Host(cfg_text=cfgs.make(**cfgs.EVP))
    ref = OracleEngine(host)
    dt0 = ref.init_from_host(host)
    sc = ref.step(nsteps)
    vel, T, stress = ref.download("VEL").reshape(3, -1), ref.download("TEMPERATURE"), ref.download("STRESS").reshape(6, -1)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert d["dt0"] == dt0 and d["dt"] == sc.dt
        assert np.array_equal(d["vel"], vel[:, d["nodes"]])
        assert np.array_equal(d["T"], T[d["nodes"]])
        assert np.array_equal(d["stress"], stress[:, d["elems"]])


def test_pseudo_transient_loop_two_ranks_over_gloo(tmp_path):
    """the same loop with one process per rank: the exchange per iteration and the residual's all-reduce over torch.distributed"""
    import torch.multiprocessing as mp
    nsteps, world = 6, 2
    ov = PT_OV % ("1e-4", "yes")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_gloo_worker, args=(world, port, nsteps, str(tmp_path), ov), nprocs=world, join=True)
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides=ov)
    ref = OracleEngine(host)
    dt0 = ref.init_from_host(host)
    sc = ref.step(nsteps)
    vel, T, stress = ref.download("VEL").reshape(3, -1), ref.download("TEMPERATURE"), ref.download("STRESS").reshape(6, -1)
    assert sc.n_pt_iterations > nsteps
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert d["dt0"] == dt0 and d["dt"] == sc.dt and d["n_pt"] == sc.n_pt_iterations and d["l2"] == sc.l2_residual
        assert np.array_equal(d["vel"], vel[:, d["nodes"]])
        assert np.array_equal(d["T"], T[d["nodes"]])
        assert np.array_equal(d["stress"], stress[:, d["elems"]])


# ---- the 2-D build on the slab decomposition (host/partition.cpp on triangles, decomp.py's 2-D record widths, the
#      wall-extent reduction of des_dev.h) with the 2-D oracle as every rank's engine ----------------------------------
NODE_FIELDS_2D = (("COORD", 2), ("VEL", 2), ("TEMPERATURE", 1), ("MASS", 1), ("VOLUME_N", 1), ("FORCE", 2), ("DHACC", 1))
ELEM_FIELDS_2D = (("STRESS", 3), ("STRAIN", 3), ("STRAIN_RATE", 3), ("PLSTRAIN", 1), ("VISCOSITY", 1), ("VOLUME", 1),
                  ("VOLUME_OLD", 1), ("DPRESSURE", 1), ("STRESSYY", 1))
# a depth profile on the x1 wall, laid out over the x0 wall's extent (bc.cxx:299): the last rank needs the first rank's wall
X1_PROFILE = ("bc.vbc_x1 = 1\nbc.vbc_val_x1 = 1e-9\nbc.vbc_val_division_x1_min = 0.25\nbc.vbc_val_division_x1_max = 0.7\n"
              "bc.vbc_val_x1_ratio0 = 1\nbc.vbc_val_x1_ratio1 = 0.6\nbc.vbc_val_x1_ratio2 = 0.3\nbc.vbc_val_x1_ratio3 = 0.1\n")
SHEAR_ZONE = "bc.vbc_x0 = 3\nbc.vbc_x1 = 2\nbc.bottom_shear_zone_thickness = 3e3\n"


@pytest.mark.parametrize("name,kw,ov,nranks", [
    ("evp_2mat_water_3", dict(cfgs.EVP, nmat=2, res=1e3, qcsi=7, water="yes"), None, 3),
    ("x1_profile_4", dict(cfgs.EVP, res=1e3), X1_PROFILE, 4),
    ("shear_zone_2", dict(cfgs.EP, res=1e3), SHEAR_ZONE, 2),
])
def test_decomposed_2d_oracle_is_bit_identical_to_one_rank(name, kw, ov, nranks):
    host = des.Host(cfg_text=cfgs.make(**kw), overrides=ov, ndims=2)
    parts = [Partition(host, nranks, r) for r in range(nranks)]
    ref = OracleEngine(host)
    dt_ref = ref.init_from_host(host)
    engines = [OracleEngine(_LocalMeshHost(p)) for p in parts]
    steppers = [PhasedStepper(e, p, None) for e, p in zip(engines, parts)]
    comm = LoopbackComm(steppers)
    from dynearthsol_amd.decomp import init_rank_mesh, init_rank_fields
    for e, p in zip(engines, parts):
        init_rank_mesh(e, p)
    comm.reduce_wall_all()
    for e, p in zip(engines, parts):
        init_rank_fields(e, p)
    dts = comm.reduce_dt_all(recompute=True)
    assert all(d == dt_ref for d in dts)
    nsteps = 32
    ref.step(nsteps)
    run_loopback(steppers, nsteps)
    for f, c in NODE_FIELDS_2D:
        got = assemble(parts, [e.download(f) for e in engines], c, host.nnode, "node")
        assert np.array_equal(got, ref.download(f)), f
    for f, c in ELEM_FIELDS_2D:
        got = assemble(parts, [e.download(f) for e in engines], c, host.nelem, "elem")
        assert np.array_equal(got, ref.download(f)), f
    assert all(e.step(0).dt == ref.step(0).dt for e in engines)


def _gloo_worker_2d(rank, world, port, nsteps, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, res=1e3)), overrides=X1_PROFILE, ndims=2)
    part = Partition(host, world, rank)
    eng = OracleEngine(_LocalMeshHost(part))
    comm = TorchComm(dist)
    dt = init_rank(eng, part, comm)
    PhasedStepper(eng, part, comm).step(nsteps)
    o0, o1 = part.owned
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), dt0=dt, dt=eng.step(0).dt,
             nodes=part.l2g_node[o0:o1], elems=part.l2g_elem[part.elem_owned],
             vel=eng.download("VEL").reshape(2, -1)[:, o0:o1],
             T=eng.download("TEMPERATURE")[o0:o1], stress=eng.download("STRESS").reshape(3, -1)[:, part.elem_owned])
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_gloo_match_one_rank_2d(tmp_path):
    import torch.multiprocessing as mp
    nsteps, world = 25, 2
    port = 31500 + os.getpid() % 2000
    mp.spawn(_gloo_worker_2d, args=(world, port, nsteps, str(tmp_path)), nprocs=world, join=True)
    host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, res=1e3)), overrides=X1_PROFILE, ndims=2)
    ref = OracleEngine(host)
    dt0 = ref.init_from_host(host)
    sc = ref.step(nsteps)
    vel, T, stress = ref.download("VEL").reshape(2, -1), ref.download("TEMPERATURE"), ref.download("STRESS").reshape(3, -1)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert d["dt0"] == dt0 and d["dt"] == sc.dt
        assert np.array_equal(d["vel"], vel[:, d["nodes"]])
        assert np.array_equal(d["T"], T[d["nodes"]])
        assert np.array_equal(d["stress"], stress[:, d["elems"]])
