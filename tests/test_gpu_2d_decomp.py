"""The 2-D build cut into node slabs (host/partition.cpp on triangles: slabs along x in the reference's renumbered
order + four element layers of ghost region), one device engine per slab on the ONE GPU of this box, against the
single 2-D engine: every nodal and elemental field and dt, bit for bit.  Two ways of stepping: des_dev_step_group
(ghost records by device-to-device copies) and the two-phase entry points with the exchange done by the caller
(des_dev_phase / des_dev_halo_pack / des_dev_halo_unpack / des_dev_wall_get / des_dev_wall_set: what a multi-process
run over torch.distributed uses).  Beside the ghost records a 2-D model shares the x0 wall's vertical extent
(apply_vbcs scales its depth profiles with it, bc.cxx:251-300) and, with a sheared bottom zone, the lowest node."""
import os

import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from dynearthsol_amd.decomp import DeviceGroup, PhasedStepper, run_loopback
from test_gpu_2d import BCS, bc_overrides

pytestmark = pytest.mark.gpu

NODE_FIELDS = (("COORD", 2), ("VEL", 2), ("TEMPERATURE", 1), ("MASS", 1), ("TMASS", 1), ("VOLUME_N", 1), ("FORCE", 2), ("DHACC", 1))
ELEM_FIELDS = (("STRESS", 3), ("STRAIN", 3), ("STRAIN_RATE", 3), ("PLSTRAIN", 1), ("DELTA_PLSTRAIN", 1), ("VISCOSITY", 1),
               ("VOLUME", 1), ("VOLUME_OLD", 1), ("DPRESSURE", 1), ("STRESSYY", 1), ("EDVOLDT", 1))


def _compare(group, ref, calls, phased=False):
    steppers = [PhasedStepper(e, p, None) for e, p in zip(group.engines, group.parts)]
    for n in calls:
        sref = ref.step(n)
        if phased:
            run_loopback(steppers, n)
            for e in group.engines:
                s = e.step(0)
                assert (s.dt, s.time, s.steps, s.status) == (sref.dt, sref.time, sref.steps, 0)
        else:
            for s in group.step(n):
                assert (s.dt, s.time, s.steps, s.status) == (sref.dt, sref.time, sref.steps, 0)
                # over ALL ranks' owned nodes (des_dev.h), the same on every rank; summed over other blocks than one engine's
                assert abs(s.l2_residual - sref.l2_residual) <= 1e-12 * sref.l2_residual
        for f, c in NODE_FIELDS:
            assert np.array_equal(group.download(f, c, "node"), ref.download(f)), f
        for f, c in ELEM_FIELDS:
            assert np.array_equal(group.download(f, c, "elem"), ref.download(f)), f


def _pair(host, nranks):
    ref = des.DeviceEngine(host)
    dt_ref = ref.init_from_host(host)
    group = DeviceGroup(host, nranks)
    assert group.init_from_host() == dt_ref
    assert sum(p.owned[1] - p.owned[0] for p in group.parts) == host.nnode
    assert sum(int(p.elem_owned.sum()) for p in group.parts) == host.nelem
    return ref, group


@pytest.mark.parametrize("nranks,phased", [(2, False), (3, True), (4, False)])
def test_2d_model_cut_n_ways_is_the_single_engine_bit_for_bit(nranks, phased):
    # two materials with a geotherm (evp), water load, surface diffusion with its marine branch, dhacc reset + top-element
    # rescaling every 7th step; compute_dt every 10th
    kw = dict(cfgs.EVP, nmat=2, res=1e3, qcsi=7, water="yes", control="surf_base_level = -100\nsurf_diff_ratio_marine = 0.5\n")
    host = des.Host(cfg_text=cfgs.make(**kw), ndims=2)
    ref, group = _pair(host, nranks)
    try:
        _compare(group, ref, (33, 1, 26), phased)
        assert np.abs(ref.download("DHACC")).max() > 0
    finally:
        group.close()


@pytest.mark.parametrize("bc", BCS)
def test_2d_boundary_conditions_on_a_cut_mesh(bc):
    # depth-dependent side velocities scaled by the x0 wall's extent on BOTH walls (the x1 wall lies on the last rank,
    # the x0 wall on the first), the sheared bottom zone (lowest node of the whole mesh), Neumann tractions
    host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EP, res=1e3)), overrides=bc_overrides(bc), ndims=2)
    ref, group = _pair(host, 3)
    try:
        _compare(group, ref, (20, 20, 20))
    finally:
        group.close()


def test_x1_wall_profile_uses_the_x0_walls_extent_across_ranks():
    # bc.cxx:299: the x1 wall's depth profile is laid out over the x0 wall's extent; x1 = 1 with a profile
    ov = ("bc.vbc_x1 = 1\nbc.vbc_val_x1 = 1e-9\nbc.vbc_val_division_x1_min = 0.25\nbc.vbc_val_division_x1_max = 0.7\n"
          "bc.vbc_val_x1_ratio0 = 1\nbc.vbc_val_x1_ratio1 = 0.6\nbc.vbc_val_x1_ratio2 = 0.3\nbc.vbc_val_x1_ratio3 = 0.1\n")
    host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, res=1e3)), overrides=ov, ndims=2)
    ref, group = _pair(host, 4)
    try:
        _compare(group, ref, (25, 25))
    finally:
        group.close()


def test_reference_test_topo_cut_3_ways_2000_steps():
    """benchmarks-cores/test-topo.cfg on the reference's Triangle mesh of topo.poly (10 km ridge, surface diffusivity
    1e-2, eight materials, evp with a weak zone): 3 slabs against one engine for the benchmark's 2000 steps."""
    host = des.Host(cfg_text=cfgs.TEST_TINY, overrides=cfgs.TEST_TOPO_OVERRIDES, ndims=2,
                    mesh_file=os.path.join(des.REPO_ROOT, "tests", "golden", "test-topo.desmesh"))
    ref, group = _pair(host, 3)
    try:
        _compare(group, ref, (500, 500, 999, 1))
    finally:
        group.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_a_million_triangles_cut_8_ways(overlap):
    """overlap: the overlapped schedule (des_dev_set_overlap; round 4) -- transfer, unpack and the wall's extent of step t on
    a side stream beside compute_mass, update_temperature + compute_dvoldt and update_stress of step t + 1 on the blocks /
    elements far from the cut; the surface bookkeeping of step t and the blocks near the cut behind the join."""
    kw = dict(cfgs.EVP, lx=400e3, lz=100e3, res=250.0)
    host = des.Host(cfg_text=cfgs.make(**kw), ndims=2)
    assert host.nelem == 2 * 1600 * 400
    ref, group = _pair(host, 8)
    try:
        ghost_share = sum(p.nelem for p in group.parts) / host.nelem - 1
        print("8 ranks: %.2f %% ghost-region elements" % (100 * ghost_share))
        for e in group.engines:
            e.set_overlap(overlap)
            assert e.comm_info()["overlapped"] == overlap
        _compare(group, ref, (21, 1))
    finally:
        group.close()


@pytest.mark.parametrize("nranks", [2, 3])
def test_2d_overlapped_schedule_on_a_cut_mesh(nranks):
    """the model of test_2d_model_cut_n_ways (two materials, water load, surface diffusion with its marine branch, dhacc
    reset + top-element rescaling every 7th step, compute_dt every 10th) on 200 x 50 km at 500 m, so that the slabs have
    blocks far from the cut: the group on the overlapped schedule == the single engine; switching schedules between calls"""
    kw = dict(cfgs.EVP, nmat=2, lx=200e3, lz=50e3, res=500.0, qcsi=7, water="yes", control="surf_base_level = -100\nsurf_diff_ratio_marine = 0.5\n")
    host = des.Host(cfg_text=cfgs.make(**kw), ndims=2)
    ref, group = _pair(host, nranks)
    try:
        for e in group.engines:
            e.set_overlap(True)
            assert e.comm_info()["overlapped"]
        _compare(group, ref, (33, 1))
        for e in group.engines: e.set_overlap(False)
        _compare(group, ref, (12,))
        for e in group.engines: e.set_overlap(True)
        _compare(group, ref, (26,))
        assert np.abs(ref.download("DHACC")).max() > 0
    finally:
        group.close()


def test_2d_step_on_rccl_equals_the_two_phase_step():
    """des_dev_step of a decomposed 2-D engine with a communicator attached (des_dev_comm_init): pack -> one grouped
    ncclSend / ncclRecv per neighbour -> unpack -> ncclAllReduce(MAX) of the wall extent, compute_dt's ncclAllReduce(MIN),
    all on the engine's stream.  RCCL refuses two ranks on this box's one GPU, so the middle slab of a three-way cut is
    its own neighbour (lists cut to equal lengths): the physics of that is meaningless, but the two-phase entry points
    driven the same way -- every message unpacked where RCCL delivers it -- must leave the same bits in every field."""
    import ctypes as C
    import types
    import torch.distributed as dist
    from dynearthsol_amd._structs import DesHalo
    from dynearthsol_amd.decomp import Partition
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 90))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        # (120 x 30 km at 500 m: the middle slab has blocks far from both cuts, which the overlapped schedule needs)
        host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, nmat=2, lx=120e3, lz=30e3, res=500.0)), ndims=2)
        part = Partition(host, 3, 1)
        assert len(part.nbr_rank) == 2
        pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        cut = lambda s_, r_: [(a[:min(len(a), len(b))], b[:min(len(a), len(b))]) for a, b in zip(s_, r_)]
        nodes, elems = cut(part.send_idx, part.recv_idx), cut(part.esend_idx, part.erecv_idx)
        ptr = np.cumsum([0] + [len(a) for a, _ in nodes]).astype(np.int32)
        eptr = np.cumsum([0] + [len(a) for a, _ in elems]).astype(np.int32)
        send, recv = [np.ascontiguousarray(np.concatenate([p[i] for p in nodes]), dtype=np.int32) for i in (0, 1)]
        esend, erecv = [np.ascontiguousarray(np.concatenate([p[i] for p in elems]), dtype=np.int32) for i in (0, 1)]
        assert ptr[-1] > 20 and eptr[-1] > 20
        nbr = np.zeros(2, np.int32)
        halo = DesHalo(part.owned[0], part.owned[1], 4, 2, pi(nbr), pi(ptr), pi(send), pi(ptr), pi(recv),
                       pi(eptr), pi(esend), pi(eptr), pi(erecv))
        fake = types.SimpleNamespace(halo=halo, owned=part.owned, host=host)
        fields = ("COORD", "VEL", "STRESS", "STRAIN", "TEMPERATURE", "PLSTRAIN", "VOLUME", "MASS", "STRAIN_RATE", "STRESSYY", "DHACC")
        results = []
        for rccl in (True, False, "overlapped"):
            eng = des.DeviceEngine(part)
            eng.set_halo(fake)
            if rccl:
                eng.comm_init(dist, 0, 1)
                assert eng.comm_info()["rccl_ranks"] == 1
                # the overlapped schedule on RCCL: transfer + unpack + the wall's all-reduce on the side stream
                eng.set_overlap(rccl == "overlapped")
                assert eng.comm_info()["overlapped"] == (rccl == "overlapped")
            for f, name in (("COORD", "coord"), ("COORD0", "coord"), ("ELEMMARKERS", "elemmarkers"), ("VEL", "vel")):
                eng.upload(f, part.local(name))
            if not rccl:
                eng.wall_set(eng.wall_get())
            eng.init_geometry()
            for f, name in (("TEMPERATURE", "temperature"), ("RADIOGENIC", "radiogenic"), ("STRESS", "stress"), ("STRAIN", "strain"),
                            ("PLSTRAIN", "plstrain"), ("VISCOSITY", "viscosity"), ("STRESSYY", "stressyy")):
                eng.upload(f, part.local(name))
            dt0 = eng.compute_dt() if rccl else eng.dt_finalize(eng.dt_partials(True))
            snaps = []
            for n in (2, 7, 4):                        # the last call crosses step 10: compute_dt inside it
                if rccl:
                    eng.step(n)
                else:
                    for _ in range(n):
                        eng.phase(0)
                        bufs = [(eng.halo_pack(0, a, part.node_width), eng.halo_pack(1, ea, part.elem_width))
                                for (a, _), (ea, _) in zip(nodes, elems)]
                        for (nb, eb), (_, b), (_, eb_idx) in zip(bufs, nodes, elems):
                            eng.halo_unpack(0, b, nb)
                            eng.halo_unpack(1, eb_idx, eb)
                        eng.wall_set(eng.wall_get())
                        if eng.phase(1):
                            eng.dt_finalize(eng.dt_partials(False))
                snaps.append({f: eng.download(f) for f in fields})
            results.append((dt0, eng.step(0).dt, snaps))
            eng.close()
        assert results[0][0] == results[1][0]
        assert results[0][1] == results[1][1] or (np.isnan(results[0][1]) and np.isnan(results[1][1]))
        o0, o1 = part.owned
        assert np.isfinite(results[0][2][0]["VEL"].reshape(2, -1)[:, o0:o1]).mean() > 0.5, "nothing left to compare"
        assert results[2][:2] == results[0][:2] or np.isnan(results[0][1])
        for a, b, c in zip(results[0][2], results[1][2], results[2][2]):
            for f in fields:
                assert np.array_equal(a[f], b[f], equal_nan=True), f
                assert np.array_equal(a[f], c[f], equal_nan=True), f + " (overlapped schedule)"
    finally:
        dist.destroy_process_group()


def test_a_cut_2d_engine_refuses_what_it_does_not_offer():
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP), ndims=2)
    group = DeviceGroup(host, 2)
    try:
        e = group.engines[0]
        with pytest.raises(des.DesError) as ei:
            e.step(1)                                   # des_dev_step on its own, without a communicator
        assert ei.value.code == 31
        assert e._lib.des_dev_exchange(e._h) == 30
    finally:
        group.close()
    # (control.has_PT on a cut mesh used to be refused here; since round 4 it runs: test_2d_pseudo_transient_loop_on_a_cut_mesh)


# ---- the whole 2-D program on several ranks (dynearthsol_amd/distributed.py) -----------------------------------
RUN_OV = ("sim.max_steps = 60\nsim.output_step_interval = 20\nsim.checkpoint_frame_interval = 2\n"
          "mesh.quality_check_step_interval = 10\nsim.is_outputting_averaged_fields = yes\n")
RUN_KW = dict(cfgs.EVP, nmat=2, res=1e3)


def _worker(rank, world, port, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    from dynearthsol_amd.distributed import run_distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"                # every rank on the one GPU of this box
    os.chdir(out_dir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    host = des.Host(cfg_text=cfgs.make(**RUN_KW), overrides=RUN_OV + "sim.modelname = multi\n", ndims=2)
    st = run_distributed(host, dist)              # 2-D: the two-phase step, ghost records over torch.distributed
    assert (st.steps, st.frames, st.exit_code) == (60, 4, 0)
    dist.barrier()
    dist.destroy_process_group()


def test_the_2d_program_on_three_ranks_writes_the_frames_of_one(tmp_path):
    """des_run() on three processes (one device engine each, all on this box's GPU), the 2-D model cut three ways:
    rank 0's frames, checkpoints and .info equal those of the plain one-engine run, bit for bit."""
    import torch.multiprocessing as mp
    from dynearthsol_amd import driver
    from test_driver_output import read_frame
    world = 3
    mp.spawn(_worker, args=(world, 29300 + os.getpid() % 500, str(tmp_path)), nprocs=world, join=True)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        host = des.Host(cfg_text=cfgs.make(**RUN_KW), overrides=RUN_OV + "sim.modelname = single\n", ndims=2)
        st = driver.run(host)
        assert (st.steps, st.frames) == (60, 4)
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


def test_2d_pseudo_transient_loop_on_a_cut_mesh():
    """control.has_PT on a 2-D model cut 3 ways (round 4): des_dev_step_group runs the loop for all engines in lockstep -- ghost
    region and wall extent refreshed before every iteration, the residual in global block order -- against the single 2-D
    engine: same iteration counts, same bits; then the same through the two-phase entry points"""
    ov = "control.has_PT = yes\ncontrol.PT_max_iter = 40\ncontrol.PT_relative_tolerance = 1e-3\n"
    host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EP, res=1e3)), overrides=ov, ndims=2)
    for phased in (False, True):
        ref, group = _pair(host, 3)
        try:
            steppers = [PhasedStepper(e, p, None) for e, p in zip(group.engines, group.parts)]
            for n in (2, 9):
                sref = ref.step(n)
                assert sref.n_pt_iterations > 0
                if phased:
                    run_loopback(steppers, n)
                    assert all(st.n_pt_iterations == sref.n_pt_iterations for st in steppers)
                else:
                    for s in group.step(n):
                        assert (s.dt, s.steps, s.n_pt_iterations) == (sref.dt, sref.steps, sref.n_pt_iterations)
                        # the step ends in the loop's residual: the GLOBAL block sum, the same bits on every engine -- not
                        # added across the engines once more (it used to come back sqrt(3) too large here)
                        assert s.l2_residual == sref.l2_residual
                for f, c in NODE_FIELDS:
                    assert np.array_equal(group.download(f, c, "node"), ref.download(f)), f
                for f, c in ELEM_FIELDS:
                    assert np.array_equal(group.download(f, c, "elem"), ref.download(f)), f
        finally:
            group.close()
