"""Domain decomposition on the device: several engines on ONE GPU, each on its slab of the
mesh plus its four-layer ghost region, the ghost state moved by the test harness through
des_dev_halo_pack/unpack between the two phases of des_dev_phase -- against one undecomposed
engine stepping with des_dev_step.  The RCCL transport of a real multi-GPU run uses the same
lists, pack/unpack kernels and phase order (des_dev_step), only the copy in the middle differs."""
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


@pytest.mark.parametrize("seed", range(8))
def test_decomposed_random_option_combinations(seed):
    """The ghost-region scheme under random option combinations (test_gpu_parity._random_overrides):
    2-4 ranks on one GPU against one engine, bit for bit."""
    from test_gpu_parity import _random_overrides
    rng = np.random.default_rng(2000 + seed)
    ov = _random_overrides(rng)
    kw = dict(cfgs.EVP, rheol=["elasto-plastic", "elastic"][seed % 2], nmat=1 + seed % 3 % 2, lx=60e3)
    test_decomposed_engines_match_one_engine_bit_for_bit("random%d" % seed, kw, 2 + seed % 3, ov)


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


def test_rccl_send_recv_path_moves_the_ghost_state():
    """The 1-GPU box cannot hold two RCCL ranks, so the ncclSend/ncclRecv path of
    des_dev_exchange is driven with the rank as its own neighbour: the state of the `send`
    nodes / elements must arrive at the `recv` nodes / elements -- through the pack kernel, the
    grouped RCCL p2p on the engine's stream and the unpack kernel."""
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
        nn, ne = host.nnode, host.nelem
        rng = np.random.default_rng(5)
        perm, eperm = rng.permutation(nn).astype(np.int32), rng.permutation(ne).astype(np.int32)
        k, ke = nn // 3, ne // 4
        send, recv = np.sort(perm[:k]), np.sort(perm[k:2 * k])
        esend, erecv = np.sort(eperm[:ke]), np.sort(eperm[ke:2 * ke])
        # two "neighbours", both rank 0, to cover the grouped loop and the message layout
        nbr = np.zeros(2, np.int32)
        ptr, eptr = np.array([0, k // 2, k], np.int32), np.array([0, ke // 3, ke], np.int32)
        pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        halo = DesHalo(0, nn, 4, 2, pi(nbr), pi(ptr), pi(send), pi(ptr), pi(recv), pi(eptr), pi(esend), pi(eptr), pi(erecv))
        eng.set_halo(types.SimpleNamespace(halo=halo, owned=(0, nn), host=host))
        eng.comm_init(dist, 0, 1)
        # the start-up self-check bench.py runs on every rank: the exchange's own messages with a verifiable pattern, the
        # reductions, the rank count -- and it must refuse a communicator of the wrong size
        eng.comm_selfcheck(1)
        with pytest.raises(des.DesError, match="ncclCommCount = 1, expected 2"):
            eng.comm_selfcheck(2)
        eng.init_from_host(host)
        vel = rng.standard_normal((3, nn)); eng.upload("VEL", vel)
        tem = rng.standard_normal(nn); eng.upload("TEMPERATURE", tem)
        stress = rng.standard_normal((6, ne)); eng.upload("STRESS", stress)
        strain = rng.standard_normal((6, ne)); eng.upload("STRAIN", strain)
        pls = rng.standard_normal(ne); eng.upload("PLSTRAIN", pls)
        coord = eng.download("COORD").reshape(3, nn).copy()
        mass = eng.download("MASS").copy()
        eng.exchange()
        eng.sync()
        v2, c2 = eng.download("VEL").reshape(3, nn), eng.download("COORD").reshape(3, nn)
        t2, s2 = eng.download("TEMPERATURE"), eng.download("STRESS").reshape(6, ne)
        e2, p2 = eng.download("STRAIN").reshape(6, ne), eng.download("PLSTRAIN")
        assert np.array_equal(v2[:, recv], vel[:, send]) and np.array_equal(c2[:, recv], coord[:, send])
        assert np.array_equal(t2[recv], tem[send]) and np.array_equal(eng.download("MASS"), mass)
        assert np.array_equal(s2[:, erecv], stress[:, esend]) and np.array_equal(e2[:, erecv], strain[:, esend])
        assert np.array_equal(p2[erecv], pls[esend])
        keep, ekeep = np.setdiff1d(np.arange(nn), recv), np.setdiff1d(np.arange(ne), erecv)
        assert np.array_equal(v2[:, keep], vel[:, keep]) and np.array_equal(t2[keep], tem[keep])
        assert np.array_equal(s2[:, ekeep], stress[:, ekeep]) and np.array_equal(p2[ekeep], pls[ekeep])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["strong", "weak"])
def test_two_rank_bench_rehearsal(mode):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank),
    rehearsed with both ranks on the one GPU of this box: RCCL refuses two ranks on one device, so
    the ghost state is staged through the host over gloo (DES_BENCH_TRANSPORT=host, which the line
    must say); everything else -- partition, engines, the collective step count on every rank, the
    JSON line -- is the real thing.  Default = strong scaling on the fixed mesh (BASELINE configs[3]),
    --weak keeps the per-GPU size.  (A profiling run taken by rank 0 alone used to hang here.)"""
    import json
    import os
    import subprocess
    import sys
    env = dict(os.environ, DES_BENCH_BACKEND="gloo", DES_BENCH_TRANSPORT="host", DES_BENCH_DEVICE="0",
               DES_BENCH_VERBOSE="1", DES_BENCH_WATCHDOG="100")
    port = 29400 + os.getpid() % 90 + (7 if mode == "weak" else 0)
    # strong: the PLAIN command of the bench contract -- bench.py starts its own ranks (bench.py: launch_ranks);
    # weak: under torch.distributed.run, as the driver launches N > 1
    launcher = [sys.executable] if mode == "strong" else \
               [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(port)]
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run(launcher + [os.path.join(des.REPO_ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "12", "--warmup", "2", "--resolution", "2000", "--cpu-steps", "0",
                          "--series-resolution", "2500", "--series-steps", "6"]
                         + (["--weak"] if mode == "weak" else []),
                         capture_output=True, text=True, timeout=170, env=env, cwd=des.REPO_ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    c = r["config"]
    assert r["n_gpus"] == 2 and r["steps"] == 12 and c["nan_entries"] == 0 and c["status"] == 0
    assert r["scaling"] == mode and c["nelem"] == (50000 if mode == "strong" else 100000) and r["value"] > 0
    assert "kernel_ms_per_call" in c and ("cut 2 ways (strong scaling)" in c["workload"]) == (mode == "strong")
    # this rehearsal must not be mistaken for an RCCL run
    assert c["rccl_ranks"] == 0 and "rehearsal" in c["parallelism"] and c["exchange_us_per_rank"] == [-1.0, -1.0]
    assert 0 < c["ghost_work_share"] < 0.5 and c["nelem_local_sum"] > c["nelem"]
    # per rank: the step without the exchange (sum of the passes' HIP-event times), in us
    assert len(c["step_us_without_exchange_per_rank"]) == 2 and all(t > 0 for t in c["step_us_without_exchange_per_rank"])
    assert c["rccl_selfcheck"].startswith("not run")
    # the second, larger mesh of the same box in the same line (strong scaling only)
    big = c["large_mesh_series"]
    if mode == "strong":
        assert big["nelem"] == 160 * 8 * 4 * 5 and big["steps"] == 6 and big["value"] > 0 and big["status"] == 0
        assert "cut 2 ways" in big["workload"] and 0 < big["ghost_work_share"] < 0.5
    else:
        assert big is None


def test_bench_refuses_to_report_without_rccl():
    """Two ranks on ONE device: RCCL declines, and bench.py must then fail (exit 3) instead of
    quietly timing another transport."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, DES_BENCH_BACKEND="gloo", DES_BENCH_DEVICE="0", DES_BENCH_VERBOSE="1", DES_BENCH_WATCHDOG="100")
    env.pop("DES_BENCH_TRANSPORT", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    # the plain command: the launcher inside bench.py hands the ranks' own code (3) on, not torch.distributed.run's 1
    out = subprocess.run([sys.executable, os.path.join(des.REPO_ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "4", "--warmup", "1", "--resolution", "4000", "--cpu-steps", "0"],
                         capture_output=True, text=True, timeout=170, env=env, cwd=des.REPO_ROOT)
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], "a result line was printed without RCCL"
    assert "did not come up" in out.stderr


def test_overlapped_schedule_gives_the_same_bits():
    """DES_OVERLAP=1 (transfer + unpack on a side stream while the end-of-step pass of the interior
    elements runs, engine/exchange.hpp) against the in-order schedule, on the middle slab of a
    three-way partition: real owned range, real ghost region, real element groups.  The 1-GPU box
    cannot hold two RCCL ranks, so the slab is its own neighbour (lists cut to equal lengths per
    neighbour) -- the physics of that is meaningless, the dependencies between the two streams are
    the real ones: both schedules must produce the same bits, compute_dt steps included."""
    import ctypes as C
    import os
    import types
    import torch.distributed as dist
    from dynearthsol_amd._structs import DesHalo
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 90))
    dist.init_process_group("gloo", rank=0, world_size=1)
    old = os.environ.get("DES_OVERLAP")
    # With the surface step left to the next step's passes (engine/launch.hpp: s2_defer_ok, the in-order schedule's
    # default) the exchange carries the heights BEFORE the diffusion instead of after it.  Between real neighbours
    # that is the same model (tests/test_gpu_headline_decomp.py); with the slab as its own neighbour it is another
    # scrambled ghost region, so the schedules would no longer see the same garbage: pinned off here.
    old_defer = os.environ.get("DES_S2_DEFER")
    os.environ["DES_S2_DEFER"] = "0"
    try:
        host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, lx=90e3)))
        part = Partition(host, 3, 1)
        assert len(part.nbr_rank) == 2
        pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        cut = lambda s, r: [(a[:min(len(a), len(b))], b[:min(len(a), len(b))]) for a, b in zip(s, r)]
        nodes, elems = cut(part.send_idx, part.recv_idx), cut(part.esend_idx, part.erecv_idx)
        ptr = np.cumsum([0] + [len(a) for a, _ in nodes]).astype(np.int32)
        eptr = np.cumsum([0] + [len(a) for a, _ in elems]).astype(np.int32)
        send, recv = [np.ascontiguousarray(np.concatenate([p[i] for p in nodes]), dtype=np.int32) for i in (0, 1)]
        esend, erecv = [np.ascontiguousarray(np.concatenate([p[i] for p in elems]), dtype=np.int32) for i in (0, 1)]
        assert ptr[-1] > 100 and eptr[-1] > 100
        nbr = np.zeros(2, np.int32)
        halo = DesHalo(part.owned[0], part.owned[1], 4, 2, pi(nbr), pi(ptr), pi(send), pi(ptr), pi(recv),
                       pi(eptr), pi(esend), pi(eptr), pi(erecv))
        results = []
        old_patch = os.environ.get("DES_PATCH")
        for mode in ("0", "1", "classic"):
            # third run: the classic passes, in order -- EN1 / EN2 / EN3 on a mesh with a ghost region,
            # inside multi-step calls, against N1 / N2 / E3 + N3
            os.environ["DES_OVERLAP"] = "0" if mode == "classic" else mode
            if mode == "classic":
                os.environ["DES_PATCH"] = "0"
            eng = des.DeviceEngine(part)
            eng.set_halo(types.SimpleNamespace(halo=halo, owned=part.owned, host=host))
            eng.comm_init(dist, 0, 1)
            info = eng.comm_info()
            assert info["rccl_ranks"] == 1 and info["overlapped"] == (mode == "1")
            if mode == "classic":
                if old_patch is None:
                    os.environ.pop("DES_PATCH", None)
                else:
                    os.environ["DES_PATCH"] = old_patch
            for f, name in (("COORD", "coord"), ("COORD0", "coord"), ("ELEMMARKERS", "elemmarkers"), ("VEL", "vel")):
                eng.upload(f, part.local(name))
            eng.init_geometry()
            for f, name in (("TEMPERATURE", "temperature"), ("RADIOGENIC", "radiogenic"), ("STRESS", "stress"), ("STRAIN", "strain"),
                            ("PLSTRAIN", "plstrain"), ("VISCOSITY", "viscosity")):
                eng.upload(f, part.local(name))
            eng.compute_dt()
            # Being its own neighbour scrambles the ghost region (the received coordinates belong elsewhere), so
            # NaNs creep inwards one element layer per step: compare early (mostly finite) and across a
            # compute_dt step, NaN == NaN -- what matters is that both schedules do the same thing to every entry.
            fields = ("COORD", "VEL", "STRESS", "STRAIN", "TEMPERATURE", "PLSTRAIN", "VOLUME", "MASS", "STRAIN_RATE")
            eng.step(2)
            early = {f: eng.download(f) for f in fields}
            eng.step(7)
            sc = eng.step(4)                               # crosses step 10: compute_dt inside the call
            results.append((sc.dt, early, {f: eng.download(f) for f in fields}))
            del eng
        o0, o1 = part.owned
        assert np.isfinite(results[0][1]["VEL"].reshape(3, -1)[:, o0:o1]).mean() > 0.5, "nothing left to compare"
        assert results[0][0] == results[1][0] or (np.isnan(results[0][0]) and np.isnan(results[1][0]))
        for other in (1, 2):
            assert results[0][0] == results[other][0] or (np.isnan(results[0][0]) and np.isnan(results[other][0]))
            for k in (1, 2):
                for f in fields:
                    assert np.array_equal(results[0][k][f], results[other][k][f], equal_nan=True), (other, k, f)
    finally:
        if old is None:
            os.environ.pop("DES_OVERLAP", None)
        else:
            os.environ["DES_OVERLAP"] = old
        if old_defer is None:
            os.environ.pop("DES_S2_DEFER", None)
        else:
            os.environ["DES_S2_DEFER"] = old_defer
        dist.destroy_process_group()


def test_switching_schedules_between_calls_keeps_the_bits():
    """des_dev_set_overlap between two des_dev_step calls -- what bench.py's schedule probe does on N > 1 GPUs before its timed
    region -- on an engine with an RCCL communicator (the middle slab of a three-way cut as its own neighbour, as above):
    in order / overlapped / in order again must leave what the in-order schedule alone leaves."""
    import ctypes as C
    import os
    import types
    import torch.distributed as dist
    from dynearthsol_amd._structs import DesHalo
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 90))
    dist.init_process_group("gloo", rank=0, world_size=1)
    old_defer = os.environ.get("DES_S2_DEFER")
    os.environ["DES_S2_DEFER"] = "0"                   # (see the test above: the slab as its own neighbour)
    try:
        host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, lx=90e3)))
        part = Partition(host, 3, 1)
        pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        cut = lambda s, r: [(a[:min(len(a), len(b))], b[:min(len(a), len(b))]) for a, b in zip(s, r)]
        nodes, elems = cut(part.send_idx, part.recv_idx), cut(part.esend_idx, part.erecv_idx)
        ptr = np.cumsum([0] + [len(a) for a, _ in nodes]).astype(np.int32)
        eptr = np.cumsum([0] + [len(a) for a, _ in elems]).astype(np.int32)
        send, recv = [np.ascontiguousarray(np.concatenate([p[i] for p in nodes]), dtype=np.int32) for i in (0, 1)]
        esend, erecv = [np.ascontiguousarray(np.concatenate([p[i] for p in elems]), dtype=np.int32) for i in (0, 1)]
        nbr = np.zeros(2, np.int32)
        halo = DesHalo(part.owned[0], part.owned[1], 4, 2, pi(nbr), pi(ptr), pi(send), pi(ptr), pi(recv),
                       pi(eptr), pi(esend), pi(eptr), pi(erecv))
        fields = ("COORD", "VEL", "STRESS", "STRAIN", "TEMPERATURE", "PLSTRAIN", "VOLUME", "MASS")
        out = []
        for switching in (False, True):
            eng = des.DeviceEngine(part)
            eng.set_halo(types.SimpleNamespace(halo=halo, owned=part.owned, host=host))
            eng.comm_init(dist, 0, 1)
            for f, name in (("COORD", "coord"), ("COORD0", "coord"), ("ELEMMARKERS", "elemmarkers"), ("VEL", "vel")):
                eng.upload(f, part.local(name))
            eng.init_geometry()
            for f, name in (("TEMPERATURE", "temperature"), ("RADIOGENIC", "radiogenic"), ("STRESS", "stress"), ("STRAIN", "strain"),
                            ("PLSTRAIN", "plstrain"), ("VISCOSITY", "viscosity")):
                eng.upload(f, part.local(name))
            snaps = []
            eng.set_clock(eng.compute_dt(), 0.0, 6)        # (the ghost region is scrambled: NaNs creep inwards a layer per step --
            for k, n in enumerate((2, 3, 2)):          #  few steps, started at step 6 so that the second call crosses compute_dt)
                if switching:
                    eng.set_overlap(k == 1)
                    assert eng.comm_info()["overlapped"] == (k == 1)
                eng.step(n, want_scalars=False)
                snaps.append({f: eng.download(f) for f in fields})
            out.append(snaps)
            eng.close()
        o0, o1 = part.owned
        assert np.isfinite(out[0][0]["VEL"].reshape(3, -1)[:, o0:o1]).mean() > 0.5, "nothing left to compare"
        for a, b in zip(out[0], out[1]):               # (after the compute_dt of a scrambled mesh: NaN == NaN)
            for f in fields:
                assert np.array_equal(a[f], b[f], equal_nan=True), f
    finally:
        if old_defer is None:
            os.environ.pop("DES_S2_DEFER", None)
        else:
            os.environ["DES_S2_DEFER"] = old_defer
        dist.destroy_process_group()


PT_OV = "control.has_PT = yes\ncontrol.PT_max_iter = 40\ncontrol.PT_relative_tolerance = %s\ncontrol.has_moving_mesh = %s\n"
PT_NODE = (("COORD", 3), ("VEL", 3), ("TEMPERATURE", 1), ("MASS", 1), ("VOLUME_N", 1), ("FORCE", 3))
PT_ELEM = (("STRESS", 6), ("STRAIN", 6), ("STRAIN_RATE", 6), ("PLSTRAIN", 1), ("VISCOSITY", 1), ("VOLUME", 1), ("VOLUME_OLD", 1))


@pytest.mark.parametrize("nranks,tol,moving", [(2, "1e-2", "yes"), (4, "1e-4", "yes"), (3, "1e-4", "no")])
def test_pseudo_transient_loop_on_a_decomposed_mesh(nranks, tol, moving):
    """control.has_PT (dynearthsol.cxx:803-864) on N ranks (round 4; des_dev_set_halo used to refuse it): des_dev_step_group
    refreshes the ghost region before every iteration and puts the residual together in global block order (des_params.h:
    DES_RES_BLOCK), so every rank takes the decision the single engine takes.  Same iteration counts, same bits in every field --
    and the residual itself equal to the bit, on the device as against the oracle (one association everywhere)."""
    from oracle_binding import OracleEngine
    from dynearthsol_amd.decomp import DeviceGroup
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides=PT_OV % (tol, moving))
    ref = des.DeviceEngine(host)
    dt_ref = ref.init_from_host(host)
    ora = OracleEngine(host)
    ora.init_from_host(host)
    group = DeviceGroup(host, nranks)
    try:
        assert group.init_from_host() == dt_ref
        total = 0
        for n in (1, 4, 8):                                  # crosses step 10 (compute_dt)
            sref, so = ref.step(n), ora.step(n)
            assert sref.n_pt_iterations == so.n_pt_iterations > 0
            for s in group.step(n):
                assert (s.dt, s.time, s.steps, s.status, s.n_pt_iterations) == (sref.dt, sref.time, sref.steps, 0, sref.n_pt_iterations)
            total += sref.n_pt_iterations
            for f, c in PT_NODE:
                assert np.array_equal(group.download(f, c, "node"), ref.download(f)), f
            for f, c in PT_ELEM:
                assert np.array_equal(group.download(f, c, "elem"), ref.download(f)), f
        assert total >= 13
    finally:
        group.close()


def test_pseudo_transient_loop_through_the_two_phase_entry_points():
    """the same loop driven by the caller (des_dev_phase 0 -> 2 ... 2 -> 3, des_dev_residual_blocks / _set: what a run over
    another transport does, dynearthsol_amd/decomp.py: run_loopback) with three device engines on this GPU"""
    from dynearthsol_amd.decomp import PhasedStepper, run_loopback, LoopbackComm, init_rank_mesh, init_rank_fields, assemble
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides=PT_OV % ("1e-3", "yes"))
    ref = des.DeviceEngine(host)
    dt_ref = ref.init_from_host(host)
    parts = [Partition(host, 3, r) for r in range(3)]
    engines = [des.DeviceEngine(p) for p in parts]
    steppers = [PhasedStepper(e, p, None) for e, p in zip(engines, parts)]
    comm = LoopbackComm(steppers)
    for e, p in zip(engines, parts):
        init_rank_mesh(e, p)
    for e, p in zip(engines, parts):
        init_rank_fields(e, p)
    assert all(d == dt_ref for d in comm.reduce_dt_all(recompute=True))
    for n in (2, 9):
        sref = ref.step(n)
        run_loopback(steppers, n)
        assert all(st.n_pt_iterations == sref.n_pt_iterations > 0 for st in steppers)
        for f, c in PT_NODE:
            assert np.array_equal(assemble(parts, [e.download(f) for e in engines], c, host.nnode, "node"), ref.download(f)), f
        for f, c in PT_ELEM:
            assert np.array_equal(assemble(parts, [e.download(f) for e in engines], c, host.nelem, "elem"), ref.download(f)), f


def test_initial_body_force_adjustment_on_a_decomposed_mesh():
    """ic.has_body_force_adjustment (dynearthsol.cxx:546-591) for the engines of a group: the pseudo-transient loop on the
    initial state with the Neumann tractions held back, in lockstep -- iteration count and bits of the single engine, the
    steps that follow included"""
    from dynearthsol_amd.decomp import DeviceGroup
    ov = ("control.has_PT = yes\ncontrol.PT_max_iter = 25\ncontrol.PT_relative_tolerance = 1e-4\n"
          "bc.stress_bc_z1 = 3\nbc.stress_val_z1 = 2e6\nbc.stress_bc_x0 = 1\nbc.stress_val_x0 = -1e6\n")
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides=ov)
    ref = des.DeviceEngine(host)
    ref.init_from_host(host)
    group = DeviceGroup(host, 3)
    try:
        group.init_from_host()
        pushed = ref.download("STRESS") * 1.03           # (the lithostatic start is in equilibrium: push it out of balance)
        ref.upload("STRESS", pushed)
        group.upload("STRESS", pushed, 6, "elem")
        sref = ref.body_force_adjustment()
        sg = group.body_force_adjustment()
        assert sref.n_pt_iterations > 3 and all(s.n_pt_iterations == sref.n_pt_iterations and s.l2_residual == sref.l2_residual for s in sg)
        for n in (1, 3, 8):
            sref = ref.step(n)
            for s in group.step(n):
                assert (s.dt, s.steps, s.n_pt_iterations) == (sref.dt, sref.steps, sref.n_pt_iterations)
        for f, c in PT_NODE:
            assert np.array_equal(group.download(f, c, "node"), ref.download(f)), f
        for f, c in PT_ELEM:
            assert np.array_equal(group.download(f, c, "elem"), ref.download(f)), f
    finally:
        group.close()
