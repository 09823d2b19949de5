#!/usr/bin/env python3
"""What an overlapped schedule could buy the 2-D build at 8 ranks, as far as ONE GPU can tell: the step
time of the middle rank's shard of the 1.28M-triangle bench mesh cut 8 ways (a) as a plain engine on the local mesh (no
exchange), (b) with the ghost-region exchange of a step inside des_dev_step on RCCL, the slab as its own neighbour (lists
cut to equal lengths, as tests/test_gpu_2d_decomp.py::test_2d_step_on_rccl_equals_the_two_phase_step does: the physics of
that is meaningless, the work per step is the rank's plus pack + grouped ncclSend / ncclRecv + unpack + the wall-extent
all-reduce).  (b) - (a) = what the exchange costs in the stream when nothing hides it -- on one device, i.e. without the
xGMI hop -- and (c) the same on the overlapped schedule.

    python tools/time_2d_shard.py [steps]        (on the MI355X box)
"""
import ctypes as C
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfgs                                            # noqa: E402
import dynearthsol_amd as des                          # noqa: E402
from dynearthsol_amd._structs import DesHalo           # noqa: E402
from dynearthsol_amd.decomp import Partition           # noqa: E402


def init(eng, part, rccl):
    for f, name in (("COORD", "coord"), ("COORD0", "coord"), ("ELEMMARKERS", "elemmarkers"), ("VEL", "vel")):
        eng.upload(f, part.local(name))
    if not rccl:
        eng.wall_set(eng.wall_get())
    eng.init_geometry()
    for f, name in (("TEMPERATURE", "temperature"), ("RADIOGENIC", "radiogenic"), ("STRESS", "stress"), ("STRAIN", "strain"),
                    ("PLSTRAIN", "plstrain"), ("VISCOSITY", "viscosity"), ("STRESSYY", "stressyy")):
        eng.upload(f, part.local(name))
    eng.compute_dt()


HOST_US = {}


def timed(eng, steps, tag=None):
    """us per step on the device (HIP events); HOST_US[tag]: what the host needs to ENQUEUE a step (the call returns before
    the device is done) -- a step cannot be faster than that"""
    import time
    eng.step(40, want_scalars=False)
    eng.sync()
    best, host = 1e9, 1e9
    for _ in range(3):
        eng.timer_start()
        t0 = time.perf_counter()
        eng.step(steps, want_scalars=False)
        host = min(host, (time.perf_counter() - t0) / steps * 1e6)
        best = min(best, eng.timer_stop() / steps)
    if tag: HOST_US[tag] = host
    return 1e3 * best


def main():
    import torch.distributed as dist
    argv = sys.argv[1:]
    only = None                              # --only inorder | overlapped: just that engine (for a profiler run)
    if "--only" in argv:
        i = argv.index("--only"); only = argv[i + 1]; del argv[i:i + 2]
    steps = int(argv[0]) if argv else 200
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 90))
    dist.init_process_group("gloo", rank=0, world_size=1)
    host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, lx=400e3, lz=100e3, res=250.0)), ndims=2)
    t_whole = t_plain = float("nan")
    if not only:
        whole = des.DeviceEngine(host)
        whole.init_from_host(host)
        t_whole = timed(whole, steps)
        whole.close()
    part = Partition(host, 8, 4)
    if not only:
        plain = des.DeviceEngine(part)
        init(plain, part, False)
        t_plain = timed(plain, steps, 'plain')
        plain.close()
    # the slab as its own RCCL neighbour, real list sizes
    pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    cut = lambda s_, r_: [(a[:min(len(a), len(b))], b[:min(len(a), len(b))]) for a, b in zip(s_, r_)]
    nodes, elems = cut(part.send_idx, part.recv_idx), cut(part.esend_idx, part.erecv_idx)
    ptr = np.cumsum([0] + [len(a) for a, _ in nodes]).astype(np.int32)
    eptr = np.cumsum([0] + [len(a) for a, _ in elems]).astype(np.int32)
    send, recv = [np.ascontiguousarray(np.concatenate([p[i] for p in nodes]), dtype=np.int32) for i in (0, 1)]
    esend, erecv = [np.ascontiguousarray(np.concatenate([p[i] for p in elems]), dtype=np.int32) for i in (0, 1)]
    nbr = np.zeros(2, np.int32)
    halo = DesHalo(part.owned[0], part.owned[1], 4, 2, pi(nbr), pi(ptr), pi(send), pi(ptr), pi(recv), pi(eptr), pi(esend), pi(eptr), pi(erecv))
    eng = des.DeviceEngine(part)
    eng.set_halo(types.SimpleNamespace(halo=halo, owned=part.owned, host=host))
    eng.comm_init(dist, 0, 1)
    init(eng, part, True)
    if only == "overlapped":
        eng.set_overlap(True)
    if only:
        print("%s: %.1f us per step" % (only, timed(eng, steps)))
        eng.close(); dist.destroy_process_group()
        return
    t_rccl = timed(eng, steps, 'rccl')
    # (c) the overlapped schedule: transfer, unpack and the wall's all-reduce on the side stream beside the next step's passes
    # on the blocks / elements far from the cut
    eng.set_overlap(True)
    assert eng.comm_info()["overlapped"]
    t_ov = timed(eng, steps, 'ov')
    eng.close()
    print("# 2-D bench mesh %d triangles / %d nodes; %d-step calls" % (host.nelem, host.nnode, steps))
    print("whole mesh on one GPU                                   %8.1f us per step" % t_whole)
    print("middle shard of the 8-way cut (%d triangles), no exchange %6.1f us per step  (%.2fx)" % (part.nelem, t_plain, t_whole / t_plain))
    print("... with the exchange in the stream (RCCL, own neighbour;  %6.1f us per step: the exchange + wall all-reduce cost %.1f us"
          % (t_rccl, t_rccl - t_plain))
    print("    %d + %d nodes and %d + %d element records per step)" % (len(nodes[0][0]), len(nodes[1][0]), len(elems[0][0]), len(elems[1][0])))
    print("... on the overlapped schedule (des_dev_set_overlap)        %6.1f us per step: %.1f us of the %.1f hidden"
          % (t_ov, t_rccl - t_ov, t_rccl - t_plain))
    print("host time to enqueue a step: %.0f us (no exchange), %.0f us (exchange in order), %.0f us (overlapped schedule)"
          % (HOST_US["plain"], HOST_US["rccl"], HOST_US["ov"]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
