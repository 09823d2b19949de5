#!/usr/bin/env python3
"""Step time of a decomposed-style engine (halo lists + RCCL communicator, the rank as its own
neighbour on the 1-GPU box) with the exchange overlap off and on, on the bench mesh."""
import ctypes as C, os, sys, time, types
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch.distributed as dist
import bench, dynearthsol_amd as des
from dynearthsol_amd._structs import DesHalo
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29812")
dist.init_process_group("gloo", rank=0, world_size=1)
host = des.Host(cfg_text=bench.BENCH_CFG.format(res=repr(400e3 / 560), xlen=repr(400e3)))
nn = host.nnode
k = 870
for overlap in (-1, 0, 1, -1, 0, 1):
    o0, o1 = k, nn - k
    from dynearthsol_amd._structs import DesMesh
    m = DesMesh.from_buffer_copy(host.mesh); m.owned_begin, m.owned_end = o0, o1
    eng = des.DeviceEngine(types.SimpleNamespace(params=host.params, mesh=m))
    recv = np.concatenate([np.arange(0, o0), np.arange(o1, nn)]).astype(np.int32)
    send = recv.copy()      # identity: the halo nodes keep their own values, the geometry stays regular
    nbr = np.zeros(2, np.int32); ptr = np.array([0, k, 2 * k], np.int32)
    pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    halo = DesHalo(o0, o1, 2 if overlap >= 0 else 0, pi(nbr), pi(ptr), pi(send), pi(ptr), pi(recv))   # -1: no exchanges at all
    eng.set_halo(types.SimpleNamespace(halo=halo, owned=(o0, o1), host=host))
    eng.comm_init(dist, 0, 1)
    eng.init_from_host(host)
    eng.set_overlap(overlap)
    eng.step(20, want_scalars=False); eng.sync()
    t = time.perf_counter()
    eng.step(200, want_scalars=False); eng.sync()
    print("overlap %d: %.4f ms/step" % (overlap, (time.perf_counter() - t) / 200 * 1e3), flush=True)
    del eng
dist.destroy_process_group()
