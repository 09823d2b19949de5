#!/usr/bin/env python3
"""Cost of the ghost-region exchange of a step (pack kernel + grouped RCCL send/recv + unpack
kernel) on the engine's stream, measured with the rank as its own neighbour (the 1-GPU box cannot
hold two RCCL ranks) at the list sizes of the 1.1M-tet bench slab: per side 4 layers of ghost
nodes and 2 layers of elements.  A lower bound for the launch / synchronisation overhead of the
real thing, and the step time with and without it."""
import ctypes as C, os, sys, time, types
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch.distributed as dist
import bench, dynearthsol_amd as des
from dynearthsol_amd._structs import DesHalo
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29811")
dist.init_process_group("gloo", rank=0, world_size=1)
host = des.Host(cfg_text=bench.BENCH_CFG.format(res=repr(400e3 / 560), xlen=repr(400e3)))
nn, ne = host.nnode, host.nelem
k, ke = 4 * 435, 2 * 1960
rng = np.random.default_rng(1)
# identity lists (send = recv): the values land where they came from, the run stays regular
nodes = np.sort(rng.choice(nn, 2 * k, replace=False)).astype(np.int32)
elems = np.sort(rng.choice(ne, 2 * ke, replace=False)).astype(np.int32)
nbr = np.zeros(2, np.int32); ptr = np.array([0, k, 2 * k], np.int32); eptr = np.array([0, ke, 2 * ke], np.int32)
pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
for with_exchange in (0, 1, 0, 1):
    eng = des.DeviceEngine(host)
    halo = DesHalo(0, nn, 4, 2 if with_exchange else 0, pi(nbr), pi(ptr), pi(nodes), pi(ptr), pi(nodes),
                   pi(eptr), pi(elems), pi(eptr), pi(elems))
    eng.set_halo(types.SimpleNamespace(halo=halo, owned=(0, nn), host=host))
    eng.comm_init(dist, 0, 1)
    eng.init_from_host(host)
    if with_exchange:
        for _ in range(20): eng.exchange()
        eng.sync(); t = time.perf_counter()
        for _ in range(200): eng.exchange()
        eng.sync()
        print("exchange alone (%d nodes + %d elements per neighbour, 2 neighbours): %.1f us"
              % (k, ke, (time.perf_counter() - t) / 200 * 1e6), flush=True)
    eng.step(20, want_scalars=False); eng.sync()
    t = time.perf_counter()
    eng.step(200, want_scalars=False); eng.sync()
    print("step %s exchange: %.4f ms" % ("with" if with_exchange else "without", (time.perf_counter() - t) / 200 * 1e3), flush=True)
    del eng
dist.destroy_process_group()
