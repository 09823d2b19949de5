#!/usr/bin/env python3
"""Cost of one halo exchange (pack kernel + grouped RCCL send/recv + unpack kernel) on the engine's
stream, measured with the rank as its own neighbour (the 1-GPU box cannot hold two RCCL ranks):
a lower bound for the launch/synchronisation overhead of the real thing."""
import ctypes as C, os, sys, time, types
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch.distributed as dist
import cfgs, dynearthsol_amd as des
from dynearthsol_amd._structs import DesHalo
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29811")
dist.init_process_group("gloo", rank=0, world_size=1)
host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, lx=80e3, ly=20e3, lz=10e3, res=1e3)))
eng = des.DeviceEngine(host)
nn = host.nnode
k = int(sys.argv[1]) if len(sys.argv) > 1 else 435
perm = np.random.default_rng(1).permutation(nn).astype(np.int32)
send, recv = np.sort(perm[:2*k]), np.sort(perm[2*k:4*k])
nbr = np.zeros(2, np.int32); ptr = np.array([0, k, 2*k], np.int32)
pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
halo = DesHalo(0, nn, 2, pi(nbr), pi(ptr), pi(send), pi(ptr), pi(recv))
eng.set_halo(types.SimpleNamespace(halo=halo, owned=(0, nn), host=host))
eng.comm_init(dist, 0, 1)
eng.init_from_host(host)
for kind in (0, 1, 2, 3):
    for _ in range(20): eng.exchange(kind)
    eng.sync()
    t = time.perf_counter()
    for _ in range(200): eng.exchange(kind)
    eng.sync()
    print("exchange kind %d (%d nodes to each of 2 neighbours): %.1f us" % (kind, k, (time.perf_counter() - t) / 200 * 1e6))
dist.destroy_process_group()
