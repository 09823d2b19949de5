"""Domain decomposition glue: slab partition of the host mesh (libdes_host.so), phased
stepping with halo exchanges, and the transports that carry them.

The data path of a production multi-GPU run does NOT go through this module: the device
engine exchanges halos itself with RCCL on its own stream (des_dev_comm_init /
des_dev_step).  What lives here is
  * `Partition` -- ctypes view of des_host_partition();
  * `init_rank` -- init() of one rank's engine from the global host model;
  * `PhasedStepper` + `LoopbackComm` / `TorchComm` -- the same four-phase step driven from
    Python with host-mediated exchanges, used by the tests (CPU oracle over gloo, several
    device engines on one GPU) to check the partition and the exchange lists.
"""
import ctypes as C

import numpy as np

from . import load_host_lib, DesError, DesMesh, F
from ._structs import DesHalo

X_WIDTH = (2, 1, 6, 2)        # DES_X_TEMP_NTMP, DES_X_NTMP, DES_X_VEL_COORD, DES_X_SURFACE (des_params.h)
NODAL = {"coord": 3, "vel": 3, "temperature": 1}
ELEMENTAL = {"stress": 6, "strain": 6, "plstrain": 1, "viscosity": 1, "radiogenic": 1}


class Partition:
    """One rank's part of the global mesh (des_host_partition)."""

    def __init__(self, host, nranks, rank):
        lib = load_host_lib()
        lib.des_host_partition.restype = C.c_void_p
        lib.des_host_partition.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        lib.des_part_destroy.argtypes = [C.c_void_p]
        lib.des_part_mesh.restype = C.POINTER(DesMesh)
        lib.des_part_mesh.argtypes = [C.c_void_p]
        lib.des_part_halo.restype = C.POINTER(DesHalo)
        lib.des_part_halo.argtypes = [C.c_void_p]
        for f in ("l2g_node", "l2g_elem", "node_ranges"):
            getattr(lib, "des_part_" + f).restype = C.POINTER(C.c_int)
            getattr(lib, "des_part_" + f).argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        err = C.c_int(0)
        p = lib.des_host_partition(host._h, nranks, rank, C.byref(err))
        if not p:
            raise DesError(err.value, lib.des_host_last_error().decode())
        self._lib, self._p = lib, C.c_void_p(p)
        self.host, self.nranks, self.rank = host, nranks, rank
        self.mesh = lib.des_part_mesh(self._p).contents
        self.halo = lib.des_part_halo(self._p).contents
        self.params = host.params
        self.nnode, self.nelem = self.mesh.nnode, self.mesh.nelem

        def ints(fn):
            n = C.c_int(0)
            ptr = fn(self._p, C.byref(n))
            return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()
        self.l2g_node = ints(lib.des_part_l2g_node)
        self.l2g_elem = ints(lib.des_part_l2g_elem)
        self.node_ranges = ints(lib.des_part_node_ranges)
        h = self.halo
        self.owned = (h.owned_begin, h.owned_end)
        nn = h.nnbr
        self.nbr_rank = [h.nbr_rank[i] for i in range(nn)]
        sp = [h.send_ptr[i] for i in range(nn + 1)]
        rp = [h.recv_ptr[i] for i in range(nn + 1)]
        self.send_idx = [np.array([h.send_idx[k] for k in range(sp[i], sp[i + 1])], dtype=np.int32) for i in range(nn)]
        self.recv_idx = [np.array([h.recv_idx[k] for k in range(rp[i], rp[i + 1])], dtype=np.int32) for i in range(nn)]

    def local(self, name):
        """This rank's slice of a global host array, in the reference's SoA layout."""
        a = self.host.array(name)
        if name in NODAL:
            c = NODAL[name]
            return np.ascontiguousarray(a.reshape(c, -1)[:, self.l2g_node]).ravel()
        if name in ELEMENTAL:
            c = ELEMENTAL[name]
            return np.ascontiguousarray(a.reshape(c, -1)[:, self.l2g_elem]).ravel()
        if name == "elemmarkers":
            nmat = self.params.nmat
            return np.ascontiguousarray(a.reshape(-1, nmat)[self.l2g_elem]).ravel()
        raise KeyError(name)

    def close(self):
        if self._p:
            self._lib.des_part_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def init_rank(engine, part, comm):
    """init() + first compute_dt of main() for one rank (dynearthsol.cxx:175-221, 643)."""
    engine.set_halo(part)
    coord = part.local("coord")
    engine.upload("COORD", coord)
    engine.upload("COORD0", coord)
    engine.upload("ELEMMARKERS", part.local("elemmarkers"))
    engine.upload("VEL", part.local("vel"))
    engine.init_geometry()
    for f, name in (("TEMPERATURE", "temperature"), ("RADIOGENIC", "radiogenic"), ("STRESS", "stress"),
                    ("STRAIN", "strain"), ("PLSTRAIN", "plstrain"), ("VISCOSITY", "viscosity")):
        engine.upload(f, part.local(name))
    return comm.reduce_dt(engine, recompute=True)


class PhasedStepper:
    """Five phases per step with the halo exchanges in between (DES_X_* of des_params.h)."""

    def __init__(self, engine, part, comm):
        self.engine, self.part, self.comm = engine, part, comm

    def step(self, nsteps):
        e = self.engine
        for _ in range(nsteps):
            e.phase(0)
            self.comm.exchange(self, 0)
            e.phase(1)
            if self.part.params.is_using_mixed_stress:
                self.comm.exchange(self, 1)
            e.phase(2)
            self.comm.exchange(self, 2)
            e.phase(3)
            if self.part.params.has_moving_mesh and self.part.params.surface_process_option == 1:
                self.comm.exchange(self, 3)
            if e.phase(4):
                self.comm.reduce_dt(e, recompute=False)


class LoopbackComm:
    """All ranks live in this process (tests): phases run rank after rank, exchanges are
    plain array copies.  Use through `run_loopback`."""

    def __init__(self, steppers):
        self.steppers = steppers

    def exchange_all(self, kind):
        w = X_WIDTH[kind]
        boxes = {}
        for st in self.steppers:
            p = st.part
            for q, idx in zip(p.nbr_rank, p.send_idx):
                boxes[(p.rank, q)] = st.engine.halo_pack(kind, idx, w)
        for st in self.steppers:
            p = st.part
            for q, idx in zip(p.nbr_rank, p.recv_idx):
                st.engine.halo_unpack(kind, idx, boxes[(q, p.rank)])

    def reduce_dt_all(self, recompute):
        parts = np.array([st.engine.dt_partials(recompute) for st in self.steppers])
        red = parts.min(axis=0)
        return [st.engine.dt_finalize(red) for st in self.steppers]


def run_loopback(steppers, nsteps):
    comm = LoopbackComm(steppers)
    prm = steppers[0].part.params
    nmd = prm.is_using_mixed_stress
    surf = prm.has_moving_mesh and prm.surface_process_option == 1
    for _ in range(nsteps):
        for st in steppers: st.engine.phase(0)
        comm.exchange_all(0)
        for st in steppers: st.engine.phase(1)
        if nmd:
            comm.exchange_all(1)
        for st in steppers: st.engine.phase(2)
        comm.exchange_all(2)
        for st in steppers: st.engine.phase(3)
        if surf:
            comm.exchange_all(3)
        flags = [st.engine.phase(4) for st in steppers]
        if any(flags):
            comm.reduce_dt_all(recompute=False)


class TorchComm:
    """One rank per process over torch.distributed (gloo on CPU in the tests)."""

    def __init__(self, dist, device="cpu"):
        import torch
        self.dist, self.torch, self.device = dist, torch, device

    def exchange(self, stepper, kind):
        torch, dist = self.torch, self.dist
        p, e, w = stepper.part, stepper.engine, X_WIDTH[kind]
        reqs, recvs = [], []
        for q, sidx, ridx in zip(p.nbr_rank, p.send_idx, p.recv_idx):
            sbuf = torch.from_numpy(e.halo_pack(kind, sidx, w))
            rbuf = torch.empty(len(ridx) * w, dtype=torch.float64)
            reqs.append(dist.isend(sbuf, dst=q))
            reqs.append(dist.irecv(rbuf, src=q))
            recvs.append((ridx, rbuf, sbuf))
        for r in reqs:
            r.wait()
        for ridx, rbuf, _ in recvs:
            e.halo_unpack(kind, ridx, rbuf.numpy())

    def reduce_dt(self, engine, recompute):
        t = self.torch.from_numpy(engine.dt_partials(recompute))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return engine.dt_finalize(t.numpy())


def assemble(parts, locals_, ncomp, nglobal, kind):
    """Global SoA array from per-rank local arrays: owned nodes / every local element (overlap
    elements are computed identically on both sides, the assembly checks that)."""
    out = np.full((ncomp, nglobal), np.nan)
    for p, a in zip(parts, locals_):
        a = a.reshape(ncomp, -1)
        if kind == "node":
            o0, o1 = p.owned
            out[:, p.l2g_node[o0:o1]] = a[:, o0:o1]
        else:
            prev = out[:, p.l2g_elem]
            same = np.isnan(prev) | (prev == a)
            if not same.all():
                raise AssertionError("overlap elements differ between ranks")
            out[:, p.l2g_elem] = a
    return out.ravel()
