"""Domain decomposition glue: slab partition of the host mesh (libdes_host.so), phased
stepping with halo exchanges, and the transports that carry them.

The data path of a production multi-GPU run does NOT go through this module: the device
engine exchanges halos itself with RCCL on its own stream (des_dev_comm_init /
des_dev_step).  What lives here is
  * `Partition` -- ctypes view of des_host_partition();
  * `init_rank` -- init() of one rank's engine from the global host model;
  * `PhasedStepper` + `LoopbackComm` / `TorchComm` -- the same two-phase step driven from
    Python with host-mediated exchanges, used by the tests (CPU oracle over gloo, several
    device engines on one GPU) to check the partition and the exchange lists.
"""
import ctypes as C

import numpy as np

from . import load_host_lib, DesError, DesMesh, F
from ._structs import DesHalo

NODE_WIDTH, ELEM_WIDTH = 8, 13     # DES_X_NODE_WIDTH, DES_X_ELEM_WIDTH (des_params.h)
WIDTHS = {3: (NODE_WIDTH, ELEM_WIDTH), 2: (6, 8)}      # ... and DES_X_NODE_WIDTH_2D, DES_X_ELEM_WIDTH_2D
NODAL = {"coord": 3, "vel": 3, "temperature": 1}
ELEMENTAL = {"stress": 6, "strain": 6, "plstrain": 1, "viscosity": 1, "radiogenic": 1}
NODAL_2D = {"coord": 2, "vel": 2, "temperature": 1}
ELEMENTAL_2D = {"stress": 3, "strain": 3, "plstrain": 1, "viscosity": 1, "radiogenic": 1, "stressyy": 1}


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
        for f in ("l2g_node", "l2g_elem", "node_ranges", "elem_owned"):
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
        self.ndims = int(host.params.ndims)
        self.node_width, self.elem_width = WIDTHS[self.ndims]
        self.nnode, self.nelem = self.mesh.nnode, self.mesh.nelem

        def ints(fn):
            n = C.c_int(0)
            ptr = fn(self._p, C.byref(n))
            return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()
        self.l2g_node = ints(lib.des_part_l2g_node)
        self.l2g_elem = ints(lib.des_part_l2g_elem)
        self.node_ranges = ints(lib.des_part_node_ranges)
        self.elem_owned = ints(lib.des_part_elem_owned).astype(bool)
        h = self.halo
        self.owned = (h.owned_begin, h.owned_end)
        nn = h.nnbr
        self.nbr_rank = [h.nbr_rank[i] for i in range(nn)]

        def lists(ptr, idx):
            p = np.ctypeslib.as_array(ptr, shape=(nn + 1,)) if nn else np.zeros(1, np.int32)
            a = np.ctypeslib.as_array(idx, shape=(int(p[nn]),)).copy() if nn and p[nn] else np.zeros(0, np.int32)
            return [a[p[i]:p[i + 1]].astype(np.int32) for i in range(nn)]
        self.send_idx, self.recv_idx = lists(h.send_ptr, h.send_idx), lists(h.recv_ptr, h.recv_idx)
        self.esend_idx, self.erecv_idx = lists(h.esend_ptr, h.esend_idx), lists(h.erecv_ptr, h.erecv_idx)

    def local(self, name):
        """This rank's slice of a global host array, in the reference's SoA layout."""
        a = self.host.array(name)
        nodal, elemental = (NODAL_2D, ELEMENTAL_2D) if self.ndims == 2 else (NODAL, ELEMENTAL)
        if name in nodal:
            c = nodal[name]
            return np.ascontiguousarray(a.reshape(c, -1)[:, self.l2g_node]).ravel()
        if name in elemental:
            c = elemental[name]
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


def init_rank_mesh(engine, part):
    engine.set_halo(part)
    coord = part.local("coord")
    engine.upload("COORD", coord)
    engine.upload("COORD0", coord)
    engine.upload("ELEMMARKERS", part.local("elemmarkers"))
    engine.upload("VEL", part.local("vel"))


def init_rank_fields(engine, part):
    engine.init_geometry()
    for f, name in (("TEMPERATURE", "temperature"), ("RADIOGENIC", "radiogenic"), ("STRESS", "stress"),
                    ("STRAIN", "strain"), ("PLSTRAIN", "plstrain"), ("VISCOSITY", "viscosity")):
        engine.upload(f, part.local(name))
    if part.ndims == 2:
        engine.upload("STRESSYY", part.local("stressyy"))


def init_rank(engine, part, comm):
    """init() + first compute_dt of main() for one rank (dynearthsol.cxx:175-221, 643)."""
    init_rank_mesh(engine, part)
    if part.ndims == 2:
        comm.reduce_wall(engine)        # apply_vbcs inside init_geometry reads the whole mesh's x0 wall (des_dev.h)
    init_rank_fields(engine, part)
    return comm.reduce_dt(engine, recompute=True)


def res_nblocks(nnode_global):
    """blocks of the partition-independent residual (des_params.h: des_res_block)"""
    b = 64 if nnode_global >= 64 * 64 else 1
    return (nnode_global + b - 1) // b


def pt_converged(l2, l2_old, tol):
    """the reference's test (dynearthsol.cxx:829-833), on the value every rank holds"""
    return abs((l2 - l2_old) / l2_old) < tol


class PhasedStepper:
    """The two phases of a step with the ghost-region exchange in between (des_params.h).  With control.has_PT phase 0
    stops in front of the pseudo-transient loop (returns 2) and the loop runs here: ghost region refreshed, one iteration
    (phase 2), the residual's block partials put together across ranks, the reference's convergence test; phase 3 is the
    rest of phase 0."""

    def __init__(self, engine, part, comm):
        self.engine, self.part, self.comm = engine, part, comm
        self.n_pt_iterations = 0

    def pt_loop(self):
        e, p = self.engine, self.part
        par = p.params
        l2_old = self.comm.residual(self)
        for _ in range(int(par.PT_max_iter)):
            self.comm.exchange(self)
            if p.ndims == 2:
                self.comm.reduce_wall(e)
            e.phase(2)
            l2 = self.comm.residual(self)
            self.n_pt_iterations += 1
            if pt_converged(l2, l2_old, par.PT_relative_tolerance):
                break
            l2_old = l2
        e.phase(3)

    def step(self, nsteps):
        e = self.engine
        self.n_pt_iterations = 0
        for _ in range(nsteps):
            if e.phase(0) == 2:
                self.pt_loop()
            self.comm.exchange(self)
            if self.part.ndims == 2:
                self.comm.reduce_wall(e)
            if e.phase(1):
                self.comm.reduce_dt(e, recompute=False)


class LoopbackComm:
    """All ranks live in this process (tests): phases run rank after rank, exchanges are
    plain array copies.  Use through `run_loopback`."""

    def __init__(self, steppers):
        self.steppers = steppers

    def exchange_all(self):
        boxes = {}
        for st in self.steppers:
            p = st.part
            for q, idx, eidx in zip(p.nbr_rank, p.send_idx, p.esend_idx):
                boxes[(p.rank, q)] = (st.engine.halo_pack(0, idx, p.node_width), st.engine.halo_pack(1, eidx, p.elem_width))
        for st in self.steppers:
            p = st.part
            for q, idx, eidx in zip(p.nbr_rank, p.recv_idx, p.erecv_idx):
                nbuf, ebuf = boxes[(q, p.rank)]
                st.engine.halo_unpack(0, idx, nbuf)
                st.engine.halo_unpack(1, eidx, ebuf)

    def residual_all(self):
        """every rank's block partials into the global array, then the fixed-shape sum on every rank: one value"""
        blocks = np.zeros(res_nblocks(self.steppers[0].part.host.nnode))
        for st in self.steppers:
            first, vals = st.engine.residual_blocks()
            blocks[first:first + len(vals)] = vals
        l2 = [st.engine.residual_set(blocks) for st in self.steppers]
        assert all(x == l2[0] for x in l2)
        return l2[0]

    def reduce_wall_all(self):
        red = np.array([st.engine.wall_get() for st in self.steppers]).max(axis=0)
        for st in self.steppers:
            st.engine.wall_set(red)

    def reduce_dt_all(self, recompute):
        parts = np.array([st.engine.dt_partials(recompute) for st in self.steppers])
        red = parts.min(axis=0)
        return [st.engine.dt_finalize(red) for st in self.steppers]


def run_loopback(steppers, nsteps):
    comm = LoopbackComm(steppers)
    two_d = steppers[0].part.ndims == 2
    par = steppers[0].part.params
    for st in steppers: st.n_pt_iterations = 0
    for _ in range(nsteps):
        flags = [st.engine.phase(0) for st in steppers]
        if any(f == 2 for f in flags):               # the pseudo-transient loop, all ranks in step (PhasedStepper.pt_loop)
            assert all(f == 2 for f in flags)
            l2_old = comm.residual_all()
            for _ in range(int(par.PT_max_iter)):
                comm.exchange_all()
                if two_d:
                    comm.reduce_wall_all()
                for st in steppers: st.engine.phase(2)
                l2 = comm.residual_all()
                for st in steppers: st.n_pt_iterations += 1
                if pt_converged(l2, l2_old, par.PT_relative_tolerance):
                    break
                l2_old = l2
            for st in steppers: st.engine.phase(3)
        comm.exchange_all()
        if steppers[0].part.ndims == 2:
            comm.reduce_wall_all()
        flags = [st.engine.phase(1) for st in steppers]
        if any(flags):
            comm.reduce_dt_all(recompute=False)


class TorchComm:
    """One rank per process over torch.distributed (gloo on CPU in the tests)."""

    def __init__(self, dist, device="cpu", group=None):
        import torch
        self.dist, self.torch, self.device, self.group = dist, torch, device, group

    def exchange(self, stepper):
        torch, dist = self.torch, self.dist
        p, e = stepper.part, stepper.engine
        reqs, recvs = [], []
        for q, sidx, ridx, seidx, reidx in zip(p.nbr_rank, p.send_idx, p.recv_idx, p.esend_idx, p.erecv_idx):
            sbuf = torch.from_numpy(np.concatenate([e.halo_pack(0, sidx, p.node_width), e.halo_pack(1, seidx, p.elem_width)]))
            rbuf = torch.empty(len(ridx) * p.node_width + len(reidx) * p.elem_width, dtype=torch.float64)
            reqs.append(dist.isend(sbuf, dst=q, group=self.group))
            reqs.append(dist.irecv(rbuf, src=q, group=self.group))
            recvs.append((ridx, reidx, rbuf, sbuf))
        for r in reqs:
            r.wait()
        for ridx, reidx, rbuf, _ in recvs:
            a = rbuf.numpy()
            e.halo_unpack(0, ridx, a[:len(ridx) * p.node_width])
            e.halo_unpack(1, reidx, a[len(ridx) * p.node_width:])

    def reduce_wall(self, engine):
        t = self.torch.from_numpy(engine.wall_get())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        engine.wall_set(t.numpy())

    def residual(self, stepper):
        """the partition-independent residual: every block has one owner, so a SUM of the zero-filled global arrays puts the
        partials together exactly (x + 0 = x); the fixed-shape sum then runs on every rank"""
        first, vals = stepper.engine.residual_blocks()
        blocks = np.zeros(res_nblocks(stepper.part.host.nnode))
        blocks[first:first + len(vals)] = vals
        t = self.torch.from_numpy(blocks)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return stepper.engine.residual_set(t.numpy())

    def reduce_dt(self, engine, recompute):
        t = self.torch.from_numpy(engine.dt_partials(recompute))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return engine.dt_finalize(t.numpy())


class DeviceGroup:
    """`nranks` device engines of THIS process as the ranks of one decomposed model, stepped in lockstep by
    des_dev_step_group: every rank runs des_dev_step's own launches (the fused multi-step path included) and
    the ghost region changes hands by device-to-device copies -- the multi-GPU step at its real partition,
    rehearsed on one GPU.  (A production run has one process per GPU and RCCL in the same place.)"""

    def __init__(self, host, nranks, device=0):
        from . import DeviceEngine, DesScalars, load_hip_lib
        self.host, self.nranks = host, nranks
        self.parts = [Partition(host, nranks, r) for r in range(nranks)]
        self.engines = [DeviceEngine(p, device=device) for p in self.parts]
        self._lib = load_hip_lib()
        self._scalars = DesScalars
        for e, p in zip(self.engines, self.parts):
            e.set_halo(p)
        self._arr = (C.c_void_p * nranks)(*[e._h for e in self.engines])
        self._lib.des_dev_group_attach.argtypes = [C.c_void_p, C.c_int]
        self._lib.des_dev_group_detach.argtypes = [C.c_void_p, C.c_int]
        self._lib.des_dev_step_group.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        rc = self._lib.des_dev_group_attach(self._arr, nranks)
        if rc:
            raise DesError(rc, self._lib.des_dev_last_error().decode())
        self._attached = True

    def init_from_host(self):
        """init() + the first compute_dt on every rank; returns the (common) dt."""
        steppers = [PhasedStepper(e, p, None) for e, p in zip(self.engines, self.parts)]
        comm = LoopbackComm(steppers)

        for e, p in zip(self.engines, self.parts):
            init_rank_mesh(e, p)
        if self.parts[0].ndims == 2:
            comm.reduce_wall_all()
        for e, p in zip(self.engines, self.parts):
            init_rank_fields(e, p)
        dts = comm.reduce_dt_all(recompute=True)
        assert all(d == dts[0] for d in dts)
        return dts[0]

    def step(self, nsteps):
        out = (self._scalars * self.nranks)()
        rc = self._lib.des_dev_step_group(self._arr, self.nranks, nsteps, out)
        if rc:
            raise DesError(rc, self._lib.des_dev_last_error().decode())
        return list(out)

    def body_force_adjustment(self):
        """initial_body_force_adjustment (dynearthsol.cxx:546-591) on every rank in lockstep (3-D engines)"""
        out = (self._scalars * self.nranks)()
        self._lib.des_dev_body_force_adjustment_group.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        rc = self._lib.des_dev_body_force_adjustment_group(self._arr, self.nranks, out)
        if rc:
            raise DesError(rc, self._lib.des_dev_last_error().decode())
        return list(out)

    def upload(self, field, global_array, ncomp, kind):
        """every rank's slice of a global SoA array (nodal or elemental) into its engine"""
        a = np.asarray(global_array, dtype=np.float64).reshape(ncomp, -1)
        for e, p in zip(self.engines, self.parts):
            idx = p.l2g_node if kind == "node" else p.l2g_elem
            e.upload(field, np.ascontiguousarray(a[:, idx]))

    def download(self, field, ncomp, kind):
        """the global SoA array of `field`, assembled from every rank's owned nodes / elements"""
        n = self.host.nnode if kind == "node" else self.host.nelem
        return assemble(self.parts, [e.download(field) for e in self.engines], ncomp, n, kind)

    def close(self):
        if self._attached:
            self._lib.des_dev_group_detach(self._arr, self.nranks)
            self._attached = False
        for e in self.engines:
            e.close()
        for p in self.parts:
            p.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def assemble(parts, locals_, ncomp, nglobal, kind):
    """Global SoA array from per-rank local arrays: owned nodes / owned elements (every node and
    every element of the global mesh has exactly one owner)."""
    out = np.full((ncomp, nglobal), np.nan)
    for p, a in zip(parts, locals_):
        a = a.reshape(ncomp, -1)
        if kind == "node":
            o0, o1 = p.owned
            out[:, p.l2g_node[o0:o1]] = a[:, o0:o1]
        else:
            out[:, p.l2g_elem[p.elem_owned]] = a[:, p.elem_owned]
    assert not np.isnan(out).any(), "a node / element has no owner"
    return out.ravel()
