"""One-process-per-GPU runs of the whole program: ``torchrun --nproc-per-node N -m
dynearthsol_amd.distributed model.cfg``.

Every rank runs the same des_run() loop (include/des_run.h = the reference's main()) on the same
global host model, over an engine table whose entries are COLLECTIVE:

  create / upload   the rank's slab + ghost region of the global arrays (host partitioner)
  step, compute_dt  des_dev_step / des_dev_compute_dt of the rank's engine; the ghost-region
                    exchange and the dt reduction run inside them on RCCL
  download          owned nodes / owned elements of every rank gathered into the global array
  mesh_quality, check_nan   reduced over the ranks

so dt, time and step count are the same numbers everywhere, every rank takes the same decisions,
and rank 0 alone writes the frames (des_engine_api.no_files on the others).  torch.distributed
carries the gathers of an output frame and the 128-byte ncclUniqueId; nothing on the time-step
path.  Tests drive the same code over gloo with the CPU oracle as the engine.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import DesError, DeviceEngine, Host, load_host_lib
from . import driver
from ._structs import F, FIELDS, INT_FIELDS, DesScalars
from .decomp import Partition, PhasedStepper, TorchComm

_NODE3 = {"COORD", "VEL", "FORCE", "FORCE_RESIDUAL", "COORD0", "COORD_AVG0"}
_NODE1 = {"TEMPERATURE", "VOLUME_N", "MASS", "TMASS", "DHACC", "NTMP"}
_ELEM6 = {"STRESS", "STRAIN", "STRAIN_RATE", "STRESS_AVG", "STRAIN0"}
_ELEM1 = {"PLSTRAIN", "DELTA_PLSTRAIN", "VISCOSITY", "VOLUME", "VOLUME_OLD", "DPRESSURE", "EDVOLDT", "RADIOGENIC",
          "DPLSTRAIN_AVG", "STRESSYY"}


class CollectiveEngine:
    """The engine of one rank behind the collective entry points des_run() calls."""

    def __init__(self, host, dist, engine_factory, stepper=None):
        import torch
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        # host-side gathers / reductions of numpy data go over gloo whatever the main backend is
        self.group = None if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
        self.host = host
        self.ndims = int(host.params.ndims)
        self.nvec, self.nsym = (3, 6) if self.ndims == 3 else (2, 3)       # components of a nodal vector / a symmetric tensor
        self.part = Partition(host, self.world, self.rank)
        self.engine = engine_factory(self.part)
        self.engine.set_halo(self.part)
        # engines without a communicator of their own (the oracle in the tests) step in phases with
        # the ghost state moved by torch.distributed
        self.stepper = stepper(self.engine, self.part) if stepper else None
        p = self.part
        self.nn, self.ne = host.nnode, host.nelem
        o0, o1 = p.owned
        self.own_nodes_l = np.arange(o0, o1)
        self.own_elems_l = np.nonzero(p.elem_owned)[0]
        # global ids of what each rank owns, in rank order (the layout of a gathered array)
        self.g_nodes = self._allgather_ints(p.l2g_node[o0:o1])
        self.g_elems = self._allgather_ints(p.l2g_elem[self.own_elems_l])
        # surface lists are global-order subsets: top facets / top nodes this rank owns
        m, gm = p.mesh, host.mesh
        top_l = np.ctypeslib.as_array(m.top_nodes, shape=(m.ntop,)).copy() if m.ntop else np.zeros(0, np.int32)
        self.top_owned_l = np.nonzero((top_l >= o0) & (top_l < o1))[0]
        gtop = np.ctypeslib.as_array(gm.top_nodes, shape=(gm.ntop,)).copy() if gm.ntop else np.zeros(0, np.int32)
        pos = {int(n): i for i, n in enumerate(gtop)}
        self.g_top = self._allgather_ints(np.array([pos[int(p.l2g_node[top_l[i]])] for i in self.top_owned_l], np.int64))
        self.ntop_global = int(gm.ntop)
        # top facets: local -> global position (a facet belongs to the rank owning its element)
        ints = lambda ptr, n: np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, np.int32)
        fe_l, ff_l = ints(m.bfacet_elem[5], m.etop), ints(m.bfacet_facet[5], m.etop)
        fe_g, ff_g = ints(gm.bfacet_elem[5], gm.etop), ints(gm.bfacet_facet[5], gm.etop)
        fpos = {(int(e), int(f)): i for i, (e, f) in enumerate(zip(fe_g, ff_g))}
        self.fac_l2g = np.array([fpos[(int(p.l2g_elem[e]), int(f))] for e, f in zip(fe_l, ff_l)], np.int64)
        self.fac_owned_l = np.nonzero(p.elem_owned[fe_l])[0] if m.etop else np.zeros(0, np.int64)
        self.g_facets = self._allgather_ints(self.fac_l2g[self.fac_owned_l])

    # ---- helpers -----------------------------------------------------------------------
    def _allgather_ints(self, a):
        out = [None] * self.world
        self.dist.all_gather_object(out, np.asarray(a, dtype=np.int64), group=self.group)
        return out

    def _gather_rows(self, local_rows, gids, nglobal):
        """local_rows [ncomp, nlocal_owned] of every rank -> global [ncomp, nglobal]"""
        out = [None] * self.world
        self.dist.all_gather_object(out, np.ascontiguousarray(local_rows), group=self.group)
        res = np.empty((local_rows.shape[0], nglobal), dtype=local_rows.dtype)
        for ids, rows in zip(gids, out):
            res[:, ids] = rows
        return res

    # ---- the collective entry points -----------------------------------------------------
    def upload(self, name, glob):
        p = self.part
        if name in _NODE3:
            loc = glob.reshape(self.nvec, -1)[:, p.l2g_node]
        elif name in _NODE1:
            loc = glob[p.l2g_node]
        elif name in _ELEM6:
            loc = glob.reshape(self.nsym, -1)[:, p.l2g_elem]
        elif name in _ELEM1:
            loc = glob[p.l2g_elem]
        elif name == "ELEMMARKERS":
            loc = glob.reshape(self.ne, -1)[p.l2g_elem]
        elif name == "EDVACC_SURF":
            loc = glob[self.fac_l2g]
        else:
            raise DesError(60, "field %s cannot be scattered" % name)
        self.engine.upload(name, np.ascontiguousarray(loc).ravel())

    def download(self, name):
        a = self.engine.download(name)
        if name in _NODE3 or name in _NODE1:
            c = self.nvec if name in _NODE3 else 1
            return self._gather_rows(a.reshape(c, -1)[:, self.own_nodes_l], self.g_nodes, self.nn).ravel()
        if name in _ELEM6 or name in _ELEM1:
            c = self.nsym if name in _ELEM6 else 1
            return self._gather_rows(a.reshape(c, -1)[:, self.own_elems_l], self.g_elems, self.ne).ravel()
        if name == "ELEMMARKERS":
            rows = a.reshape(self.part.nelem, -1)[self.own_elems_l].T
            return np.ascontiguousarray(self._gather_rows(rows, self.g_elems, self.ne).T).ravel()
        if name == "DH":
            return self._gather_rows(a[self.top_owned_l][None, :], self.g_top, self.ntop_global).ravel()
        if name == "EDVACC_SURF":
            return self._gather_rows(a[self.fac_owned_l][None, :], self.g_facets, int(self.host.mesh.etop)).ravel()
        raise DesError(60, "field %s cannot be gathered" % name)

    def step(self, n):
        if self.stepper is not None:
            self.stepper.step(n)
            sc = self.engine.step(0)
        else:
            sc = self.engine.step(n)
        t = self.torch.tensor([sc.max_surf_vel, sc.max_global_vel_mag], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        sc.max_surf_vel, sc.max_global_vel_mag = float(t[0]), float(t[1])
        return sc

    def compute_dt(self):
        if self.stepper is not None:
            return self.stepper.comm.reduce_dt(self.engine, recompute=True)
        return self.engine.compute_dt()

    def mesh_quality(self, smallest_vol, bottom, bottom_dist):
        eng = self.engine
        q = driver.DesQuality()
        f = getattr(eng._lib, eng.prefix + "_mesh_quality")
        f.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.POINTER(driver.DesQuality)]
        rc = f(eng._h, smallest_vol, bottom, bottom_dist, C.byref(q))
        if rc:
            raise DesError(rc, "mesh_quality")
        p = self.part
        big = 2 ** 62
        small = int(p.l2g_elem[q.small_elem]) if q.small_elem >= 0 else big
        bnode = int(p.l2g_node[q.bottom_node]) if q.bottom_node >= 0 else big
        mine = (q.worst_quality, int(p.l2g_elem[q.worst_elem]), small, bnode)
        alls = [None] * self.world
        self.dist.all_gather_object(alls, mine, group=self.group)
        # the first element / node in GLOBAL numbering, the worst quality with the lowest index
        small, bnode = min(a[2] for a in alls), min(a[3] for a in alls)
        wq, we = min((a[0], a[1]) for a in alls)
        return (small if small < big else -1, bnode if bnode < big else -1, we, wq)

    def check_nan(self):
        t = self.torch.tensor([self.engine.check_nan()], dtype=self.torch.int64)
        self.dist.all_reduce(t, group=self.group)
        return int(t[0])


def collective_api(ce):
    """des_engine_api whose entries call the CollectiveEngine `ce` (ctypes callbacks)."""
    _vp = C.c_void_p

    def guard(fn):
        def wrapped(*a):
            try:
                return fn(*a)
            except DesError as e:
                sys.stderr.write("%s\n" % e)
                return e.code
            except Exception as e:                 # a Python error must not unwind through C
                sys.stderr.write("collective engine: %r\n" % (e,))
                return 60
        return wrapped

    def view(ptr, count, dtype):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int if dtype == np.int32 else C.c_double)), shape=(count,))

    @driver.CREATE_T
    def create(device, params, mesh, err):
        return 1                                           # the engine exists already (ce.engine)

    @driver.DESTROY_T
    def destroy(h):
        return None

    @driver.UPLOAD_T
    @guard
    def upload(h, field, ptr, count):
        name = FIELDS[field]
        ce.upload(name, view(ptr, count, np.int32 if name in INT_FIELDS else np.float64).copy())
        return 0

    @driver.DOWNLOAD_T
    @guard
    def download(h, field, ptr, count):
        name = FIELDS[field]
        view(ptr, count, np.int32 if name in INT_FIELDS else np.float64)[:] = ce.download(name)
        return 0

    @driver.COUNT_T
    def field_count(h, field):
        name = FIELDS[field]
        if name in _NODE3: return ce.nvec * ce.nn
        if name in _NODE1: return ce.nn
        if name in _ELEM6: return ce.nsym * ce.ne
        if name in _ELEM1: return ce.ne
        if name == "ELEMMARKERS": return ce.ne * ce.host.params.nmat
        if name == "DH": return ce.ntop_global
        if name == "EDVACC_SURF": return int(ce.host.mesh.etop)
        return -1

    @driver.CLOCK_T
    @guard
    def set_clock(h, dt, time, steps):
        ce.engine.set_clock(dt, time, steps)
        return 0

    @driver.INITGEOM_T
    @guard
    def init_geometry(h):
        if ce.ndims == 2 and ce.stepper is not None:
            ce.stepper.comm.reduce_wall(ce.engine)         # apply_vbcs reads the whole mesh's x0 wall (des_dev.h)
        ce.engine.init_geometry()
        return 0

    @driver.DT_T
    @guard
    def compute_dt(h, dt):
        v = ce.compute_dt()
        if dt:
            dt[0] = v
        return 0

    @driver.STEP_T
    @guard
    def step(h, n, out):
        sc = ce.step(n)
        if out:
            C.memmove(out, C.byref(sc), C.sizeof(DesScalars))
        return sc.status

    @driver.ISO_T
    @guard
    def set_isostasy(h, on):
        ce.engine.set_isostasy(on)
        return 0

    @driver.NAN_T
    @guard
    def check_nan(h, n_nan):
        n = ce.check_nan()
        if n_nan:
            n_nan[0] = n
        return 50 if n else 0

    @driver.QUALITY_T
    @guard
    def mesh_quality(h, smallest_vol, bottom, bottom_dist, out):
        small, bnode, we, wq = ce.mesh_quality(smallest_vol, bottom, bottom_dist)
        out[0].small_elem, out[0].bottom_node, out[0].worst_elem, out[0].worst_quality = small, bnode, we, wq
        return 0

    api = driver.EngineApi()
    api.create, api.destroy, api.upload, api.download, api.field_count = create, destroy, upload, download, field_count
    api.set_clock, api.init_geometry, api.compute_dt, api.step = set_clock, init_geometry, compute_dt, step
    api.check_nan, api.mesh_quality, api.set_isostasy = check_nan, mesh_quality, set_isostasy
    api.no_files = 0 if ce.rank == 0 else 1
    api._keep = (create, destroy, upload, download, field_count, set_clock, init_geometry, compute_dt, step,
                 check_nan, mesh_quality, set_isostasy)
    return api


def run_distributed(host, dist, engine_factory=None, stepper=None, quiet=True):
    """des_run() on every rank of `dist`; returns RunStats (identical on all ranks)."""
    if engine_factory is None:
        import torch
        local = int(os.environ.get("LOCAL_RANK", dist.get_rank() % max(1, torch.cuda.device_count())))

        def engine_factory(part):
            eng = DeviceEngine(part, device=local)
            return eng
    two_d = int(host.params.ndims) == 2
    # RCCL wants one device per rank: several 2-D ranks on ONE GPU (tests, rehearsals; DES_2D_TRANSPORT=host) step in two
    # phases with the ghost records and the two small reductions moved by torch.distributed instead
    callers_stepper = stepper is not None              # the caller moves the ghost state itself (tests: PhasedStepper over gloo)
    staged = two_d and stepper is None and os.environ.get("DES_2D_TRANSPORT", "rccl" if dist.get_backend() == "nccl" else "host") == "host"
    if staged:
        comm = TorchComm(dist, group=None if dist.get_backend() == "gloo" else dist.new_group(backend="gloo"))
        stepper = lambda e, p: PhasedStepper(e, p, comm)
    ce = CollectiveEngine(host, dist, engine_factory, stepper)
    if isinstance(ce.engine, DeviceEngine) and not staged and not callers_stepper:
        ce.engine.comm_init(dist, ce.rank, ce.world)
    api = collective_api(ce)
    return driver.run(host, quiet=quiet or ce.rank != 0, api=api)


def run_distributed_with_remesher(make_host, remesher, dist, engine_factory=None, stepper=None, quiet=True, max_rounds=100):
    """The remeshing round trip of include/des_run.h on N ranks (driver.run_with_remesher is the one-process form): where the
    loop stops for a remesh -- every rank takes that decision from the same reduced mesh-quality numbers, rank 0 has written
    the frame + checkpoint -- rank 0 runs `remesher(modelname, frame)` (a callable, or a command run as `<command> <modelname>
    <frame>`), which must leave the remeshed model as frame + 1; then EVERY rank restarts from that pair: a new host model, a
    new partition of the new mesh (node and element counts may have changed), new engines, the clock / frame numbering /
    .info continued.  `make_host(overrides)`: overrides = None for the first round, the restart keys afterwards.
    Returns the list of RunStats, one per mesh."""
    import subprocess
    stats, overrides = [], None
    for _ in range(max_rounds):
        host = make_host(overrides)
        model = host.cfg_string("sim.modelname")
        st = run_distributed(host, dist, engine_factory=engine_factory, stepper=stepper, quiet=quiet)
        host.close()
        stats.append(st)
        if not st.remesh_needed:
            return stats
        # rank 0 remeshes; EVERY rank then learns how that went (a bare barrier would leave the others waiting for the
        # backend's timeout while rank 0 is long gone) and they all raise together
        why = ""
        if dist.get_rank() == 0:
            try:
                if callable(remesher):
                    remesher(model, st.last_frame)
                else:
                    subprocess.check_call("%s %s %d" % (remesher, model, st.last_frame), shell=True)
            except Exception as e:          # noqa: BLE001 -- whatever the tool raised is reported on every rank
                why = "%s: %s" % (type(e).__name__, e)
        box = [why]
        dist.broadcast_object_list(box, src=0)
        if box[0]:
            raise DesError(21, "the remesher failed on rank 0 for %s frame %d (%s)" % (model, st.last_frame, box[0]))
        overrides = ("sim.is_restarting = yes\nsim.restarting_from_modelname = %s\nsim.restarting_from_frame = %d\n"
                     % (model, st.last_frame + 1))
    raise DesError(31, "the mesh needed remeshing more than %d times" % max_rounds)


def main(argv=None):
    import torch
    import torch.distributed as dist
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        sys.stderr.write("usage: torchrun --nproc-per-node N -m dynearthsol_amd.distributed [--ndims 2|3] [--remesher CMD] config.cfg [mesh.desmesh]\n")
        return 1
    ndims, remesher = 3, os.environ.get("DES_REMESH_CMD")
    while argv and argv[0] in ("--ndims", "--remesher"):
        if argv[0] == "--ndims":
            ndims = int(argv[1])
        else:
            remesher = argv[1]
        argv = argv[2:]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    try:
        if remesher:
            # (the restart reads the mesh from the frame the remesher left: the mesh file only serves the first round)
            make_host = lambda ov: Host(cfg_path=argv[0], overrides=ov, mesh_file=(argv[1] if len(argv) > 1 and not ov else None), ndims=ndims)
            return run_distributed_with_remesher(make_host, remesher, dist, quiet=False)[-1].exit_code
        host = Host(cfg_path=argv[0], mesh_file=argv[1] if len(argv) > 1 else None, ndims=ndims)
        st = run_distributed(host, dist, quiet=False)
        return st.exit_code
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
