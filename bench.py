#!/usr/bin/env python3
"""bench.py -- explicit time-steps/s x #elements of the MI355X time-stepper.

    python bench.py --gpus N --steps K --warmup W

One "step" is one explicit time step (dynearthsol.cxx:768-894) of the whole mesh.  The
workload at N=1 is BASELINE.json configs[2] as SURVEY.md 8(d) pins it down: the
benchmarks-cores/test-3d-big.cfg box (400 x 20 x 10 km, elasto-visco-plastic variant) at
>= 1M tetrahedra.  The reference's TetGen mesher is a host-side library that cannot run on
the GPU box, so the mesh is the reference's own regular mesher (meshing_option = 1,
meshing_elem_shape = 1: 5 tets per grid cell, mesh.cxx:1431-1459), rebuilt by the host
library: 560 x 28 x 14 cells = 1,097,600 tets / 244,035 nodes.  Fields are the model's own
initial conditions (synthetic in the sense of: no input data files).

Prints ONE JSON line (rank 0).  `value` = elements x steps / s summed over all ranks, timed
with state resident in HBM; `roofline` is the dominant kernel's algorithmic bytes over its
measured HIP-event duration against 8 TB/s; `cpu_baseline` is the CPU oracle (OpenMP build,
kind "port") timed on this box's host cores on a bounded number of steps of the same mesh.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

BENCH_CFG = """
[sim]
modelname = bench
max_steps = 1000000
output_step_interval = 1000000
is_outputting_averaged_fields = no
[mesh]
meshing_option = 1
meshing_elem_shape = 1
xlength = {xlen}
ylength = 20e3
zlength = 10e3
resolution = {res}
quality_check_step_interval = 1000000
[control]
surface_process_option = 1
surface_diffusivity = 1e-6
dt_fraction = 1.0
inertial_scaling = 1e4
[bc]
vbc_x0 = 1
vbc_x1 = 1
vbc_val_x0 = -1e-9
vbc_val_x1 = 1e-9
vbc_y0 = 1
vbc_y1 = 1
vbc_val_y0 = 0
vbc_val_y1 = 0
has_water_loading = no
surface_temperature = 273
mantle_temperature = 1573
[ic]
oceanic_plate_age_in_yr = 1e6
weakzone_option = 1
weakzone_azimuth = 15
weakzone_inclination = -60
weakzone_halfwidth = 1.2
weakzone_depth_min = 0.5
weakzone_depth_max = 1.0
weakzone_xcenter = 0.5
weakzone_ycenter = 0.5
weakzone_zcenter = 0
weakzone_plstrain = 0.5
[mat]
rheology_type = elasto-visco-plastic
rho0 = [2700]
alpha = [3e-5]
bulk_modulus = [50e9]
shear_modulus = [30e9]
pls0 = [0]
pls1 = [0.5]
cohesion0 = [4.4e7]
cohesion1 = [4e6]
friction_angle0 = [30]
friction_angle1 = [30]
min_viscosity = 1e19
"""

# Algorithmic HBM bytes of each pass per element / per node (the
# six rows sum to B_alg = 1420*ne + 348*nn; evp adds 24*ne + 8*nn to E2).
KERNEL_BYTES = {
    "E1_geom_rotate_strainrate":  (364, 56),
    "N1_mass_temperature_dvoldt": (224, 64),
    "E2_update_stress":           (364, 8),
    "N2_nmd_gather":              (48, 20),
    "E3_nmd_force":               (260, 40),
    "N3_force_velocity_coord":    (160, 160),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--resolution", type=float, default=400e3 / 560)
    ap.add_argument("--cpu-steps", type=int, default=-1, help="steps of the CPU baseline (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: keep the 1.1M-tet mesh and cut it N ways (default: weak scaling, "
                         "the box grows N times in x so every GPU keeps 1.1M tets)")
    ap.add_argument("--mesh-file", default=None,
                    help="take the mesh from a .desmesh file (e.g. the reference's TetGen mesh of test-3d-big.cfg at "
                         "resolution 460 m, 1,001,310 tets, written by oracle/_ref/tetmesh) instead of the regular mesher; N=1")
    ap.add_argument("--workload", default="test-3d-big", choices=["test-3d-big", "test-3d-equ-long"],
                    help="test-3d-big (default, the BASELINE config) or the reference's 984,375-tet seven-material "
                         "regular-mesh benchmark benchmarks-cores/test-3d-equ-long.cfg (values restated in tests/cfgs.py); N=1")
    ap.add_argument("--averaged-fields", action="store_true",
                    help="also run Output::average_fields inside the step (sim.is_outputting_averaged_fields = yes)")
    ap.add_argument("--rheology", default="elasto-visco-plastic",
                    help="diagnostic only: the headline workload is elasto-visco-plastic")
    args = ap.parse_args()

    t_begin = time.perf_counter()
    if os.environ.get("DES_BENCH_VERBOSE"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("DES_BENCH_WATCHDOG", "60")), repeat=False)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DES_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend)

    import dynearthsol_amd as des

    def note(msg):
        if os.environ.get("DES_BENCH_VERBOSE"):
            sys.stderr.write("[bench rank %d %.1fs] %s\n" % (rank, time.perf_counter() - t_begin, msg)); sys.stderr.flush()

    # weak scaling: the test-3d-big box is repeated N times along x (same resolution), then cut
    # into N slabs of contiguous node ids -- every GPU holds ~1.1M tets plus its four-layer ghost region
    xlen = 400e3 * (1 if args.strong else world)
    overrides = "" if args.rheology == "elasto-visco-plastic" else "mat.rheology_type = %s\n" % args.rheology
    if args.averaged_fields:
        overrides += "sim.is_outputting_averaged_fields = yes\nmesh.quality_check_step_interval = 100\n"
    if args.mesh_file:
        assert world == 1, "--mesh-file is a single-GPU workload"
        overrides += "mesh.meshing_option = 2\nmesh.meshing_elem_shape = 0\n"
    note("building the host model")
    if args.workload == "test-3d-equ-long":
        assert world == 1 and not args.mesh_file, "--workload test-3d-equ-long is a single-GPU workload"
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import cfgs
        host = des.Host(cfg_text=cfgs.make_equ(long=True),
                        overrides=(overrides or "") + "sim.max_steps = 1000000\nsim.output_step_interval = 1000000\n")
    else:
        host = des.Host(cfg_text=BENCH_CFG.format(res=repr(args.resolution), xlen=repr(xlen)), overrides=overrides or None,
                        mesh_file=args.mesh_file)
    device = int(os.environ.get("DES_BENCH_DEVICE", local_rank))
    transport = "RCCL ncclSend/ncclRecv on the engine stream"
    if world == 1:
        dev = des.DeviceEngine(host, device=device)
        dev.init_from_host(host)
        ne_local = host.nelem
    else:
        from dynearthsol_amd.decomp import Partition, init_rank

        class _Comm:          # init only: the first compute_dt goes through the engine's own allreduce
            def reduce_dt(self, engine, recompute):
                return engine.compute_dt()
        note("partition")
        part = Partition(host, world, rank)
        note("engine")
        dev = des.DeviceEngine(part, device=device)
        dev.set_halo(part)
        # The ghost-region exchange runs inside des_dev_step on RCCL.  Should the engine's own
        # communicator fail to come up on this node, the same step is driven in its two phases
        # with the ghost state staged through the host over gloo -- slower, said so in the line.
        ok = 1
        try:
            if os.environ.get("DES_BENCH_TRANSPORT", "rccl") != "rccl":
                raise des.DesError(31, "host transport requested")
            dev.comm_init(dist, rank, world)
        except des.DesError as e:
            sys.stderr.write("rank %d: %s\n" % (rank, e))
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            init_rank(dev, part, _Comm())
        else:
            from dynearthsol_amd.decomp import PhasedStepper, TorchComm
            transport = "host-staged over gloo (engine communicator unavailable)"
            comm = TorchComm(dist, group=dist.new_group(backend="gloo"))
            init_rank(dev, part, comm)
            stepper = PhasedStepper(dev, part, comm)
            _step = dev.step
            dev.step = lambda n, want_scalars=True: (stepper.step(n), _step(0, want_scalars=want_scalars))[1]
        ne_local = part.nelem
    note("initialised; warm-up")
    ne, nn = host.nelem, host.nnode          # global counts: `value` counts every element once

    dev.step(args.warmup, want_scalars=False) if args.warmup > 0 else None
    dev.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        dev.sync()

    note("timed region")
    barrier()
    t0 = time.perf_counter()
    dev.timer_start()
    dev.step(args.steps, want_scalars=False)
    ev_ms = dev.timer_stop()
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    sc = dev.step(0)
    nan = dev.check_nan()

    value = float(ne) * args.steps / wall
    result = {
        "metric": "explicit time-steps/sec x #elements",
        "value": value,
        "unit": "element-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * wall / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": ("test-3d-equ-long.cfg box 250x50x125 km, 7 materials, " if args.workload == "test-3d-equ-long" else
                         "test-3d-big.cfg box 400x20x10 km, ") + args.rheology + ", thermal+NMD+surface diffusion on, "
                        + ("averaged output fields on, " if args.averaged_fields else "")
                        + ("mesh file %s, " % os.path.basename(args.mesh_file) if args.mesh_file else "regular 5-tet mesh, ")
                        +
                        "%d tets / %d nodes in total" % (ne, nn),
            "nelem": ne, "nnode": nn, "nelem_local_rank0": ne_local,
            "parallelism": "single GPU" if world == 1 else
                           "%d slabs of contiguous node ids, four-layer ghost region, one exchange per step; transport: %s" % (world, transport),
            "steps_per_s": args.steps / wall,
            "hip_event_ms_per_step": ev_ms / args.steps,
            "nan_entries": nan, "status": sc.status,
        },
    }

    # per-kernel HIP-event timing on the engine's own stream (separate short run).  EVERY rank
    # takes these steps: a step is collective on a decomposed mesh.
    prof = None
    if not args.no_profile:
        dev.profile_enable(True)
        dev.step(20, want_scalars=False)
        prof = dev.profile_read()
        dev.profile_enable(False)

    if rank == 0:
        bytes_step = dev.algorithmic_bytes_per_step() if world == 1 else (
            dev.algorithmic_bytes_per_step() / max(ne_local, 1) * ne)       # whole job, all ranks
        result["config"]["algorithmic_bytes_per_step"] = bytes_step
        result["config"]["whole_step_frac_of_hbm_peak"] = bytes_step * args.steps / (ev_ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world)
        roof = None
        if prof is not None:
            kern = {n: (ms, calls) for n, ms, calls in prof}
            result["config"]["kernel_ms_per_call"] = {n: ms / calls for n, (ms, calls) in kern.items()}
            cands = [(ms, n) for n, (ms, calls) in kern.items() if n in KERNEL_BYTES]
            if cands:
                _, dom = max(cands)
                ms, calls = kern[dom]
                be, bn = KERNEL_BYTES[dom]
                if dom == "E2_update_stress" and args.rheology == "elasto-visco-plastic":
                    be, bn = be + 24, bn + 8
                kbytes = be * ne_local + bn * (nn if world == 1 else part.nnode)     # rank 0's launch
                achieved = kbytes / (ms / calls * 1e-3) / 1e9
                traffic = None
                tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
                same_workload = (world == 1 and not args.mesh_file and args.workload == "test-3d-big"
                                 and args.resolution == 400e3 / 560 and args.rheology == "elasto-visco-plastic")
                if os.path.exists(tpath) and same_workload:       # the PMC passes were made on exactly this workload
                    try:
                        # PMC passes cannot share a run with the timed one: the committed summary of
                        # `tools/summarize_pmc.py` for the same workload is reported (bytes per launch)
                        traffic = json.load(open(tpath)).get(dom, {}).get("traffic_bytes_per_launch")
                    except Exception:
                        traffic = None
                roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                        "algorithmic_bytes_per_launch": kbytes, "avg_launch_ms": ms / calls}
        result["roofline"] = roof

        cpu = None
        cpu_steps = args.cpu_steps
        if world == 1 and cpu_steps != 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            # host cores this job may really use: the affinity mask, capped at the GPU box's
            # per-GPU CPU share (16); must be set before libgomp starts its pool
            ncores = min(16, len(os.sched_getaffinity(0)))
            os.environ["OMP_NUM_THREADS"] = str(ncores)
            os.environ.setdefault("OMP_PROC_BIND", "close")
            from oracle_binding import OracleEngine, load_oracle
            threads = load_oracle(omp=True).des_oracle_set_threads(ncores)
            ora = OracleEngine(host, omp=True)
            ora.init_from_host(host)
            ora.step(1)
            if cpu_steps < 0:
                t1 = time.perf_counter(); ora.step(1); one = time.perf_counter() - t1
                cpu_steps = max(2, min(200, int(15.0 / max(one, 1e-6))))
            t1 = time.perf_counter()
            ora.step(cpu_steps)
            cw = time.perf_counter() - t1
            cpu = {"value": float(ne) * cpu_steps / cw, "unit": "element-steps/s", "cores": threads,
                   "kind": "port", "sample": "%d steps of the same %d-tet mesh (oracle, OpenMP, %d threads)"
                                                % (cpu_steps, ne, threads)}
        result["cpu_baseline"] = cpu
        print(json.dumps(result))

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
