#!/usr/bin/env python3
"""bench.py -- explicit time-steps/s x #elements of the MI355X time-stepper.

    python bench.py --gpus N --steps K --warmup W

works as written for every N: with N > 1 and no launcher around it (WORLD_SIZE unset) the process starts its own ranks
-- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py
<same arguments>` as a child process, before torch or HIP are touched -- relays rank 0's JSON line and exits with the
ranks' code.  Launched BY torch.distributed.run (RANK / WORLD_SIZE set) it is one rank, as before.

One "step" is one explicit time step (dynearthsol.cxx:768-894) of the whole mesh.  The workload
is BASELINE.json configs[2] / configs[3] as SURVEY.md 8(d) pins it down: the
benchmarks-cores/test-3d-big.cfg box (400 x 20 x 10 km), elasto-visco-plastic variant, thermal
diffusion + mixed stress + surface diffusion on, on the REFERENCE'S OWN MESH of that box at
mesh.resolution = 460 m: 1,001,310 tets / 185,637 nodes, made by the reference's TetGen
(data/test-3d-big-460.desmesh.xz; recipe `make -C oracle refmesh`).  Fields are the model's own
initial conditions (synthetic in the sense of: no input data files).  Where the mesh file is
missing, or with --mesh regular, the reference's regular mesher (meshing_option = 1,
meshing_elem_shape = 1: 5 tets per grid cell, mesh.cxx:1431-1459) builds 560 x 28 x 14 cells =
1,097,600 tets instead; `config.workload` says which.

N > 1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling by default -- the same
~1M-tet mesh cut into N slabs of contiguous node ids (configs[3]); --weak keeps 1.1M tets per GPU
(regular mesh, box N times as long).  The ghost-region exchange is RCCL inside des_dev_step; if
the engine's communicator does not come up the run FAILS (exit 3) unless DES_BENCH_TRANSPORT=host
asks for the host-staged rehearsal transport.  DES_OVERLAP=1 selects the overlapped schedule.

Prints ONE JSON line (rank 0).  `value` = elements x steps / s of the whole job, timed with state
resident in HBM; `roofline` is the dominant kernel's algorithmic bytes over its measured
HIP-event duration against 8 TB/s, with a measured device-copy ceiling beside it;
`cpu_baseline` is the CPU oracle (OpenMP build, kind "port") timed on this box's host cores on a
bounded number of steps of the same mesh.

--ndims 2: the same line for the 2-D (triangle) build -- BASELINE's "tet/tri meshes": the 400 x 100 km box of the
regular triangle mesher at 250 m (1,280,000 triangles), elasto-visco-plastic, the 2-D engine's patch passes; roofline on
its dominant kernel's own minimum bytes, CPU baseline from the 2-D oracle (oracle/libdes_oracle2d_omp.so).

N > 1 also (a) runs des_dev_comm_selfcheck before the first step (rank count, the exchange's own messages filled with a
pattern the receiver verifies double by double, the reductions; exit 4 on any mismatch), (b) reports per rank the step
time WITHOUT the exchange (sum of the passes' HIP-event times) beside the exchange's own, and (c) repeats the timed
region on a second, LARGE mesh of the same box (`config.large_mesh_series`: 8,780,800 tets at 357 m by default, i.e. 1.1M
per GPU at N = 8; --series-resolution 230 gives 32.5M) so that the record separates "the design scales" from "1M tets is
125k per MI355X".  --no-large-series skips it; N = 1 runs it too (the scaling curve needs its N = 1 point).
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
    # The node-block patch passes (csrc/passes/en1.hpp, en2.hpp, en3.hpp) are priced with THEIR OWN minimum, by
    # SURVEY 8(d)'s rules (every array once per pass, nodal records once per node) -- not with the rows of the
    # passes they replace, whose stored temporaries (ftmp 192 B, mrec / ttmp 64 B per element) they no longer move;
    # KERNEL_ROWS keeps the rows' figures for `frac_of_replaced_survey_rows`.
    # EN3: per element stress 48, volume 8, dpressure 8, marker word 4, ddp 8 W, patch record 16; per node {x,y,z,T} 32 R
    #      + 32 W, {v,m} 32 R + 32 W, ntmp 8, force 24 W, force_residual 24 W, bcflag 4, CSR offset 4
    "EN3_force_nodes":            (92, 192),
    # EN1: per element marker word 4, patch record 16; per node {x,y,z,T} 32 R + 32 W, {v,m} 32 R + 32 W, volume_n 8 W,
    #      tmass 8 W, ntmp 8 W, bcflag 4, CSR offset 4
    "EN1_mass_temperature_dvoldt": (20, 160),
    # EN2: per element etmp2 8, patch record 16; per node volume_n 8, ntmp 8 W, CSR offset 4
    "EN2_nmd_gather":             (24, 20),
    # E2<GEO> (csrc/passes/e2.hpp): the stress update that also does the end-of-step pass of the step before
    # and this step's strain rate.  It stands for SURVEY's E1 and E2 rows (728 B per element) but moves less
    # than half of that -- no stress / strain round trip for rotate_stress, no strain_rate round trip, no
    # mrec / ttmp -- so it is credited with ITS OWN minimum, counted by SURVEY 8(d)'s rules: read conn 16,
    # stress 48, strain 48, plstrain 8, volume 8, ddp 8, markers 4, top flag 1; write stress 48, strain 48,
    # strain_rate 48, delta_plstrain 8, dpressure 8, etmp 8, volume 8, volume_old 8 (evp: + viscosity 8 W,
    # T gathered anyway); nodes once each: coord 24, vel 24, T 8, ntmp 8.  KERNEL_ROWS has the rows' figure.
    # Only the LAST step of a des_dev_step call moves all of that: in the steps before it the launch leaves out
    # the stores nothing reads before the next launch overwrites them (strain_rate 48, delta_plstrain 8,
    # volume_old 8, viscosity 8; E2G_INTERIOR below), and is credited with what it then has to move.
    "E2G_geom_rotate_update_stress": (325, 64),
}
# The 2-D engine (csrc/des_dev2d.hip), dominant launch k2_stress<M, 2> = compute_volume + rotate_stress of the step before +
# compute_edvoldt + update_stress: per triangle read conn 12, mono word 4, bulk / shear modulus 16, stress 24,
# strain 24, volume 8, plstrain 8 = 96 (round 5, late: NOT the strain rate any more -- the pass forms it from the coordinates and
# velocities it gathers anyway and the temperature / dvoldt pass no longer stores it; DES2D_SR_FUSE=0: + 24, K2_SR_STORED); write
# volume 8, stress 24, strain 24, dpressure 8, etmp 8 = 72; nodes once each: coord 16, vel 16, T 8, ntmp 8.  The last step
# of a call also stores volume_old 8, edvoldt 8, strain_rate 24, viscosity 8, delta_plstrain 8 (K2_STRESS_LAST).
# (Round 5: a model with ONE material does not read the two moduli per triangle -- the means of one material are its values,
#  des_dev2d.hip: prop2 --: 16 B less, K2_ONE_MATERIAL; the bench model is such a model.)
KERNEL_BYTES_2D = {"K2_stress": (168, 48)}
K2_STRESS_LAST = (224, 48)
K2_SR_STORED = 24
K2_ONE_MATERIAL = 16
E2G_INTERIOR = (261, 64)          # read 141 (as above) + write stress 48, strain 48, dpressure 8, etmp 8, volume 8
KERNEL_ROWS = {"E2G_geom_rotate_update_stress": (364 + 364, 56 + 8), "EN3_force_nodes": (260 + 160, 40 + 160),
               "EN1_mass_temperature_dvoldt": (224, 64), "EN2_nmd_gather": (48, 20)}


def _note_factory(rank, t_begin):
    def note(msg):
        if os.environ.get("DES_BENCH_VERBOSE"):
            sys.stderr.write("[bench rank %d %.1fs] %s\n" % (rank, time.perf_counter() - t_begin, msg)); sys.stderr.flush()
    return note


class _Ctx:
    """what every series of one bench.py run shares: ranks, torch.distributed, the transport choice"""
    pass


def make_engine(ctx, host, des):
    """One engine on this rank for `host`'s model: the whole mesh (N = 1) or this rank's slab with its RCCL communicator
    (self-checked) attached.  Returns (engine, partition or None, local element count, transport text)."""
    import torch
    world, rank, dist = ctx.world, ctx.rank, ctx.dist
    transport = "RCCL ncclSend/ncclRecv on the engine stream"
    if world == 1:
        dev = des.DeviceEngine(host, device=ctx.device)
        dev.init_from_host(host)
        return dev, None, host.nelem, transport
    from dynearthsol_amd.decomp import Partition, init_rank

    class _Comm:          # init only: the first compute_dt / wall extent go through the engine's own allreduce
        def reduce_wall(self, engine): pass
        def reduce_dt(self, engine, recompute):
            return engine.compute_dt()
    part = Partition(host, world, rank)
    dev = des.DeviceEngine(part, device=ctx.device)
    dev.set_halo(part)
    # The ghost-region exchange runs inside des_dev_step on RCCL.  If the engine's communicator does not come up, or
    # its start-up self-check fails, the bench FAILS: a number over another transport, or over wires that do not carry
    # the bytes, must not pass for the real thing.  DES_BENCH_TRANSPORT=host asks for the rehearsal transport
    # explicitly (the same step driven in its two phases, ghost state staged through the host over gloo).
    want_host = os.environ.get("DES_BENCH_TRANSPORT", "rccl") != "rccl"
    if not want_host:
        def agree(ok, why, what, rc):
            """every rank learns whether ALL ranks got through `what`; if not, all of them leave together with `rc`"""
            if why:
                sys.stderr.write("rank %d: %s\n" % (rank, why))
            flag = torch.tensor([ok], dtype=torch.int32, device=ctx.tdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                sys.stderr.write("bench.py: %s; no result (DES_BENCH_TRANSPORT=host selects the host-staged rehearsal transport)\n" % what)
                _set_exit_code(rc)
                dist.destroy_process_group()
                sys.exit(rc)
        ok, why = 1, ""
        try:
            dev.comm_init(dist, rank, world)
        except des.DesError as e:
            ok, why = 0, str(e)
        agree(ok, why, "the RCCL communicator of the engine did not come up on every rank", 3)
        # the self-check is collective: entered only once every rank is known to hold a communicator
        ok, why = 1, ""
        try:
            dev.comm_selfcheck(world)
        except des.DesError as e:
            ok, why = 0, str(e)
        agree(ok, why, "the RCCL self-check failed on at least one rank", 4)
        init_rank(dev, part, _Comm())
    else:
        from dynearthsol_amd.decomp import PhasedStepper, TorchComm
        transport = "host-staged over gloo (DES_BENCH_TRANSPORT=host: rehearsal, not RCCL)"
        if ctx.gloo_group is None:
            ctx.gloo_group = dist.new_group(backend="gloo")
        comm = TorchComm(dist, group=ctx.gloo_group)
        init_rank(dev, part, comm)
        stepper = PhasedStepper(dev, part, comm)
        _step = dev.step
        dev.step = lambda n, want_scalars=True: (stepper.step(n), _step(0, want_scalars=want_scalars))[1]
    return dev, part, part.nelem, transport


def timed_region(ctx, dev, steps):
    """EXACTLY `steps` steps between two barrier + synchronize pairs; wall = MAX over ranks; also the HIP-event time"""
    import torch
    ctx.barrier(dev)
    t0 = time.perf_counter()
    dev.timer_start()
    dev.step(steps, want_scalars=False)
    ev_ms = dev.timer_stop()
    ctx.barrier(dev)
    wall = time.perf_counter() - t0
    if ctx.dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device=ctx.tdev)
        ctx.dist.all_reduce(t, op=ctx.dist.ReduceOp.MAX)
        wall = float(t.item())
    return wall, ev_ms


def probe_schedules(ctx, dev, transport):
    """N > 1, RCCL transport: which schedule -- everything in order on the engine's stream, or the exchange on a side stream
    beside the next step's passes on the deep part of the slab (engine/launch.hpp: deep_split_ok)?  Neither has run on two
    physical GPUs in the build container, so the untimed part of the run measures both (40 steps each, MAX over ranks)
    and the timed region takes the faster; DES_OVERLAP=0 / 1 pins it.  Both give the same bits
    (tests/test_gpu_headline_decomp.py; the 2-D engine's overlapped schedule: tests/test_gpu_2d_decomp.py)."""
    import torch
    if ctx.world == 1 or not transport.startswith("RCCL") or os.environ.get("DES_OVERLAP") is not None:
        return None
    probe = {}
    for on in (0, 1):
        dev.set_overlap(on)
        dev.step(10, want_scalars=False)
        ctx.barrier(dev)
        t1 = time.perf_counter()
        dev.step(40, want_scalars=False)
        ctx.barrier(dev)
        tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=ctx.tdev)
        ctx.dist.all_reduce(tt, op=ctx.dist.ReduceOp.MAX)
        probe[on] = float(tt.item()) / 40
    pick = 1 if probe[1] < probe[0] else 0
    dev.set_overlap(pick)
    dev.step(10, want_scalars=False)
    dev.sync()
    return {"in_order_ms_per_step": 1e3 * probe[0], "overlapped_ms_per_step": 1e3 * probe[1], "picked": "overlapped" if pick else "in order"}


def profile_leg(ctx, dev, nsteps=20):
    """per-kernel HIP-event timing on the engine's own stream (separate short run; EVERY rank takes these steps: a step
    is collective on a decomposed mesh).  Returns [(name, ms, calls)] and, for N > 1, per rank the exchange's us per call
    and the step time WITHOUT the exchange (sum of every other pass's HIP-event time / steps; the 2-D engine times its six
    priced launches only, so its figure is a lower bound: the small surface / boundary kernels are not in it)."""
    import torch
    dev.profile_enable(True)
    dev.step(nsteps, want_scalars=False)
    prof = dev.profile_read()
    dev.profile_enable(False)
    per_rank = None
    if ctx.world > 1:
        ex = [ms / calls * 1e3 for n, ms, calls in prof if n == "ghost_exchange"]
        rest = sum(ms for n, ms, calls in prof if n != "ghost_exchange") / nsteps * 1e3
        t = torch.tensor([ex[0] if ex else -1.0, rest], dtype=torch.float64, device=ctx.tdev)
        allx = [torch.zeros_like(t) for _ in range(ctx.world)]
        ctx.dist.all_gather(allx, t)
        # per rank: pack + grouped ncclSend/ncclRecv + unpack, HIP events on the stream they run on (includes waiting
        # for the slower neighbour); -1: this rank's exchange did not go through the engine
        per_rank = {"exchange_us_per_rank": [float(x[0].item()) for x in allx],
                    "step_us_without_exchange_per_rank": [float(x[1].item()) for x in allx]}
    return prof, per_rank


def _set_exit_code(rc):
    """a rank that refuses to report (3: no RCCL communicator, 4: its self-check failed) leaves its code for the launcher:
    torch.distributed.run itself exits 1 whatever its children returned"""
    path = os.environ.get("DES_BENCH_RC_FILE")
    if path:
        try:
            with open(path, "w") as f:
                f.write("%d\n" % rc)
        except OSError:
            pass


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks as `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` in a child process,
    relay rank 0's JSON line on stdout and everything else on stderr, return the exit code (3 / 4 of the RCCL refusals
    kept).  Called before torch is imported."""
    import socket
    import subprocess
    import tempfile
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rc_file = tempfile.NamedTemporaryFile(prefix="des_bench_rc_", delete=False)
    rc_file.close()
    env = dict(os.environ, DES_BENCH_RC_FILE=rc_file.name)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # the host driver only supports dmabuf IPC (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    try:
        for line in proc.stdout:
            is_result = line.startswith("{") and '"metric"' in line
            (sys.stdout if is_result else sys.stderr).write(line)
            (sys.stdout if is_result else sys.stderr).flush()
        rc = proc.wait()
    except BaseException:
        proc.terminate()
        try:
            proc.wait(10)
        except subprocess.TimeoutExpired:
            proc.kill()
        raise
    finally:
        try:
            left = open(rc_file.name).read().strip()
            os.unlink(rc_file.name)
        except OSError:
            left = ""
    if rc != 0 and left.isdigit() and int(left) != 0:
        rc = int(left)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ndims", type=int, default=3, choices=[2, 3],
                    help="3 (default): the headline tet-mesh workload; 2: the 2-D (triangle) build on its regular mesh")
    ap.add_argument("--mesh", default=None, choices=["tetgen", "regular"],
                    help="tetgen: the reference's TetGen mesh of the box at 460 m, 1,001,310 tets (default when "
                         "data/test-3d-big-460.desmesh.xz is there); regular: the reference's regular mesher, 1,097,600 tets")
    ap.add_argument("--resolution", type=float, default=None, help="regular mesh: cell size in m (3-D default 400e3/560, 2-D default 250); implies --mesh regular")
    ap.add_argument("--cpu-steps", type=int, default=-1, help="steps of the CPU baseline (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the device-copy ceiling measurement")
    ap.add_argument("--no-large-series", action="store_true", help="skip the second timed region on the large mesh (config.large_mesh_series)")
    ap.add_argument("--series-resolution", type=float, default=400e3 / 1120,
                    help="cell size of the large-mesh series (default 357.14 m: 8,780,800 tets; 230: 32.5M tets, ~2 min of host mesh building per rank)")
    ap.add_argument("--series-steps", type=int, default=50)
    ap.add_argument("--no-elide-compare", action="store_true", help="skip the extra DES_E2_ELIDE=0 engine (config.ms_per_step_every_step_stores_every_field)")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: weak scaling (regular mesh, the box grows N times in x so every GPU keeps 1.1M tets) "
                         "instead of the default strong scaling on the fixed ~1M-tet mesh")
    ap.add_argument("--strong", action="store_true", help="(the default; kept for older command lines)")
    ap.add_argument("--mesh-file", default=None, help="another .desmesh file of the same box instead")
    ap.add_argument("--workload", default="test-3d-big", choices=["test-3d-big", "test-3d-equ-long"],
                    help="test-3d-big (default, the BASELINE config) or the reference's 984,375-tet seven-material "
                         "regular-mesh benchmark benchmarks-cores/test-3d-equ-long.cfg (values restated in tests/cfgs.py); N=1")
    ap.add_argument("--averaged-fields", action="store_true",
                    help="also run Output::average_fields inside the step (sim.is_outputting_averaged_fields = yes)")
    ap.add_argument("--rheology", default="elasto-visco-plastic",
                    help="diagnostic only: the headline workload is elasto-visco-plastic")
    args = ap.parse_args()

    t_begin = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("DES_BENCH_VERBOSE"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("DES_BENCH_WATCHDOG", "60")), repeat=False)
    elif world > 1:
        # first contact with several GPUs: a rank stuck in a collective (a neighbour that never arrives) must not hang the
        # job silently -- after DES_BENCH_WATCHDOG seconds (default 900) every rank prints where it stands and exits non-zero
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("DES_BENCH_WATCHDOG", "900")), exit=True)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  Nothing here has imported torch or
        # touched HIP, and the ranks are CHILD processes (never an exec of a process that has initialised the GPU).
        sys.exit(launch_ranks(args.gpus))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch `python bench.py --gpus N`, or torch.distributed.run with "
                 "--nproc-per-node equal to --gpus)" % (args.gpus, world))

    import torch
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DES_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend)

    import dynearthsol_amd as des
    note = _note_factory(rank, t_begin)

    ctx = _Ctx()
    ctx.world, ctx.rank, ctx.dist, ctx.ndims = world, rank, dist, args.ndims
    ctx.device = int(os.environ.get("DES_BENCH_DEVICE", local_rank))
    ctx.tdev = "cuda" if backend == "nccl" else "cpu"
    ctx.gloo_group = None

    def barrier(dev):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        dev.sync()
    ctx.barrier = barrier

    # ---- the model ------------------------------------------------------------------------------------------------
    strong = not args.weak
    mesh_file = args.mesh_file
    overrides = "" if args.rheology == "elasto-visco-plastic" else "mat.rheology_type = %s\n" % args.rheology
    if args.averaged_fields:
        overrides += "sim.is_outputting_averaged_fields = yes\nmesh.quality_check_step_interval = 100\n"
    xlen = 400e3 * (1 if strong else world)
    note("building the host model")
    if args.ndims == 2:
        assert args.workload == "test-3d-big" and not mesh_file and not args.weak, "--ndims 2 runs the regular 2-D box (strong scaling)"
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import cfgs
        resolution = args.resolution if args.resolution is not None else 250.0
        mesh_kind = "regular2d"
        host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, lx=400e3, lz=100e3, res=resolution)), overrides=overrides or None, ndims=2)
    else:
        # which mesh: the reference's TetGen mesh of the box unless a regular one is asked for / needed
        mesh_kind = "file" if mesh_file else args.mesh
        if mesh_kind is None:
            mesh_kind = "regular" if (args.resolution is not None or args.weak or args.workload != "test-3d-big") else "tetgen"
        if mesh_kind == "tetgen":
            mesh_file = des.reference_mesh("test-3d-big-460")
            if mesh_file is None:
                if args.mesh == "tetgen":
                    sys.exit("bench.py: data/test-3d-big-460.desmesh.xz is missing (make -C oracle refmesh)")
                mesh_kind = "regular"
            elif args.weak:
                sys.exit("bench.py: --weak scales the regular mesh; it cannot be combined with --mesh tetgen")
        resolution = args.resolution if args.resolution is not None else 400e3 / 560
        # weak scaling: the box is repeated N times along x (same resolution), then cut into N slabs of
        # contiguous node ids -- every GPU holds ~1.1M tets plus its four-layer ghost region
        if mesh_file:
            overrides += "mesh.meshing_option = 2\nmesh.meshing_elem_shape = 0\n"
        if args.workload == "test-3d-equ-long":
            assert world == 1 and not mesh_file, "--workload test-3d-equ-long is a single-GPU regular-mesh workload"
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import cfgs
            host = des.Host(cfg_text=cfgs.make_equ(long=True),
                            overrides=(overrides or "") + "sim.max_steps = 1000000\nsim.output_step_interval = 1000000\n")
        else:
            host = des.Host(cfg_text=BENCH_CFG.format(res="460.0" if mesh_kind == "tetgen" else repr(resolution), xlen=repr(xlen)),
                            overrides=overrides or None, mesh_file=mesh_file)
    note("engine")
    dev, part, ne_local, transport = make_engine(ctx, host, des)
    note("initialised; warm-up")
    ne, nn = host.nelem, host.nnode          # global counts: `value` counts every element once

    dev.step(args.warmup, want_scalars=False) if args.warmup > 0 else None
    dev.sync()
    schedule_probe = probe_schedules(ctx, dev, transport)

    note("timed region")
    wall, ev_ms = timed_region(ctx, dev, args.steps)
    sc = dev.step(0)
    nan = dev.check_nan()

    value = float(ne) * args.steps / wall
    if args.ndims == 2:
        workload = ("2-D build: 400x100 km box, " + args.rheology + ", thermal+NMD+surface diffusion on, the reference's regular triangle "
                    "mesher at %.6g m, %d triangles / %d nodes in total" % (resolution, ne, nn))
    else:
        workload = (("test-3d-equ-long.cfg box 250x50x125 km, 7 materials, " if args.workload == "test-3d-equ-long" else
                     "test-3d-big.cfg box %.0fx20x10 km, " % (xlen / 1e3)) + args.rheology + ", thermal+NMD+surface diffusion on, "
                    + ("averaged output fields on, " if args.averaged_fields else "")
                    + {"tetgen": "the reference's TetGen mesh at mesh.resolution = 460 m (data/test-3d-big-460.desmesh.xz), ",
                       "file": "mesh file %s, " % os.path.basename(mesh_file or ""),
                       "regular": "the reference's regular 5-tet mesher at %.6g m, " % resolution}[mesh_kind]
                    + "%d tets / %d nodes in total" % (ne, nn))
    workload += "" if world == 1 else (", cut %d ways (strong scaling)" % world if strong else ", %d x the box (weak scaling)" % world)
    result = {
        "metric": "explicit time-steps/sec x #elements",
        "value": value,
        "unit": "element-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * wall / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "ndims": args.ndims,
            "nelem": ne, "nnode": nn, "nelem_local_rank0": ne_local,
            "parallelism": "single GPU" if world == 1 else
                           "%d slabs of contiguous node ids, four-layer ghost region, one exchange per step; transport: %s" % (world, transport),
            "steps_per_s": args.steps / wall,
            "hip_event_ms_per_step": ev_ms / args.steps,
            "nan_entries": nan, "status": sc.status,
            "libm": os.environ.get("DES_LIBM", "portable (pow/exp = glibc's bits)"),
            # inside the one des_dev_step call of the timed region only the LAST step stores strain_rate, viscosity,
            # delta_plstrain and volume_old (nothing reads them before the next step overwrites them; every field a
            # caller can download after the call is the reference's).  DES_E2_ELIDE=0 (2-D: DES2D_ELIDE=0): every step stores them
            "interior_step_store_elision": os.environ.get("DES2D_ELIDE" if args.ndims == 2 else "DES_E2_ELIDE", "1") != "0",
            # the whole timed region is ONE des_dev_step call of `steps` steps: its first and last step take the classic
            # passes, every 10th step carries a compute_dt, the others are the fused step (EN1, E2<GEO>, EN2, EN3; the
            # surface step rides in the next step's EN1 / E2 unless DES_S2_DEFER=0) -- a shorter call has a larger share
            # of the slower first / last steps
            "steps_per_call": args.steps,
            "surface_step_in_next_steps_passes": os.environ.get("DES2D_SURF_DEFER" if args.ndims == 2 else "DES_S2_DEFER", "1") != "0",
            "first_step_on_finished_state": world == 1 and os.environ.get("DES_FRESH", "1") != "0" and args.warmup > 0,
        },
    }
    if world > 1:
        # what actually carried the ghost region: ranks of the engine's RCCL communicator (0 = none:
        # host-staged rehearsal), and the redundant work the four ghost layers cost
        info = dev.comm_info()
        counts = torch.tensor([float(ne_local), float(info["rccl_ranks"])], dtype=torch.float64, device=ctx.tdev)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        ne_sum = float(counts[0].item())
        result["config"].update({
            "rccl_ranks": info["rccl_ranks"],
            "rccl_ranks_sum_over_ranks": int(counts[1].item()),          # = world^2 when every rank is in one communicator
            "rccl_selfcheck": "passed on every rank (des_dev_comm_selfcheck)" if transport.startswith("RCCL") else "not run (host-staged transport)",
            "overlapped_schedule": info["overlapped"],
            "schedule_probe": schedule_probe,
            "ghost_work_share": (ne_sum - ne) / ne_sum,                    # elements computed redundantly / all computed
            "nelem_local_sum": int(ne_sum),
        })

    prof = None
    if not args.no_profile:
        prof, per_rank = profile_leg(ctx, dev)
        if per_rank:
            result["config"].update(per_rank)

    ceiling = None
    if rank == 0 and not args.no_ceiling:
        try:
            ceiling = des.copy_ceiling(1 << 30, 20, ctx.device)
        except des.DesError as e:
            sys.stderr.write("copy ceiling: %s\n" % e)
    # ... and what a kernel reaches from HBM that ONLY moves bytes in the dominant launch's memory shape (SoA planes of doubles,
    # one element per lane: 18 read + 15 written for E2<GEO>, 12 + 9 for the 2-D stress update; 8.8M elements, past the 256-MiB
    # Infinity Cache): the pass could not be faster than that if its arithmetic were free (des_dev_plane_ceiling)
    shape_ceiling = None
    if rank == 0 and not args.no_ceiling:
        try:
            shape_ceiling = des.plane_ceiling(12 if args.ndims == 2 else 18, 9 if args.ndims == 2 else 15, 8800000, 10, ctx.device)
        except des.DesError as e:
            sys.stderr.write("plane ceiling: %s\n" % e)

    bytes_step_local = dev.algorithmic_bytes_per_step()
    nn_local = nn if world == 1 else part.nnode
    # every DES_* switch the library has found set so far ('' = all defaults): taken here, before the comparison engine below
    # is created with its one deliberate switch
    engine_switches = des.config_string()

    # ---- the same model with every step storing every field (no store elision): a second engine, N = 1 only ------------
    no_elide_ms = None
    elide_var = "DES2D_ELIDE" if args.ndims == 2 else "DES_E2_ELIDE"
    if world == 1 and not args.no_elide_compare and os.environ.get(elide_var, "1") != "0":
        note("engine without the store elision")
        os.environ[elide_var] = "0"
        try:
            dev2 = des.DeviceEngine(host, device=ctx.device)
            dev2.init_from_host(host)
            dev2.step(max(args.warmup, 2), want_scalars=False)
            w2, _ = timed_region(ctx, dev2, args.steps)
            no_elide_ms = 1e3 * w2 / args.steps
            dev2.close()
        finally:
            del os.environ[elide_var]
    result["config"]["ms_per_step_every_step_stores_every_field"] = no_elide_ms

    # ---- second series: the same box on a LARGE mesh (a shard stays bandwidth-bound at N = 8) ---------------------------
    series = None
    if not args.no_large_series and args.ndims == 3 and args.workload == "test-3d-big" and strong and not args.averaged_fields:
        note("large-mesh series: host model")
        dev.close()
        dev = None
        t_s = time.perf_counter()
        host_l = des.Host(cfg_text=BENCH_CFG.format(res=repr(args.series_resolution), xlen=repr(400e3)),
                          overrides=("" if args.rheology == "elasto-visco-plastic" else "mat.rheology_type = %s\n" % args.rheology) or None)
        build_s = time.perf_counter() - t_s
        note("large-mesh series: engine")
        dev_l, part_l, ne_local_l, transport_l = make_engine(ctx, host_l, des)
        dev_l.step(10, want_scalars=False)
        dev_l.sync()
        probe_l = probe_schedules(ctx, dev_l, transport_l)
        wall_l, ev_l = timed_region(ctx, dev_l, args.series_steps)
        sc_l = dev_l.step(0)
        series = {"workload": "test-3d-big.cfg box 400x20x10 km, %s, the reference's regular 5-tet mesher at %.6g m, %d tets / %d nodes in total%s"
                              % (args.rheology, args.series_resolution, host_l.nelem, host_l.nnode, "" if world == 1 else ", cut %d ways (strong scaling)" % world),
                  "nelem": host_l.nelem, "nnode": host_l.nnode, "nelem_local_rank0": ne_local_l, "steps": args.series_steps, "warmup": 10,
                  "ms_per_step": 1e3 * wall_l / args.series_steps, "value": float(host_l.nelem) * args.series_steps / wall_l,
                  "unit": "element-steps/s", "hip_event_ms_per_step": ev_l / args.series_steps, "status": sc_l.status,
                  "host_mesh_build_s": build_s, "schedule_probe": probe_l}
        if world > 1:
            series["overlapped_schedule"] = dev_l.comm_info()["overlapped"]
            c = torch.tensor([float(ne_local_l)], dtype=torch.float64, device=ctx.tdev)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            series["ghost_work_share"] = (float(c.item()) - host_l.nelem) / float(c.item())
        if not args.no_profile:
            prof_l, per_rank_l = profile_leg(ctx, dev_l, 10)
            series["kernel_ms_per_call"] = {n: ms / calls for n, ms, calls in prof_l}
            if per_rank_l:
                series.update(per_rank_l)
        dev_l.close()
    result["config"]["large_mesh_series"] = series

    if rank == 0:
        bytes_step = bytes_step_local if world == 1 else (bytes_step_local / max(ne_local, 1) * ne)       # whole job, all ranks
        result["config"]["algorithmic_bytes_per_step"] = bytes_step
        # CONTRACT bytes (SURVEY 8d's B_alg: what the reference's pass structure would have to move) per second against
        # the peak -- work done per second, not a bandwidth: the fused step moves far less (real_traffic_* below)
        result["config"]["whole_step_frac_of_hbm_peak"] = bytes_step * args.steps / (ev_ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world)
        result["config"]["engine_switches"] = engine_switches
        roof = None
        # HBM bytes from the PMC counters cannot be collected inside the timed run (separate rocprofv3 passes,
        # MI355X_MICROARCH.md): tools/measure_traffic.py makes them for a named workload and commits the summary; it is
        # quoted only for that very workload
        tj, tsrc = None, None
        for tname in (("r05_pmc_traffic_2d.json", "r04_pmc_traffic_2d.json") if args.ndims == 2 else
                      ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "pmc_traffic.json")):
            tpath = os.path.join(ROOT, "profiles", tname)
            if not os.path.exists(tpath):
                continue
            try:
                cand = json.load(open(tpath))
            except Exception:
                continue
            if (world == 1 and args.rheology == "elasto-visco-plastic" and not args.averaged_fields
                    and cand.get("workload", {"nelem": 1097600, "nnode": 244035}) == {"nelem": ne, "nnode": nn}):
                tj, tsrc = cand, "profiles/" + tname
                break
        if tj is not None:
            # what a PLAIN fused step really moves: the PMC traffic of its launches, summed
            # (2-D: compute_mass rides in the temperature / dvoldt pass of a plain step unless DES2D_MASS_FUSE=0)
            plain = (("K2P_temp_dvoldt", "K2_stress", "K2_node_avg", "K2P_force") + (("K2P_mass",) if os.environ.get("DES2D_MASS_FUSE") == "0" else ())) if args.ndims == 2 else \
                    ("EN1_mass_temperature_dvoldt", "E2G_geom_rotate_update_stress", "EN2_nmd_gather", "EN3_force_nodes")
            if all(k in tj for k in plain):
                real = sum(tj[k]["traffic_bytes_per_launch"] for k in plain)
                result["config"]["real_traffic_bytes_per_plain_step"] = real
                result["config"]["real_traffic_source"] = tsrc + " (PMC FETCH_SIZE / WRITE_SIZE of " + " + ".join(k.split("_")[0] for k in plain) + ")"
                result["config"]["real_traffic_frac_of_hbm_peak"] = real / (ev_ms / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS
        if prof is not None:
            kern = {n: (ms, calls) for n, ms, calls in prof}
            result["config"]["kernel_ms_per_call"] = {n: ms / calls for n, (ms, calls) in kern.items()}
            table = KERNEL_BYTES_2D if args.ndims == 2 else KERNEL_BYTES
            cands = [(ms, n) for n, (ms, calls) in kern.items() if n in table]
            if cands:
                _, dom = max(cands)
                ms, calls = kern[dom]
                be, bn = table[dom]
                elided = result["config"]["interior_step_store_elision"]
                if dom == "E2_update_stress" and args.rheology == "elasto-visco-plastic":
                    be, bn = be + 24, bn + 8
                if dom == "E2G_geom_rotate_update_stress" and args.rheology == "elasto-visco-plastic":
                    be += 8
                if dom == "E2G_geom_rotate_update_stress" and elided and calls > 1:
                    # the profiled leg is ONE des_dev_step call: its last launch stores every field, the others
                    # are interior launches -- the average launch is credited with the average of the two
                    be = (E2G_INTERIOR[0] * (calls - 1) + be) / calls
                if dom == "E2G_geom_rotate_update_stress" and args.averaged_fields:
                    be += 112           # Output::average_fields rides in the pass: stress_avg 48 R + 48 W, dplstrain_avg 8 R + 8 W
                if dom == "K2_stress":
                    if not elided:
                        be = K2_STRESS_LAST[0]
                    elif calls > 1:
                        be = (be * (calls - 1) + K2_STRESS_LAST[0]) / calls
                    if os.environ.get("DES2D_SR_FUSE") == "0" or world > 1:
                        be += K2_SR_STORED           # (a decomposed mesh keeps the stored strain rate)
                    if int(host.params.nmat) == 1:
                        be -= K2_ONE_MATERIAL
                kbytes = be * ne_local + bn * nn_local     # rank 0's launch
                achieved = kbytes / (ms / calls * 1e-3) / 1e9
                traffic = tj[dom].get("traffic_bytes_per_launch") if (tj is not None and dom in tj) else None
                roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc if traffic else None,
                        "ceiling_GBs": ceiling, "frac_of_ceiling": (achieved / ceiling) if ceiling else None,
                        "shape_ceiling_GBs": shape_ceiling if dom in ("E2G_geom_rotate_update_stress", "K2_stress") else None,
                        "algorithmic_bytes_per_launch": kbytes, "avg_launch_ms": ms / calls}
                if roof["shape_ceiling_GBs"]:
                    roof["frac_of_shape_ceiling"] = achieved / roof["shape_ceiling_GBs"]
                if dom in KERNEL_ROWS:
                    # the same launch priced with the SURVEY rows it replaces (E1 + E2; evp + 24 / + 8)
                    re_, rn_ = KERNEL_ROWS[dom]
                    if args.rheology == "elasto-visco-plastic" and dom == "E2G_geom_rotate_update_stress":
                        re_, rn_ = re_ + 24, rn_ + 8
                    rows_bytes = re_ * ne_local + rn_ * nn_local
                    roof["frac_of_replaced_survey_rows"] = rows_bytes / (ms / calls * 1e-3) / 1e9 / HBM_PEAK_GBS
        result["roofline"] = roof

        cpu = None
        cpu_steps = args.cpu_steps
        if world == 1 and cpu_steps != 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            # host cores this job may really use: the affinity mask, capped at the GPU box's
            # per-GPU CPU share (16); must be set before libgomp starts its pool
            from oracle_binding import cpu_budget
            ncores = min(16, cpu_budget())
            os.environ["OMP_NUM_THREADS"] = str(ncores)
            os.environ.setdefault("OMP_PROC_BIND", "close")
            from oracle_binding import OracleEngine, load_oracle
            threads = load_oracle(omp=True, ndims=args.ndims).des_oracle_set_threads(ncores)
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
                   "kind": "port", "sample": "%d steps of the same %d-%s mesh (oracle, OpenMP, %d threads)"
                                                % (cpu_steps, ne, "triangle" if args.ndims == 2 else "tet", threads)}
        result["cpu_baseline"] = cpu
        import faulthandler; faulthandler.cancel_dump_traceback_later()
        print(json.dumps(result))

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
