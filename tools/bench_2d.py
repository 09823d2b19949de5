#!/usr/bin/env python3
"""Step rate of the 2-D (triangle) engine on the regular 2-D mesh, on 1 or N GPUs of one node -- the tri-mesh line beside
bench.py's tet-mesh one (bench.py stays the headline benchmark: BASELINE.json's metric is quoted on the 3-D model).

    python tools/bench_2d.py [--resolution 250] [--steps 400] [--warmup 40]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_2d.py ...

N > 1: strong scaling, the same 400 km x 100 km box cut into N node slabs (host/partition.cpp), one process per GPU, the
ghost-region exchange and the two small reductions inside des_dev_step on RCCL (csrc/des_dev2d.hip: exchange_rccl); the
time is the MAX over ranks between two barriers, `value` counts every element once.  Without RCCL (several ranks on one
GPU) the run fails rather than time another transport."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfgs                                            # noqa: E402
import dynearthsol_amd as des                          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--resolution", type=float, default=250.0)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, lx=400e3, lz=100e3, res=args.resolution)), ndims=2)
    dist = None
    if world == 1:
        dev = des.DeviceEngine(host, device=local)
        dev.init_from_host(host)
        ne_local = host.nelem
    else:
        import torch
        import torch.distributed as dist
        from dynearthsol_amd.decomp import Partition, init_rank
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl")

        class _Comm:                                   # the engine reduces by itself once the communicator is attached
            def reduce_wall(self, engine): pass
            def reduce_dt(self, engine, recompute): return engine.compute_dt()
        part = Partition(host, world, rank)
        dev = des.DeviceEngine(part, device=local)
        dev.set_halo(part)
        dev.comm_init(dist, rank, world)
        init_rank(dev, part, _Comm())
        ne_local = part.nelem

    def barrier():
        if dist is not None:
            dist.barrier()
        dev.sync()
    dev.step(args.warmup, want_scalars=False)
    barrier()
    t0 = time.perf_counter()
    sc = dev.step(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([ne_local], dtype=torch.int64, device="cuda")
        dist.all_reduce(c)
        ne_sum = int(c.item())
    else:
        ne_sum = ne_local
    if rank == 0:
        print(json.dumps({"metric": "explicit time-steps/sec x #elements (2-D, triangles)", "value": host.nelem * args.steps / dt,
                          "unit": "element-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "dtype": "f64",
                          "data": "synthetic",
                          "config": {"workload": "400 km x 100 km box at %g m, elasto-visco-plastic, thermal + NMD + surface diffusion, "
                                                 "regular triangle mesh" % args.resolution, "nelem": host.nelem, "nnode": host.nnode,
                                     "nelem_local_sum": ne_sum, "status": sc.status, "dt": sc.dt,
                                     "transport": "RCCL inside des_dev_step" if world > 1 else "single GPU"}}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
