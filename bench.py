#!/usr/bin/env python3
"""Headline benchmark: DCGAN 28x28 (GAN_distributed graphs) data-parallel training, bf16, images/s.

    python bench.py --gpus N --steps K --warmup W            (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = the reference iteration (/root/reference/GAN_distributed.py:133-135): a D update on
[B real ; B generated] and a G update on B noise vectors, Adam(2e-4), per rank, with one flat
RCCL all-reduce per network per step.  B = 1000 per GPU as in the reference (weak scaling: the
global batch grows with N).  `value` = real images consumed per second by the whole job.
Other BASELINE.json configs: --config dcgan64 | classifier | localsgd.
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: must be in the environment before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="dcgan28", choices=["dcgan28", "dcgan64", "classifier", "localsgd", "ensemble32"])
    ap.add_argument("--members", type=int, default=10, help="ensemble32: members per GPU")
    ap.add_argument("--preset", default="ddl", help="ensemble32: search-space preset (ddl, ddl_ray, ddl_normal3, ...)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU real images per step (0 = config default)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off", "full"],
                    help="hipGraph capture: on = graph segments around eagerly issued collectives, full = the "
                         "collectives are captured too (one graph per step), auto = full if the start-up self-test "
                         "of a captured all-reduce passes on every rank (parallel/selfcheck.py), else on")
    ap.add_argument("--comm-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--comm", default="torch", choices=["torch", "rccl", "p2p"],
                    help="gradient all-reduce through torch.distributed (nccl == RCCL), the native C++ RCCL "
                         "communicator, or the xGMI direct all-reduce kernel with the optimizer fused into its "
                         "epilogue (csrc/comm/p2p_allreduce.hip; validated against RCCL at start-up, RCCL if it fails)")
    ap.add_argument("--avg-every", type=int, default=8, help="local-SGD averaging period (config localsgd)")
    ap.add_argument("--zero", action="store_true",
                    help="dcgan configs, N>1: reduce-scatter + sharded update + all-gather instead of all-reduce")
    ap.add_argument("--force-dp", action="store_true",
                    help="diagnostics: run the data-parallel code path (segmented graphs + collectives) on a "
                         "1-rank process group, to price its overhead without a second GPU")
    ap.add_argument("--device", default="cuda")
    a = ap.parse_args()

    from distributeddeeplearning_amd.parallel import ddp
    on_gpu = a.device == "cuda"
    if on_gpu and not torch.cuda.is_available():
        print("bench.py needs a GPU (use --device cpu for a plumbing run)", file=sys.stderr)
        sys.exit(2)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(f"cuda:{local}") if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(device)
    if a.force_dp and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        kw = {"device_id": device} if on_gpu else {}
        torch.distributed.init_process_group("nccl" if on_gpu else "gloo", rank=0, world_size=1, **kw)
    rank, world, _ = ddp.init_process_group(device=device)
    if world != a.gpus:
        if rank == 0:
            print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dp = None
    if world > 1 or a.force_dp:
        cd = torch.bfloat16 if a.comm_dtype == "bf16" else None
        if a.comm == "rccl" and on_gpu:
            from distributeddeeplearning_amd.parallel.rccl import RcclCommunicator
            dp = RcclCommunicator.from_env(device, comm_dtype=cd)
        else:
            dp = ddp.FlatDataParallel(comm_dtype=cd)
        if a.comm == "p2p" and on_gpu:
            from distributeddeeplearning_amd.parallel.p2p import P2PAllReduce
            p2p, p2p_ok = None, False
            try:
                p2p = P2PAllReduce(dp, device, 1 << 22)          # 16 MB of fp32: the largest flat gradient is 4.3 MB
            except Exception as e:      # noqa: BLE001 - fall back below, on every rank together
                print(f"[rank {rank}] xGMI all-reduce unavailable: {e!r}", file=sys.stderr, flush=True)
            f = torch.tensor([0.0 if p2p is not None else 1.0], device=device)
            torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.MAX)
            if float(f.item()) == 0.0:
                p2p_ok = p2p.validate(1 << 20)
            if p2p_ok:
                dp = p2p
            else:
                a.comm = "torch"
                if rank == 0:
                    print("xGMI all-reduce failed its validation against RCCL: using RCCL", file=sys.stderr, flush=True)

    from distributeddeeplearning_amd import benchmarks
    from distributeddeeplearning_amd.parallel import selfcheck

    # N > 1: prove the communicator spans N ranks, then let the captured-collective self-test pick the graph mode
    seen = selfcheck.ranks_seen(dp, device)
    captured_ok = None
    if a.graph == "auto":
        if dp is None or not on_gpu:
            a.graph = "on" if on_gpu else "off"
        elif getattr(dp, "captures_safely", False):
            a.graph = "full"             # the xGMI kernel is an ordinary launch: nothing of RCCL inside the graph
        else:
            captured_ok = selfcheck.captured_collective_ok(dp, device, verbose=True)
            a.graph = "full" if captured_ok else "on"
    if dp is None:
        backend = "none"
    elif a.comm == "rccl" and on_gpu:
        backend = "rccl-native"
    elif a.comm == "p2p" and on_gpu:
        backend = "xgmi-p2p all-reduce + fused optimizer (bootstrap torch.distributed/" + torch.distributed.get_backend() + ")"
    else:
        backend = "torch.distributed/" + torch.distributed.get_backend()

    def build_runner():
        r = benchmarks.make(a.config, device, dp, rank, a)
        r.setup()
        return r

    # hipGraph capture next to a live RCCL group is the one step of setup that can fail for reasons outside the
    # model code (utils/hipgraph.CaptureError, a communicator error); if it does on ANY rank, every rank steps down
    # together (full -> on -> off: the measurement stays valid, only slower) and the record says why
    # (`graph_fallback_reason`).  Anything else -- a programming error -- is raised on every rank: a fallback must
    # never turn a bug into a slower-but-green run.
    from distributeddeeplearning_amd.utils.hipgraph import CaptureError
    soft = (CaptureError, torch.distributed.DistBackendError) if hasattr(torch.distributed, "DistBackendError") \
        else (CaptureError,)
    chain = {"full": ["full", "on", "off"], "on": ["on", "off"], "off": ["off"]}[a.graph] if on_gpu else [a.graph]
    runner = None
    fallback_reason = None
    for k, mode in enumerate(chain):
        a.graph = mode
        err, code = None, 0.0
        try:
            runner = build_runner()
        except soft as e:
            err, code = e, 1.0
        except Exception as e:          # noqa: BLE001 - agreed on with the other ranks, then re-raised everywhere
            err, code = e, 2.0
        if world > 1:
            f = torch.tensor([code], device=device)
            torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.MAX)
            code = float(f.item())
        if code == 0.0:
            break
        runner = None
        if code == 2.0 or k + 1 == len(chain):
            raise err if err is not None else RuntimeError("setup failed on another rank")
        fallback_reason = f"--graph {mode}: " + (repr(err) if err is not None else "capture failed on another rank")
        print(f"[rank {rank}] setup with --graph {mode} failed ({err!r}); falling back to --graph {chain[k + 1]}",
              file=sys.stderr, flush=True)
        if on_gpu:
            torch.cuda.synchronize()

    def sync():
        if world > 1:
            torch.distributed.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        runner.step()
    if on_gpu:          # the graphed runners persisted their tuned plans before capture; eager ones do it here
        from distributeddeeplearning_amd.ops import functional as F
        F.hip_backend().save_choice_cache()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        runner.step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    info = runner.describe()
    images = info["images_per_step_per_rank"] * world * a.steps
    if rank == 0:
        out = {"metric": info["metric"], "value": images / dt, "unit": "images/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16" if on_gpu else "fp32", "data": "synthetic", "config": info["config"]}
        out["config"]["parallelism"] = f"dp{world}"
        out["config"]["global_batch"] = info["images_per_step_per_rank"] * world
        out["config"]["graph"] = a.graph
        out["rccl_ranks_seen"] = seen
        out["graph_mode_used"] = a.graph
        out["graph_fallback_reason"] = fallback_reason
        out["captured_collective_selftest"] = captured_ok
        out["comm_backend"] = backend
        out.update(info.get("extra", {}))
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
