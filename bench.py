#!/usr/bin/env python3
"""Headline benchmark: DCGAN 28x28 (GAN_distributed graphs) data-parallel training, bf16, images/s.

    python bench.py --gpus N --steps K --warmup W            (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = the reference iteration (/root/reference/GAN_distributed.py:133-135): a D update on
[B real ; B generated] and a G update on B noise vectors, Adam(2e-4), per rank, gradients averaged over the ranks
every step.  B = 1000 per GPU as in the reference (weak scaling: the global batch grows with N).
`value` = real images consumed per second by the whole job.
Other BASELINE.json configs: --config dcgan64 | classifier | localsgd | ensemble32.

Under N > 1 the job chooses its data-parallel step by measurement before the timed region
(distributeddeeplearning_amd/parallel/dp_tune.py): RCCL or the xGMI direct all-reduce kernel (probed in child
processes first, validated against RCCL, health-checked after warm-up and after the timed loop), merged or split
generator pass, bucketed or flat gradient all-reduce; the record says what was chosen and what it cost
(`comm_chosen`, `dp_variant`, `allreduce_us_{D,G}`, `exposed_comm_us_per_step`).
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
    ap.add_argument("--comm", default="auto", choices=["auto", "torch", "rccl", "p2p"],
                    help="gradient all-reduce through torch.distributed (nccl == RCCL), the native C++ RCCL "
                         "communicator, or the xGMI direct all-reduce kernel with the optimizer fused into its "
                         "epilogue (csrc/comm/p2p_allreduce.hip).  auto (dcgan configs): both are built, the kernel is "
                         "validated against RCCL, and the faster one on the real step is used")
    ap.add_argument("--dp-variant", default="auto", choices=["auto", "default"],
                    help="dcgan28 under data parallelism: auto = time {merged, split generator pass} x {bucketed, "
                         "flat all-reduce} x communicators at start-up and keep the fastest; default = merged + "
                         "bucketed on the communicator --comm names (RCCL for auto)")
    ap.add_argument("--no-probe", action="store_true",
                    help="skip the child-process probe of the xGMI kernel (N > 1 only; it costs a few seconds)")
    ap.add_argument("--avg-every", type=int, default=8, help="local-SGD averaging period (config localsgd)")
    ap.add_argument("--zero", action="store_true",
                    help="dcgan configs, N>1: reduce-scatter + sharded update + all-gather instead of all-reduce")
    ap.add_argument("--force-dp", action="store_true",
                    help="diagnostics: run the data-parallel code path (collectives, start-up selection) on a "
                         "1-rank process group, to price its overhead without a second GPU")
    ap.add_argument("--device", default="cuda")
    a = ap.parse_args()

    from distributeddeeplearning_amd.parallel import ddp, dp_tune
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

    # ---- communicators ------------------------------------------------------------------------------------------
    base, p2p, base_name = None, None, "none"
    if world > 1 or a.force_dp:
        cd = torch.bfloat16 if a.comm_dtype == "bf16" else None
        if a.comm == "rccl" and on_gpu:
            from distributeddeeplearning_amd.parallel.rccl import RcclCommunicator
            base, base_name = RcclCommunicator.from_env(device, comm_dtype=cd), "rccl-native"
        else:
            base = ddp.FlatDataParallel(comm_dtype=cd)
            base_name = "torch.distributed/" + torch.distributed.get_backend()
        want_p2p = on_gpu and not a.zero and (a.comm == "p2p" or (a.comm == "auto" and a.config in ("dcgan28", "dcgan64")))
        if want_p2p:        # 16 MB of fp32 staging: the largest flat gradient (64x64 generator) is 12.8 MB
            p2p = dp_tune.build_p2p(base, device, 1 << 22, probe=not a.no_probe, verbose=True)
            if p2p is None and rank == 0:
                print("xGMI all-reduce not available / not validated: staying on RCCL", file=sys.stderr, flush=True)
    p2p_name = "xgmi-p2p all-reduce + fused optimizer (bootstrap " + base_name + ")"

    from distributeddeeplearning_amd import benchmarks
    from distributeddeeplearning_amd.parallel import selfcheck
    from distributeddeeplearning_amd.utils.hipgraph import CaptureError

    # N > 1: prove the communicator spans N ranks, then let the captured-collective self-test pick the graph mode
    seen = selfcheck.ranks_seen(base, device)
    captured_ok = None
    if a.graph == "auto":
        if base is None or not on_gpu:
            a.graph = "on" if on_gpu else "off"
        else:
            captured_ok = selfcheck.captured_collective_ok(base, device, verbose=True)
            a.graph = "full" if captured_ok else "on"
    graph_wanted = a.graph

    def sync():
        if world > 1:
            torch.distributed.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    def build_runner(use_p2p: bool):
        """dp / comms for this attempt; dcgan configs may tune between the candidates in `a.comms`."""
        a.comms, a.full_ok = {}, {}
        if base is not None:
            if a.comm != "p2p" or p2p is None or not use_p2p:
                a.comms[base_name], a.full_ok[base_name] = base, a.graph == "full"
            if p2p is not None and use_p2p:
                a.comms[p2p_name], a.full_ok[p2p_name] = p2p, True
        dp = None
        if base is not None:
            dp = p2p if (a.comm == "p2p" and p2p is not None and use_p2p) else base
        r = benchmarks.make(a.config, device, dp, rank, a)
        r.setup()
        return r

    # hipGraph capture next to a live RCCL group is the one step of setup that can fail for reasons outside the
    # model code (utils/hipgraph.CaptureError, a communicator error); if it does on ANY rank, every rank steps down
    # together (full -> on -> off: the measurement stays valid, only slower) and the record says why
    # (`graph_fallback_reason`).  Anything else -- a programming error -- is raised on every rank: a fallback must
    # never turn a bug into a slower-but-green run.
    soft = (CaptureError, torch.distributed.DistBackendError) if hasattr(torch.distributed, "DistBackendError") \
        else (CaptureError,)

    def setup_with_fallback(use_p2p: bool):
        chain = {"full": ["full", "on", "off"], "on": ["on", "off"], "off": ["off"]}[a.graph] if on_gpu else [a.graph]
        reason = None
        for k, mode in enumerate(chain):
            a.graph = mode
            err, code, runner = None, 0.0, None
            try:
                runner = build_runner(use_p2p)
            except soft as e:
                err, code = e, 1.0
            except Exception as e:          # noqa: BLE001 - agreed on with the other ranks, then re-raised everywhere
                err, code = e, 2.0
            code = dp_tune.agree_max(code, device)
            if code == 0.0:
                return runner, reason
            if code == 2.0 or k + 1 == len(chain):
                raise err if err is not None else RuntimeError("setup failed on another rank")
            reason = f"--graph {mode}: " + (repr(err) if err is not None else "capture failed on another rank")
            print(f"[rank {rank}] setup with --graph {mode} failed ({err!r}); falling back to --graph {chain[k + 1]}",
                  file=sys.stderr, flush=True)
            if on_gpu:
                torch.cuda.synchronize()

    def p2p_in_use(runner):
        dp = getattr(getattr(runner, "t", None), "dp", None) or getattr(runner, "dp", None)
        return dp if (dp is not None and hasattr(dp, "healthy_everywhere")) else None

    # The xGMI kernel answers a time-out (a peer that never signals) with a sticky status word and from then on skips
    # every reduction AND the fused update: a dead communicator runs FASTER than a healthy one.  So its health is
    # read, and agreed across ranks, after the warm-up and after the timed loop; an unhealthy run is thrown away and
    # repeated on RCCL.
    p2p_healthy, p2p_dropped = None, None
    for attempt, use_p2p in enumerate((True, False)):
        a.graph = graph_wanted
        runner, fallback_reason = setup_with_fallback(use_p2p and p2p is not None)
        for _ in range(a.warmup):
            runner.step()
        if on_gpu:          # the graphed runners persisted their tuned plans before capture; eager ones do it here
            from distributeddeeplearning_amd.ops import functional as F
            F.hip_backend().save_choice_cache()
        live = p2p_in_use(runner)
        if live is not None and not live.healthy_everywhere():
            p2p_healthy, p2p_dropped = False, "a wait inside the xGMI kernel timed out during warm-up"
            continue
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            runner.step()
        sync()
        dt = time.perf_counter() - t0
        if live is not None:
            p2p_healthy = live.healthy_everywhere()
            if not p2p_healthy:
                p2p_dropped = "a wait inside the xGMI kernel timed out during the timed loop"
                continue
        break
    else:
        raise RuntimeError("the run did not complete on either communicator")
    if p2p_dropped is not None and rank == 0:
        print(f"xGMI all-reduce dropped ({p2p_dropped}); the record below was measured on RCCL", file=sys.stderr, flush=True)
    dt = dp_tune.agree_max(dt, device)
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
        live = p2p_in_use(runner)
        out["comm_backend"] = "none" if base is None else (p2p_name if live is not None else base_name)
        out["p2p_validated"] = None if base is None else (p2p is not None)
        out["p2p_healthy"] = p2p_healthy
        out["p2p_dropped"] = p2p_dropped
        out.update(info.get("extra", {}))
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
