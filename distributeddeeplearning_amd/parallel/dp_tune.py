"""Start-up selection of the data-parallel step by MEASUREMENT, agreed across ranks.

The development box has one GPU; the scaling run has eight, joined by xGMI links nothing here has ever timed.  So the
job decides on the real machine, before its timed region, from candidates that are all correct by construction:

  * the wire: RCCL (torch.distributed "nccl") or the xGMI direct all-reduce kernel with the optimizer in its epilogue
    (csrc/comm/p2p_allreduce.hip).  The kernel is first exercised in CHILD processes -- one per rank on the same
    device, their own gloo group, the same IPC mappings (`probe_p2p`) -- so a fault in a code path that has never
    crossed a real link costs a child, not the job; then it is created and validated against RCCL in-process;
  * the step shape of the DCGAN trainer: merged generator pass or a G-phase forward of its own (which hides D's
    last gradient bucket), bucketed all-reduces under the backward pass or one collective per network.

Every candidate is captured, timed for a few steps between barriers, the per-rank times are MAX-reduced (so all
ranks read the same numbers) and the fastest wins; a candidate that fails to build on ANY rank is dropped on all.
The training state is restored afterwards: tuning steps do not count.  The record carries what was chosen and what
it cost (`comm_chosen`, `dp_variant`, `allreduce_us_{D,G}`, `exposed_comm_us_per_step`).

Replaces the exchange sites /root/reference/DDL_Ray3.py:186-205 (driver-side `ray.get` of every actor's result) by
collectives whose cost the job measures and hides.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


# ------------------------------------------------------------------------------------------------ agreement helpers
def _flag_device(device):
    on_dev = dist.is_initialized() and dist.get_backend() == "nccl"
    return torch.device(device) if on_dev else torch.device("cpu")


def agree_max(value: float, device) -> float:
    """MAX over ranks of a per-rank number (the same value on every rank afterwards)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_flag_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _sync(device):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()


def time_calls(fn, n: int, device) -> float:
    """Seconds per call of `fn`, `n` calls between barrier + device synchronisation, MAX over ranks."""
    _sync(device)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    _sync(device)
    return agree_max((time.perf_counter() - t0) / n, device)


def allreduce_latency_us(dp, numel: int, device, iters: int = 30) -> float:
    """One eager all-reduce of a flat fp32 buffer of `numel` elements through `dp`, microseconds (MAX over ranks)."""
    numel = max(8, numel - numel % 8)
    buf = torch.zeros(numel, dtype=torch.float32, device=device)
    for _ in range(5):
        dp.all_reduce_sum_(buf)
    return time_calls(lambda: dp.all_reduce_sum_(buf), iters, device) * 1e6


# ------------------------------------------------------------------------------------------------ xGMI kernel probe
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def probe_p2p(device, capacity_elems: int, timeout_s: float = 90.0, verbose: bool = False) -> bool:
    """Run `parallel/p2p_probe.py` as a child of every rank (same device, a gloo group of their own) and agree on the
    verdict.  True iff every child mapped its peers, validated one- and two-shot all-reduces and the fused update
    against host sums, stayed healthy and exited cleanly.  A collective: every rank calls it."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    box = [_free_port() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(box[0]),
               HSA_ENABLE_IPC_MODE_LEGACY="0", DDL_PROBE_DEVICE=str(torch.device(device).index or 0),
               DDL_PROBE_CAPACITY=str(int(capacity_elems)))
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "GROUP_RANK",
              "ROLE_RANK", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    rc, tail = 1, ""
    try:
        r = subprocess.run([sys.executable, "-m", "distributeddeeplearning_amd.parallel.p2p_probe"], env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                           capture_output=True, text=True, timeout=timeout_s)
        rc, tail = r.returncode, (r.stdout + r.stderr)[-600:]
    except Exception as e:      # noqa: BLE001 - a probe that cannot run is a probe that failed
        tail = repr(e)
    if verbose and rc != 0:
        print(f"[rank {rank}] xGMI all-reduce probe failed (rc={rc}): {tail}", file=sys.stderr, flush=True)
    return agree_max(0.0 if rc == 0 else 1.0, device) == 0.0


def build_p2p(base, device, capacity_elems: int, probe: bool = True, verbose: bool = False):
    """The xGMI communicator over `base`, or None (on every rank together) when the probe, the construction or the
    validation against `base` fails anywhere."""
    from .p2p import P2PAllReduce
    rank = getattr(base, "rank", 0)
    if probe and base.world > 1 and not probe_p2p(device, capacity_elems, verbose=verbose):
        return None
    p2p = None
    try:
        # generous in-kernel time-out: ranks can be seconds apart at the first call after set-up (autotuning)
        p2p = P2PAllReduce(base, device, capacity_elems, timeout_s=30.0)
    except Exception as e:      # noqa: BLE001 - the constructor raises on every rank together (parallel/p2p.py)
        if verbose:
            print(f"[rank {rank}] xGMI all-reduce unavailable: {e!r}", file=sys.stderr, flush=True)
        return None
    if not p2p.validate(min(capacity_elems, 1 << 20)):
        if verbose and rank == 0:
            print("xGMI all-reduce failed its validation against RCCL", file=sys.stderr, flush=True)
        p2p.close()
        return None
    return p2p


# ------------------------------------------------------------------------------------------------ GAN step tuner
def _snapshot(t):
    snap = {"GP": t.G.P.clone(), "GS": t.G.S.clone(), "DP": t.D.P.clone(), "DS": t.D.S.clone(),
            "rng": {k: (v.clone() if torch.is_tensor(v) else v) for k, v in t.rng.state_dict().items()}}
    for name, opt in (("oG", t.optG), ("oD", t.optD)):
        snap[name] = {"t": opt.t, "m": None if opt.m is None else opt.m.clone(),
                      "v": None if opt.v is None else opt.v.clone(),
                      "step": None if opt.step_dev is None else opt.step_dev.clone()}
    return snap


def _restore(t, snap):
    t.G.P.copy_(snap["GP"]); t.G.S.copy_(snap["GS"]); t.D.P.copy_(snap["DP"]); t.D.S.copy_(snap["DS"])
    t.rng.load_state_dict(snap["rng"])
    for name, opt in (("oG", t.optG), ("oD", t.optD)):
        s = snap[name]
        opt.t = s["t"]
        if opt.m is not None:
            opt.m.copy_(s["m"])
        if opt.v is not None:
            opt.v.copy_(s["v"])
        if opt.step_dev is not None:
            opt.step_dev.copy_(s["step"])
    t.G.refresh_working_copies()
    t.D.refresh_working_copies()
    t.D.release()
    t.G.release()


def pretune_gan(trainer, real_prepared):
    """Kernel-plan autotuning (ops/hip.py) before any data-parallel step: ONE training step with every collective
    skipped, so plan timing is not disturbed by communication kernels waiting for a peer that is itself busy tuning;
    the training state is restored and every rank then adopts rank 0's plans.  Until the plans are common and fixed,
    two ranks may run different (equally correct) kernels for the same layer -- harmless for training, since
    gradients are summed, but the arithmetic of a step only becomes a pure function of its inputs afterwards."""
    from ..ops import functional as F
    snap = _snapshot(trainer)
    keep = trainer.comm_off
    trainer.comm_off = True
    try:
        trainer.train_step(real_prepared, prepared=True)
    finally:
        trainer.comm_off = keep
    if trainer.device.type == "cuda":
        torch.cuda.synchronize()
    _restore(trainer, snap)
    if trainer.G.on_gpu:
        F.hip_backend().sync_choices(trainer.dp)


def tune_gan_step(trainer, real_prepared, comms: Dict[str, object], full_ok: Dict[str, bool], steps: int = 20,
                  warmup: int = 2, verbose: bool = False):
    """Pick (communicator, merged, buckets) for `trainer` by timing captured steps; returns (GraphedGANStep, report).

    `comms`: name -> communicator (insertion order = preference on ties); `full_ok[name]`: may its collectives be
    captured into the step graph (RCCL: the captured-collective self-test; the xGMI kernel: always).  On return the
    trainer is configured with the winner, its training state is what it was on entry, and the returned step object
    is captured and warm."""
    from ..models.dcgan import GraphedGANStep
    device = trainer.device
    can_merge = trainer.merged
    if can_merge:                       # plans of both generator-pass shapes, tuned without communication
        trainer.merged = False
        pretune_gan(trainer, real_prepared)
        trainer.merged = True
    pretune_gan(trainer, real_prepared)
    snap = _snapshot(trainer)
    cands: List[dict] = []
    for name, dp in comms.items():
        for merged in ((True, False) if can_merge else (False,)):
            for buckets in ((True, False) if full_ok.get(name, False) else (False,)):
                cands.append({"comm": name, "merged": merged, "buckets": buckets})
    timings, best, best_t = [], None, None

    def build(c, comm_off=False):
        trainer.set_comm(comms[c["comm"]])
        trainer.merged, trainer.buckets, trainer.comm_off = c["merged"], c["buckets"], comm_off
        trainer._fake = None
        return GraphedGANStep(trainer, real_prepared, warmup=warmup, capture_collectives=full_ok.get(c["comm"], False))

    dropped = []

    def attempt(c, comm_off=False):
        """Build + time one candidate.  A candidate that cannot be built on ANY rank is dropped on all of them, whatever
        the reason (the eight-GPU machine is where this code meets real links for the first time: the job must end on
        the candidates that work, and the record says which did not and why)."""
        step, err = None, None
        try:
            step = build(c, comm_off)
        except Exception as e:      # noqa: BLE001 - recorded in `dp_candidates_dropped`, agreed on below
            err = e
        failed = agree_max(0.0 if err is None else 1.0, device) > 0.0
        if failed:
            dropped.append(dict(c, comm_off=comm_off, error=repr(err) if err is not None else "failed on another rank"))
            if verbose:
                print(f"[dp_tune] candidate {c} dropped: {err!r}", file=sys.stderr, flush=True)
            if trainer.G.on_gpu:
                torch.cuda.synchronize()
            return None, None
        for _ in range(3):
            step.step()
        return step, time_calls(step.step, steps, device)

    for c in cands:
        step, dt = attempt(c)
        _restore(trainer, snap)
        if step is None:
            continue
        rec = dict(c, us_per_step=round(dt * 1e6, 2))
        timings.append(rec)
        if best_t is None or dt < best_t * 0.995:       # ties (0.5 %) go to the earlier, preferred candidate
            best, best_t = c, dt
        del step
        torch.cuda.synchronize()
    if best is None:
        raise RuntimeError("no data-parallel step candidate could be captured")
    # what the collectives cost on the critical path: the winner again with every collective skipped
    exposed = None
    step, dt0 = attempt(best, comm_off=True)
    _restore(trainer, snap)
    if step is not None:
        exposed = round((best_t - dt0) * 1e6, 2)
        del step
        torch.cuda.synchronize()
    dp = comms[best["comm"]]
    trainer.set_comm(dp)
    lat = {}
    for tag, net in (("D", trainer.D), ("G", trainer.G)):
        lat[f"allreduce_us_{tag}"] = round(allreduce_latency_us(dp, net.G.numel(), device), 2)
        if best["buckets"]:
            lat[f"allreduce_us_{tag}_buckets"] = [round(allreduce_latency_us(dp, hi - lo, device), 2)
                                                  for lo, hi, _ in trainer._bk[id(net)]]
    final = build(best)
    report = {"comm_chosen": best["comm"],
              "dp_variant": {"merged_generator_pass": best["merged"], "bucketed_allreduce": best["buckets"],
                             "fused_optimizer_in_allreduce": bool(trainer.fused),
                             "buckets_D": [hi - lo for lo, hi, _ in trainer._bk[id(trainer.D)]] if best["buckets"] else None,
                             "buckets_G": [hi - lo for lo, hi, _ in trainer._bk[id(trainer.G)]] if best["buckets"] else None},
              "dp_candidates_us": timings, "dp_candidates_dropped": dropped, "exposed_comm_us_per_step": exposed,
              "step_us_without_collectives": None if exposed is None else round(dt0 * 1e6, 2), **lat}
    return final, report
