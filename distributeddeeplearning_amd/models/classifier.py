"""`RandomModel`: one ensemble member = a network, its optimizer/loss, its private data split.

Behaviour of /root/reference/DDL_Ray3.py:125-161 and DDL_Normal4.py:126-177:
  * `train()`        : up to epochs/mini_epochs rounds of `fit(epochs=mini_epochs)` (batch 32,
                       reshuffled every epoch, partial last batch kept), prints `i [loss_e1, loss_e2]`,
                       early stop when `1 - loss_last/loss_first < early_stop_rate`;
  * `predict(x)`     : softmax probabilities;
  * `get_accuracy()` : argmax agreement on the member's own validation split or a given (x, y);
  * `get_model()`    : (spec, weights) -- what a transfer child or another rank needs;
  * `save()`         : models/model<N>/{architecture.json, weights.h5, graph.*}.

Device execution: the dataset is resident in HBM, a batch is formed by one gather kernel from an
index slice, and the whole train step (gather -> forward -> fused softmax-CE -> backward -> fused
optimizer) is replayed from a hipGraph, so the reference's launch-bound batch-32 regime costs one
graph launch per step.  Optional data parallelism: flat gradient all-reduce every step, or
periodic weight averaging (local SGD) through `averager`.
"""
from __future__ import annotations

import math
import os
import time
from typing import Dict, List, Optional, Tuple

import torch

from ..config import SearchSpace
from ..ops import functional as F
from ..utils import hipgraph
from ..utils.data import Data, Sample
from .network import Network
from .optim import FlatOptimizer
from .search import CompiledSpec
from ..utils.streams import comm_stream


class RandomModel:
    def __init__(self, data: Data, sample: Sample, space: SearchSpace, compiled: CompiledSpec,
                 device="cpu", number: Optional[int] = None, seed: int = 0,
                 parent_state: Optional[Dict] = None, inherited: Optional[set] = None,
                 dp=None, averager=None, use_graph: Optional[bool] = None, verbose: bool = True):
        self.data, self.sample, self.space, self.compiled = data, sample, space, compiled
        self.device = torch.device(device)
        self.number = number
        self.verbose = verbose
        self.dp, self.averager = dp, averager
        self.net = Network(compiled.spec, device, seed)
        self.net.set_no_input_grad()    # a classifier never needs the gradient of its input images
        # wgrad chain as a parallel graph branch (network.SideBranch): +2 % on the 5-replica step at batch 512, but at
        # the reference's batch 32 the fork / join edges cost more than they hide (10 MLP members: 66.9 -> 44.4 ms
        # per 32 steps without it) and they keep members on different streams from overlapping (-> 22.3 ms)
        self.net.branch_wgrad = space.batch_size >= 128
        if parent_state is not None:
            # clone_model + set_weights(get_weights()): inherited layers start from the parent's
            # weights (BN moving statistics included); the new head keeps its fresh init.
            keep = {k: v for k, v in parent_state.items() if inherited is None or k[0] in inherited}
            self.net.load_logical_state(keep, strict=False)
        self.opt = FlatOptimizer(self.net, compiled.optimizer)
        self.loss_name = compiled.loss
        self.gen = torch.Generator().manual_seed(seed + 7)
        self.history: List[List[float]] = []
        self.use_graph = (self.device.type == "cuda") if use_graph is None else use_graph
        self._graph = None
        self._gB = None
        if averager is not None:
            averager.attach(self.net)

    # ------------------------------------------------------------------ one optimisation step
    def _loss(self, logits, y, want_grad=True, index=None):
        if self.loss_name == "mean_squared_error":
            return F.softmax_mse(logits, y, want_grad, index=index)
        return F.softmax_ce(logits, y, want_grad, index=index)

    def _fwd_bwd(self, idx):
        net = self.net
        x = net.prepare_input(self.data.train_x, index=idx)
        logits = net.forward(x, training=True)
        # the minibatch's labels are read through the same index vector inside the loss kernel (no gather pass)
        loss, dlogits, _ = self._loss(logits, self.data.train_y, index=idx)
        net.backward(dlogits)
        return loss

    def _update(self):
        if self._fused_dp:
            pass                                          # done by the fused all-reduce kernel (_reduce)
        elif self._sync_dp:
            self.opt.step(self.dp.grad_scale)
        else:
            self.opt.step()
        self.net.release()

    def _reduce(self):
        """Synchronous DP: ONE flat collective per step -- or one fused all-reduce + update kernel (parallel/p2p.py)."""
        if self._fused_dp:
            self.opt.step_allreduce_fused(self.dp)
        else:
            self.dp.all_reduce_sum_(self.net.G)

    @property
    def _fused_dp(self):
        return (self.dp is not None and self.averager is None and self.net.on_gpu
                and getattr(self.dp, "supports_fused_update", False))

    @property
    def _sync_dp(self):
        return self.dp is not None and self.dp.world > 1 and self.averager is None

    def _step_body(self, idx):
        loss = self._fwd_bwd(idx)
        if self._sync_dp or self._fused_dp:
            self._reduce()
        self._update()
        return loss

    def _graphed_step(self, idx):
        """hipGraph replay of the step for the steady-state batch shape.  Single rank / local SGD:
        one graph.  Synchronous DP: two graph segments with the flat all-reduce issued between."""
        B = idx.numel()
        if self._graph is None:
            self._idx = idx.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):        # warm-up really trains (first step of the epoch)
                first = self._step_body(self._idx)
            torch.cuda.current_stream().wait_stream(side)
            hipgraph.quiesce()
            F.hip_backend().save_choice_cache()
            pool = torch.cuda.graph_pool_handle()
            if self._sync_dp and not getattr(self.dp, "captures_safely", False):
                ga, self._static_loss = hipgraph.capture(lambda: self._fwd_bwd(self._idx), pool)
                gb, _ = hipgraph.capture(self._update, pool)
                self._graph = (ga, gb)
            else:
                g, self._static_loss = hipgraph.capture(lambda: self._step_body(self._idx), pool)
                self._graph = (g,)
            self._gB = B
            return first                         # capture records, it does not execute
        if B != self._gB:
            return self._step_body(idx)          # ragged last batch of an epoch runs eagerly
        return self._replay(idx)

    def _replay(self, idx):
        self._idx.copy_(idx)
        self._graph[0].replay()
        if len(self._graph) == 2:
            self.dp.all_reduce_sum_(self.net.G)
            self._graph[1].replay()
        return self._static_loss

    def train_step(self, idx):
        if self.use_graph:
            loss = self._graphed_step(idx)
        else:
            loss = self._step_body(idx)
        if self.averager is not None:
            self.averager.after_step()
        return loss

    # ------------------------------------------------------------------ multi-step graphs
    # At the reference's batch of 32 (Keras fit() default, /root/reference/DDL.py:92) one optimizer step is tens of
    # microseconds of tiny kernels; replaying one graph PER step leaves the host (slice the permutation, copy the
    # index batch, launch, accumulate the loss) as the bottleneck.  `chunk` steps are therefore captured into ONE
    # graph that reads its index batches from a static [chunk, B] tensor and adds `loss * B` into a device-side
    # accumulator: an epoch of 1563 steps is ~49 graph launches and no per-step host work at all.
    CHUNK = int(os.environ.get("DDL_CHUNK", "32"))        # 0 / 1: one graph per step (round-1 behaviour)

    def _chunk_ok(self):
        return self.use_graph and self.dp is None and self.averager is None and self.device.type == "cuda"

    def prepare(self, batch_size: Optional[int] = None):
        """Autotune + capture the step graphs WITHOUT training: one eager step on the first indices, the captures,
        then parameters, optimizer state and BN statistics are restored.  Lets a scheduler prepare many members one
        after another (plan autotuning times kernels on an otherwise idle GPU) and train them concurrently."""
        if not self._chunk_ok() or self._graph is not None:
            return
        B = batch_size or self.space.batch_size
        n = self.sample.train_idx.numel()
        if n < B:
            return
        net, opt = self.net, self.opt
        keep = [t.clone() for t in (net.P, net.S)] + [None if t is None else t.clone() for t in (opt.m, opt.v)]
        keep_step, keep_t = opt.step_dev.clone(), opt.t
        idx = self.sample.train_idx[:B].contiguous()
        self._graphed_step(idx)                                      # eager warm-up step + single-step capture
        S = min(self.CHUNK, n // B)
        if S >= 2:
            self._idx_chunk = torch.zeros(S, B, dtype=torch.int64, device=self.device)
            self._idx_chunk.copy_(self.sample.train_idx[:S * B].view(S, B))
            self._loss_acc = torch.zeros((), dtype=torch.float32, device=self.device)

            def body():
                for k in range(S):
                    loss = self._step_body(self._idx_chunk[k])
                    self._loss_acc.add_(loss.detach().float() * B)
            hipgraph.quiesce()
            self._chunk_graph, _ = hipgraph.capture(body, torch.cuda.graph_pool_handle())
            self._chunk_S = S
            # one timed replay (state is restored below): the scheduler balances concurrent members by it
            torch.cuda.current_stream().synchronize()
            t0 = time.perf_counter()
            self._chunk_graph.replay()
            torch.cuda.current_stream().synchronize()
            self.chunk_ms = (time.perf_counter() - t0) * 1e3
        net.P.copy_(keep[0])
        net.S.copy_(keep[1])
        for t, k in ((opt.m, keep[2]), (opt.v, keep[3])):
            if t is not None:
                t.copy_(k)
        opt.step_dev.copy_(keep_step)
        opt.t = keep_t
        net.refresh_working_copies()
        torch.cuda.current_stream().synchronize()

    # ------------------------------------------------------------------ fit / train
    # fit / train are written as generators: they yield after every batch of launches they issue (None) and, when
    # they need a value back from the device, a recorded event -- `train()` simply waits on it, while
    # `train_concurrently` keeps issuing other members' work on their own streams and comes back once it is done.
    def _fit_gen(self, epochs: int, batch_size: Optional[int] = None):
        B = batch_size or self.space.batch_size
        tr = self.sample.train_idx
        n = tr.numel()
        on_gpu = self.device.type == "cuda"
        if self._chunk_ok():
            self.prepare(B)
        S = getattr(self, "_chunk_S", 0) if (self._chunk_ok() and getattr(self, "_gB", None) == B) else 0
        losses = []
        for _ in range(epochs):
            perm = tr[torch.randperm(n, generator=self.gen).to(tr.device)]
            acc = torch.zeros((), dtype=torch.float32, device=self.device)
            i = 0
            if S:
                self._loss_acc.zero_()
                while i + S * B <= n:
                    self._idx_chunk.copy_(perm[i:i + S * B].view(S, B))
                    self._chunk_graph.replay()
                    i += S * B
                    yield None
                acc = acc + self._loss_acc
            k = 0
            while i < n:
                idx = perm[i:i + B]
                loss = self.train_step(idx)
                acc = acc + loss.detach() * idx.numel()      # Keras: sample-weighted epoch mean
                i += B
                k += 1
                if k % 16 == 0:
                    yield None
            losses.append(acc / n)
        if self.dp is not None and self.dp.world > 1:
            # every rank must take the same early-stop decision or the ones that stay would wait forever in
            # their next collective: the decision is made on the rank-averaged epoch losses
            t = torch.stack(losses).float()
            self.dp.all_reduce_mean_(t)
            losses = list(t)
        if on_gpu:
            ev = torch.cuda.Event()
            ev.record()
            yield ev
        return [float(v) for v in losses]

    def _train_gen(self):
        sp = self.space
        for i in range(int(sp.epochs / sp.mini_epochs)):
            hist = yield from self._fit_gen(sp.mini_epochs)
            self.history.append(hist)
            if self.verbose:
                print(i, hist, flush=True)
            # the reference's test, literally (DDL.py:95): with a NaN loss `1 - nan < rate` is False and it trains on
            if hist[0] == 0 or 1 - (hist[-1] / hist[0]) < sp.early_stop_rate:
                break
        if self.averager is not None:
            self.averager.finish()

    @staticmethod
    def _drive(gen):
        try:
            while True:
                ev = next(gen)
                if ev is not None:
                    ev.synchronize()
        except StopIteration as stop:
            return stop.value

    def fit(self, epochs: int, batch_size: Optional[int] = None) -> List[float]:
        return self._drive(self._fit_gen(epochs, batch_size))

    def train(self):
        self._drive(self._train_gen())

    def sync_bn_stats(self):
        """Synchronous DP exchanges gradients only: average the BN moving statistics over the ranks before they
        are read by anything rank-dependent (evaluation, export).  A collective: every rank calls it."""
        if self.dp is not None and self.dp.world > 1 and self.averager is None:
            self.dp.all_reduce_mean_(self.net.S)

    # ------------------------------------------------------------------ inference
    def predict(self, x=None, index: Optional[torch.Tensor] = None, batch_size: int = 4096) -> torch.Tensor:
        """Softmax outputs [N,10] (fp32, on device).  `x` is a resident uint8 image tensor (default:
        the training pool) optionally gathered by `index`."""
        x = self.data.train_x if x is None else x
        n = x.shape[0] if index is None else index.numel()
        outs = []
        for i in range(0, n, batch_size):
            if index is None:
                xb = self.net.prepare_input(x[i:i + batch_size])
            else:
                xb = self.net.prepare_input(x, index=index[i:i + batch_size])
            y = self.net.forward(xb, training=False)
            outs.append(F.softmax(y, self.net.out_shape[-1]))
        return torch.cat(outs, 0)

    def get_accuracy(self, validation_data: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> float:
        self.sync_bn_stats()
        if validation_data is None:
            idx = self.sample.val_idx
            if idx.numel() == 0:
                return float("nan")
            pred = self.predict(self.data.train_x, index=idx)
            target = self.data.train_y.index_select(0, idx)
        else:
            x, target = validation_data
            pred = self.predict(x)
        correct = F.argmax_match_count(pred, target)
        return int(correct) / pred.shape[0]

    def get_model(self):
        return self.compiled, self.net.logical_state()

    def save(self, directory: Optional[str] = None):
        from ..utils import ckpt
        self.sync_bn_stats()
        d = directory or self.space.model_save_directory
        return ckpt.save_model(self.net, f"{d}/model{self.number}")


class ReplicaGroupStep:
    """One synchronous data-parallel step of SEVERAL networks (BASELINE config 2: the five replicas) with the
    flat gradient all-reduces taken off the critical path.

    Issued one network at a time, a step is fwd/bwd_0, AR_0, update_0, fwd/bwd_1, AR_1, ...: five exposed
    collectives, each a latency-bound 0.2-3.5 MB message on xGMI.  Here network i's all-reduce runs on a
    communication stream while network i+1 computes, and the optimizer updates follow once at the end:

        main:  A_0 | A_1 | A_2 | A_3 | A_4 |        wait | U_0 U_1 U_2 U_3 U_4
        comm:        AR_0  AR_1  AR_2  AR_3  AR_4 ---^

    so only the last collective is exposed.  `mode`:
      * "full"     -- the whole picture above is ONE hipGraph (collectives captured, the fork/join become graph
                      branches); chosen when `parallel.selfcheck.captured_collective_ok` passed on every rank;
      * "segments" -- five captured A_i graphs, eager collectives on the comm stream, one captured update graph;
      * "eager"    -- no graphs (CPU / gloo plumbing, debugging).
    All three produce identical parameters (tests/test_dist_gloo.py compares against the serial order).
    """

    def __init__(self, members: List[RandomModel], dp, idx: torch.Tensor, mode: str = "eager"):
        if mode not in ("eager", "segments", "full"):
            raise ValueError(mode)
        self.members, self.dp, self.mode = members, dp, mode
        self.idx = idx.clone()
        self.on_gpu = idx.is_cuda
        self.comm = comm_stream() if self.on_gpu else None
        self.losses = None
        self.graphs = None
        if mode == "eager" or not self.on_gpu:
            self.mode = "eager"
            return
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm-up (really trains): autotuning, allocator, tables
            self._body()
        torch.cuda.current_stream().wait_stream(side)
        hipgraph.quiesce()
        F.hip_backend().sync_choices(dp)    # every rank captures rank 0's tuned plans
        F.hip_backend().save_choice_cache()
        pool = torch.cuda.graph_pool_handle()
        if mode == "full":
            g, self.losses = hipgraph.capture(self._body, pool)
            self.graphs = [g]
        else:
            self.graphs, self.losses = [], []
            for m in members:
                g, loss = hipgraph.capture(lambda m=m: m._fwd_bwd(self.idx), pool)
                self.graphs.append(g)
                self.losses.append(loss)
            g, _ = hipgraph.capture(self._updates, pool)
            self.graphs.append(g)

    def _fork_allreduce(self, m):
        if self.comm is None:
            m._reduce()
            return
        self.comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            m._reduce()

    def _join(self):
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)

    def _updates(self):
        for m in self.members:
            m._update()

    def _body(self):
        losses = []
        for m in self.members:
            losses.append(m._fwd_bwd(self.idx))
            self._fork_allreduce(m)
        self._join()
        self._updates()
        return losses

    def step(self, idx: torch.Tensor):
        self.idx.copy_(idx)
        if self.mode == "eager":
            self.losses = self._body()
        elif self.mode == "full":
            self.graphs[0].replay()
        else:
            for g, m in zip(self.graphs[:-1], self.members):
                g.replay()
                self._fork_allreduce(m)
            self._join()
            self.graphs[-1].replay()
        return self.losses[-1]


ENS_STREAMS = int(os.environ.get("DDL_ENS_STREAMS", "4"))     # streams for concurrent members (0 = one per member)


def member_streams(models: List[RandomModel], device, k: int = None):
    """HIP streams for members trained side by side on one GPU: (per-member stream list, distinct streams).

    One stream per member looks natural, but the runtime folds the streams onto its four hardware queues in
    creation order, so with ten members of very different cost one queue ends up with the three slowest and the
    generation waits for it.  Instead: `k` streams (default 4, one per hardware queue), members placed longest-first
    on the least-loaded stream by the measured time of one replay of their step graph (`RandomModel.prepare`;
    parameter counts where a member has no multi-step graph).  Measured on MI355X
    (`bench.py --config ensemble32`, 10 members): ddl_ray 472k -> 785k img/s, ddl_normal3 242k -> 328k, ddl 74k -> 87k."""
    k = ENS_STREAMS if k is None else k
    if k <= 0 or k >= len(models):
        streams = [torch.cuda.Stream(device=device) for _ in models]
        return streams, streams
    cost = [getattr(m, "chunk_ms", None) for m in models]       # measured by RandomModel.prepare
    if any(c is None for c in cost):
        cost = [float(m.net.num_params(True)) for m in models]
    pool = [torch.cuda.Stream(device=device) for _ in range(k)]
    load = [0.0] * k
    assign = [None] * len(models)
    for i in sorted(range(len(models)), key=lambda i: -cost[i]):
        q = min(range(k), key=lambda j: load[j])
        assign[i] = pool[q]
        load[q] += cost[i]
    return assign, pool


def train_concurrently(models: List[RandomModel], verbose: bool = False) -> None:
    """Train several ensemble members of ONE GPU at the same time (the reference trains its members one after
    another in a single process, /root/reference/DDL.py:142-144, or one per Ray actor).

    A batch-32 step of these small networks cannot fill 256 CUs, so each member gets its own HIP stream and the
    members' step graphs are issued round-robin from this one host thread; the hardware queues run them side by
    side.  Members are prepared (autotuned + captured) one at a time first, so plan timing is never disturbed by a
    neighbour.  A member that needs a value from the device (the epoch losses for its early-stop test) parks on an
    event; the others keep being issued meanwhile.  Each member's random stream, data split and arithmetic are its
    own: the result is identical to training the members serially."""
    import time
    if not models:
        return
    on_gpu = models[0].device.type == "cuda"
    if not on_gpu:
        for m in models:
            m.train()
        return
    # MLP members (the DDL_Ray* search spaces) go to the persistent whole-network kernel as ONE group: one launch
    # per epoch trains all of them, a workgroup each (models/mlp_fast.py); everything else runs its step graphs
    from . import mlp_fast
    group = []
    if os.environ.get("DDL_MLP_PERSISTENT", "1") == "1":
        group = [m for m in models if mlp_fast.eligible(m)]
    rest = [m for m in models if all(m is not g for g in group)]
    for m in rest:
        m.prepare()
    torch.cuda.synchronize()
    if os.environ.get("DDL_SERIAL_MEMBERS", "0") == "1":      # measurement knob: same graphs, one member at a time
        if group:
            mlp_fast.train_group(group, verbose)
        for m in rest:
            m.train()
        return
    gens = [m._train_gen() for m in rest] + ([mlp_fast.train_group_gen(group, verbose)] if group else [])
    names = [f"member {m.number}" for m in rest] + ([f"MLP group of {len(group)}"] if group else [])
    streams, _ = member_streams(rest, models[0].device)
    if group:                           # the MLP group's one persistent launch per epoch: a stream of its own
        streams = list(streams) + [torch.cuda.Stream(device=models[0].device)]
    waiting = [None] * len(gens)
    active = list(range(len(gens)))
    while active:
        progressed = False
        for i in list(active):
            ev = waiting[i]
            if ev is not None and not ev.query():
                continue
            progressed = True
            with torch.cuda.stream(streams[i]):
                try:
                    waiting[i] = next(gens[i])
                except StopIteration:
                    active.remove(i)
                    if verbose:
                        print(f"{names[i]} finished", flush=True)
        if not progressed:
            time.sleep(20e-6)
    torch.cuda.synchronize()
