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
from typing import Dict, List, Optional, Tuple

import torch

from ..config import SearchSpace
from ..ops import functional as F
from ..utils import hipgraph
from ..utils.data import Data, Sample
from .network import Network
from .optim import FlatOptimizer
from .search import CompiledSpec


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
        self.net.branch_wgrad = True     # measured +2 % on the 5-replica step (network.SideBranch)
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
    def _loss(self, logits, y, want_grad=True):
        if self.loss_name == "mean_squared_error":
            return F.softmax_mse(logits, y, want_grad)
        return F.softmax_ce(logits, y, want_grad)

    def _fwd_bwd(self, idx):
        net = self.net
        x = net.prepare_input(self.data.train_x, index=idx)
        y = self.data.train_y.index_select(0, idx)
        logits = net.forward(x, training=True)
        loss, dlogits, _ = self._loss(logits, y)
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

    # ------------------------------------------------------------------ fit / train
    def fit(self, epochs: int, batch_size: Optional[int] = None) -> List[float]:
        B = batch_size or self.space.batch_size
        tr = self.sample.train_idx
        n = tr.numel()
        losses = []
        for _ in range(epochs):
            perm = tr[torch.randperm(n, generator=self.gen).to(tr.device)]
            acc = torch.zeros((), dtype=torch.float32, device=self.device)
            for i in range(0, n, B):
                idx = perm[i:i + B]
                loss = self.train_step(idx)
                acc = acc + loss.detach() * idx.numel()      # Keras: sample-weighted epoch mean
            losses.append(acc / n)
        if self.dp is not None and self.dp.world > 1:
            # every rank must take the same early-stop decision or the ones that stay would wait forever in
            # their next collective: the decision is made on the rank-averaged epoch losses
            t = torch.stack(losses).float()
            self.dp.all_reduce_mean_(t)
            return [float(v) for v in t]
        return [float(v) for v in losses]

    def train(self):
        sp = self.space
        for i in range(int(sp.epochs / sp.mini_epochs)):
            hist = self.fit(sp.mini_epochs)
            self.history.append(hist)
            if self.verbose:
                print(i, hist, flush=True)
            # the reference's test, literally (DDL.py:95): with a NaN loss `1 - nan < rate` is False and it trains on
            if hist[0] == 0 or 1 - (hist[-1] / hist[0]) < sp.early_stop_rate:
                break
        if self.averager is not None:
            self.averager.finish()

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
        self.comm = torch.cuda.Stream() if self.on_gpu else None
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
