"""`EnsembleModel`: generations of random members, ranking, transfer generation, top-k vote.

Reference behaviour (most complete variant /root/reference/DDL_Ray3.py:180-244; single-process
twins DDL.py:133-165, DDL_Normal3.py:176-238):
  * `make_and_train_new_models(n)`      n members with independent random splits, trained, scored
                                         on their own validation split;
  * `make_and_train_transfer_models(n)` children of the current top-n (weights cloned, last 1-2
                                         layers cut, new random head), trained, scored, pooled;
  * `sort_models()`                     descending validation accuracy;
  * `get_accuracy()`                    unweighted sum of the top-k members' softmax outputs on the
                                         common test set, argmax vs labels;
  * `plot_models()` / `save_models()`   per-member graph / models/modelN export.

Three execution back-ends share that logic:
  * mode="local"    everything in this process, members trained one after another (DDL.py);
  * mode="barrier"  one actor per member on the worker pool, wait for all trainings, then gather
                    accuracies (DDL_Ray.py:144-149);
  * mode="async"    fire-and-forget train + queued accuracy futures, resolved only when ranking or
                    voting needs them (DDL_Ray2.py:151-158, DDL_Ray3.py:185-237);
plus `SpmdEnsemble` for `torchrun`: the members of a generation are placed on the ranks longest-first by
estimated cost (every rank computes the same assignment, `SpmdEnsemble._place`), and the driver-side
gathers become RCCL collectives -- all_reduce of the accuracy slots (M3/M5; each slot written by its
owner), all_reduce(sum) of the vote matrix (M4), broadcast of a parent's flat weights (M6).  Images never travel: each
process keeps the dataset resident and receives split seeds.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from ..config import SearchSpace
from ..models import search
from ..models.classifier import RandomModel, train_concurrently
from ..models.search import CompiledSpec
from ..ops import functional as F
from ..utils.data import Data
from . import runtime as R

# ------------------------------------------------------------------------------------ members
_DATA_CACHE: Dict[Tuple, Data] = {}


def shared_data(device, channels: int, data_kw: Optional[dict] = None) -> Data:
    """One resident dataset per (process, device, channels)."""
    kw = dict(data_kw or {})
    key = (str(device), channels, tuple(sorted(kw.items())))
    if key not in _DATA_CACHE:
        _DATA_CACHE[key] = Data(device, channels=channels, **kw)
    return _DATA_CACHE[key]


class MemberActor:
    """A `RandomModel` hosted in a worker process (the `@ray.remote(num_gpus=1)` class of
    DDL_Ray3.py:67).  Only small things cross the process boundary: the compiled spec, seeds,
    accuracies, [n_test,10] softmax matrices and, for transfer learning, one model's weights."""

    def __init__(self, space: SearchSpace, compiled: CompiledSpec, sample_seed: int, seed: int, number: int,
                 data_kw: Optional[dict] = None, parent=None, device: Optional[str] = None, verbose: bool = True):
        device = device or R.worker_device()
        self.data = shared_data(device, space.channels, data_kw)
        sample = self.data.get_random_sample(space.data_train_size, seed=sample_seed)
        parent_state, inherited = parent if parent is not None else (None, None)
        parent_state = _resolve(parent_state)       # local mode hands over an already-resolved future
        self.model = RandomModel(self.data, sample, space, compiled, device, number, seed,
                                 parent_state, inherited, verbose=verbose)

    def train(self):
        self.model.train()
        return self.model.history

    def get_accuracy(self, on_test: bool = False) -> float:
        if on_test:
            return self.model.get_accuracy((self.data.test_x, self.data.test_y))
        return self.model.get_accuracy()

    def predict_test(self) -> np.ndarray:
        return self.model.predict(self.data.test_x).float().cpu().numpy()

    def get_model(self):
        compiled, state = self.model.get_model()
        return compiled, state

    def get_state(self):
        """Weights only (the driver already holds the member's spec): what a transfer child inherits (M6)."""
        return self.model.get_model()[1]

    def save(self, directory: Optional[str] = None) -> str:
        return str(self.model.save(directory))

    def plot(self, path: str) -> str:
        from ..utils import plot
        return str(plot.plot_model(self.model.net.spec, path, show_shapes=True))


class _Now:
    """An already-resolved future (local mode speaks the same protocol as ObjectRef)."""

    def __init__(self, v):
        self._v = v

    def get(self, timeout=None):
        return self._v

    def done(self):
        return True


class _LocalHandle:
    def __init__(self, actor: MemberActor):
        self._a = actor

    def call(self, name, *args, **kw):
        return _Now(getattr(self._a, name)(*args, **kw))


class _RemoteHandle:
    def __init__(self, handle):
        self._h = handle

    def call(self, name, *args, **kw):
        return getattr(self._h, name).remote(*args, **kw)


def _resolve(v):
    return v.get() if hasattr(v, "get") and hasattr(v, "done") else v


def _sort_key(acc: float):
    return -acc if not math.isnan(acc) else math.inf


# ------------------------------------------------------------------------------------ orchestrator
class EnsembleModel:
    def __init__(self, space: SearchSpace, mode: str = "local", runtime: Optional[R.Runtime] = None,
                 device: str = "cpu", data_kw: Optional[dict] = None, verbose: bool = True,
                 drop_failed: bool = True):
        if mode not in ("local", "barrier", "async"):
            raise ValueError(mode)
        if mode != "local" and runtime is None:
            raise ValueError("actor modes need a Runtime")
        self.space, self.mode, self.rt, self.device = space, mode, runtime, device
        self.data_kw, self.verbose, self.drop_failed = data_kw, verbose, drop_failed
        self.rng = np.random.default_rng(space.seed)
        from ..models.ir import NameScope
        self.names = NameScope()          # Keras-style global layer counters
        self.accuracy_with_models: List[list] = []      # [accuracy | future, handle, compiled]
        self._number = 0
        self._remote_cls = runtime.remote(MemberActor, num_gpus=1) if runtime is not None else None
        # the driver needs labels for the vote; images stay in the workers
        self.data = shared_data(device if mode == "local" else "cpu", space.channels, data_kw)

    # ---- member creation
    def _spawn(self, compiled: CompiledSpec, parent=None):
        n = self._number
        self._number += 1
        sample_seed = int(self.rng.integers(0, 2 ** 31 - 1))
        seed = int(self.rng.integers(0, 2 ** 31 - 1))
        args = (self.space, compiled, sample_seed, seed, n, self.data_kw, parent)
        if self.mode == "local":
            return _LocalHandle(MemberActor(*args, device=self.device, verbose=self.verbose))
        return _RemoteHandle(self._remote_cls.remote(*args, verbose=self.verbose))

    def _train_and_score(self, members):
        if self.mode == "barrier":                      # DDL_Ray.py:148-149: wait, then gather
            for f in [h.call("train") for h, _ in members]:
                try:
                    f.get()
                except (R.WorkerDiedError, R.RemoteError):
                    if not self.drop_failed:
                        raise
            accs = [h.call("get_accuracy") for h, _ in members]
            for (h, c), a in zip(members, accs):
                self.accuracy_with_models.append([a, h, c])
            self._settle()
        elif self.mode == "local":
            # one process, one device: the members of a generation train side by side on their own HIP streams
            # (serially on the CPU); same results as one after another (models/classifier.py train_concurrently)
            train_concurrently([h._a.model for h, _ in members])
            for h, c in members:
                self.accuracy_with_models.append([h.call("get_accuracy"), h, c])
            self._settle()
        else:                                           # async: fire-and-forget train, queued accuracy futures
            for h, c in members:
                h.call("train")
                self.accuracy_with_models.append([h.call("get_accuracy"), h, c])

    def make_and_train_new_models(self, n: Optional[int] = None):
        n = self.space.random_model_size if n is None else n
        members = []
        for _ in range(n):
            compiled = search.make_new_model(self.space, self.rng, self.data.input_shape, 10, self.names)
            members.append((self._spawn(compiled), compiled))
        self._train_and_score(members)

    train_new_model = make_and_train_new_models          # DDL.py:138 name

    def make_and_train_transfer_models(self, n: Optional[int] = None):
        n = self.space.transfer_model_size if n is None else n
        self.sort_models()
        members = []
        for acc, h, c in self.accuracy_with_models[:n]:
            # M6: the parent's weights travel worker -> driver -> child's worker as a FUTURE: the runtime parks the
            # child's creation until they exist (parallel/runtime.py outbox), the driver goes on building the
            # generation from the spec it already holds (DDL_Ray3.py:198 blocks in ray.get here)
            state = h.call("get_state")
            child, inherited = search.make_transfer_model(self.space, self.rng, c.spec, 10)
            members.append((self._spawn(child, parent=(state, inherited)), child))
        self._train_and_score(members)

    # ---- ranking / voting
    def _settle(self):
        """Resolve accuracy futures; a member whose worker died is dropped (or re-raised)."""
        kept = []
        for entry in self.accuracy_with_models:
            try:
                entry[0] = float(_resolve(entry[0]))
                kept.append(entry)
            except (R.WorkerDiedError, R.RemoteError) as e:
                if not self.drop_failed:
                    raise
                if self.verbose:
                    print(f"dropping failed ensemble member: {e}", flush=True)
        self.accuracy_with_models = kept

    def sort_models(self):
        self._settle()
        self.accuracy_with_models.sort(key=lambda e: _sort_key(e[0]))     # stable; never compares models

    def _gather(self, entries, method, *args):
        """Call `method` on every entry's member; -> [(entry, value)].  Members whose worker died are
        removed from the pool (drop_failed) -- ensemble members are independent, so losing one only
        shrinks the vote -- or the error is re-raised."""
        futs = [(e, e[1].call(method, *args)) for e in entries]
        out = []
        for e, f in futs:
            try:
                out.append((e, _resolve(f)))
            except (R.WorkerDiedError, R.RemoteError) as err:
                if not self.drop_failed:
                    raise
                if self.verbose:
                    print(f"dropping failed ensemble member: {err}", flush=True)
                if e in self.accuracy_with_models:
                    self.accuracy_with_models.remove(e)
        return out

    def member_test_accuracies(self) -> List[float]:
        self._settle()
        return [float(v) for _, v in self._gather(list(self.accuracy_with_models), "get_accuracy", True)]

    def get_accuracy(self, k: Optional[int] = None) -> float:
        k = self.space.predict_model_size if k is None else k
        self.sort_models()
        got = self._gather(self.accuracy_with_models[:k], "predict_test")
        if len(got) < k:                                 # refill the vote from the next-best live members
            have = {id(e) for e, _ in got}
            rest = [e for e in self.accuracy_with_models if id(e) not in have][:k - len(got)]
            got += self._gather(rest, "predict_test")
        if not got:
            return float("nan")
        votes = None
        for _, p in got:
            p = torch.from_numpy(np.asarray(p, dtype=np.float32))
            votes = p if votes is None else votes + p
        target = self.data.test_y.cpu()
        return int(F.argmax_match_count(votes, target)) / votes.shape[0]

    # ---- artifacts
    def plot_models(self, directory: Optional[str] = None) -> List[str]:
        d = directory or self.space.model_save_directory
        self._settle()
        futs = [h.call("plot", f"{d}/model{i} accuracy {acc:f}.png")
                for i, (acc, h, _) in enumerate(self.accuracy_with_models)]
        return [_resolve(f) for f in futs]

    def save_models(self, directory: Optional[str] = None) -> List[str]:
        self._settle()
        return [_resolve(f) for f in [h.call("save", directory) for _, h, _ in self.accuracy_with_models]]


# ------------------------------------------------------------------------------------ SPMD (torchrun)
class _CommOps:
    """The three collectives an SPMD ensemble needs, expressed on the communicator facade."""

    def __init__(self, comm):
        self.comm = comm

    def all_reduce(self, t):
        self.comm.all_reduce_sum_(t)

    def broadcast(self, t, src=0):
        self.comm.broadcast_(t, src=src)


class SpmdEnsemble:
    """The same generations under `torchrun`: one process per GPU, every rank draws the same architectures from
    the same seeded RNG, the members of a generation are spread over the ranks by estimated cost (longest
    processing time first onto the least-loaded rank -- every rank computes the same assignment, nothing is
    communicated), and the only communication is at generation boundaries."""

    def __init__(self, space: SearchSpace, device, comm=None, data_kw: Optional[dict] = None,
                 verbose: bool = False):
        """`comm`: a communicator facade (parallel.ddp.FlatDataParallel, parallel.rccl.RcclCommunicator,
        parallel.p2p.P2PAllReduce); default: FlatDataParallel over the initialised torch.distributed group."""
        import torch.distributed as dist
        self.space, self.device, self.verbose = space, torch.device(device), verbose
        if comm is None and dist.is_available() and dist.is_initialized():
            from .ddp import FlatDataParallel
            comm = FlatDataParallel()
        self.comm = comm
        self.dist = _CommOps(comm) if comm is not None else None
        self.rank = comm.rank if comm is not None else 0
        self.world = comm.world if comm is not None else 1
        self.rng = np.random.default_rng(space.seed)
        self.data = shared_data(device, space.channels, data_kw)
        self.data_kw = data_kw
        self.members: List[dict] = []     # {number, owner, compiled, acc, actor (owner only)}

    def _add(self, compiled, parent=None) -> dict:
        """Register a member (seeds drawn now, in member order); its owner rank and actor follow in `_place`."""
        n = len(self.members)
        sample_seed = int(self.rng.integers(0, 2 ** 31 - 1))
        seed = int(self.rng.integers(0, 2 ** 31 - 1))
        m = dict(number=n, owner=None, compiled=compiled, acc=float("nan"), actor=None,
                 _make=(sample_seed, seed, parent))
        self.members.append(m)
        return m

    @staticmethod
    def member_cost(compiled) -> float:
        """Relative training cost of a member: multiply-accumulates of one forward pass (convolutions count their
        output pixels, Dense layers their weights); parameters alone would rate a small conv stack as free."""
        spec = compiled.spec
        shapes = spec.infer_shapes()
        cost = 1.0
        for lname, wname, ks, _ in spec.param_specs():
            n = float(np.prod(ks))
            if "kernel" in wname and spec.layer(lname).kind != "dense":
                out = shapes[lname]
                n *= float(out[0] * out[1])          # every output pixel applies the whole kernel
            cost += n
        return cost

    def _place(self, new: List[dict]):
        """Longest-processing-time-first assignment of a generation to the ranks; identical on every rank."""
        load = [0.0] * self.world
        order = sorted(range(len(new)), key=lambda i: (-self.member_cost(new[i]["compiled"]), i))
        for i in order:
            r = min(range(self.world), key=lambda k: (load[k], k))
            load[r] += self.member_cost(new[i]["compiled"])
            new[i]["owner"] = r
        for m in new:
            sample_seed, seed, parent = m.pop("_make")
            if m["owner"] == self.rank:
                m["actor"] = MemberActor(self.space, m["compiled"], sample_seed, seed, m["number"], self.data_kw,
                                         parent, device=str(self.device), verbose=self.verbose)

    def _train_and_gather(self, new: List[dict]):
        self._place(new)
        accs = torch.zeros(len(new), dtype=torch.float32, device=self.device)
        train_concurrently([m["actor"].model for m in new if m["actor"] is not None])     # this rank's members
        for i, m in enumerate(new):
            if m["actor"] is not None:
                a = m["actor"].get_accuracy()
                accs[i] = a if not math.isnan(a) else -1.0
        if self.dist:
            self.dist.all_reduce(accs)                      # each slot is written by exactly one rank
        for m, a in zip(new, accs.tolist()):
            m["acc"] = a

    def make_and_train_new_models(self, n: Optional[int] = None):
        n = self.space.random_model_size if n is None else n
        new = [self._add(search.make_new_model(self.space, self.rng, self.data.input_shape, 10))
               for _ in range(n)]
        self._train_and_gather(new)

    def _broadcast_state(self, m: dict):
        """Parent weights -> every rank as one flat fp32 buffer (M6), unpacked by spec order."""
        specs = m["compiled"].spec.param_specs()
        sizes = [int(np.prod(s)) for _, _, s, _ in specs]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=self.device)
        state = None
        if m["actor"] is not None:
            state = m["actor"].model.net.logical_state()
            flat.copy_(torch.cat([state[(l, w)].reshape(-1) for l, w, _, _ in specs]).to(self.device))
        if self.dist:
            self.dist.broadcast(flat, src=m["owner"])
        if state is None:
            net_shapes = self._logical_shapes(m["compiled"].spec)
            state, off = {}, 0
            for (l, w, _, _), k in zip(specs, sizes):
                state[(l, w)] = flat[off:off + k].cpu().reshape(net_shapes[(l, w)])
                off += k
        return state

    @staticmethod
    def _logical_shapes(spec):
        """Canonical (kernel-layout) logical shapes per weight, matching Network.logical_state()."""
        out = {}
        for l, w, ks, _ in spec.param_specs():
            kind = spec.layer(l).kind
            if w == "kernel" and kind == "dense":
                out[(l, w)] = (ks[1], ks[0])
            elif w == "kernel" and kind == "conv2d":
                out[(l, w)] = (ks[3], ks[0], ks[1], ks[2])
            elif w == "kernel":                       # conv2d_transpose (kh,kw,filters,in) -> [F][KH][KW][Cin]
                out[(l, w)] = (ks[2], ks[0], ks[1], ks[3])
            elif w == "depthwise_kernel":
                out[(l, w)] = (ks[0], ks[1], ks[2])
            else:
                out[(l, w)] = tuple(ks)
        return out

    def make_and_train_transfer_models(self, n: Optional[int] = None):
        n = self.space.transfer_model_size if n is None else n
        self.sort_models()
        new = []
        for parent in list(self.members[:n]):
            state = self._broadcast_state(parent)
            child, inherited = search.make_transfer_model(self.space, self.rng, parent["compiled"].spec, 10)
            new.append(self._add(child, parent=(state, inherited)))
        self._train_and_gather(new)

    def sort_models(self):
        self.members.sort(key=lambda m: _sort_key(m["acc"]))

    def member_test_accuracies(self) -> List[float]:
        accs = torch.zeros(len(self.members), dtype=torch.float32, device=self.device)
        for i, m in enumerate(self.members):
            if m["actor"] is not None:
                accs[i] = m["actor"].get_accuracy(True)
        if self.dist:
            self.dist.all_reduce(accs)
        return accs.tolist()

    def get_accuracy(self, k: Optional[int] = None) -> float:
        k = self.space.predict_model_size if k is None else k
        self.sort_models()
        votes = torch.zeros(self.data.test_x.shape[0], 10, dtype=torch.float32, device=self.device)
        for m in self.members[:k]:
            if m["actor"] is not None:
                votes += m["actor"].model.predict(self.data.test_x).float()
        if self.dist:
            self.dist.all_reduce(votes)                     # M4: non-selected ranks contribute zeros
        return int(F.argmax_match_count(votes, self.data.test_y)) / votes.shape[0]

    def save_models(self, directory: Optional[str] = None) -> List[str]:
        return [m["actor"].save(directory) for m in self.members if m["actor"] is not None]
