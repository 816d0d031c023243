"""Actor runtime: the capability the reference gets from Ray (/root/reference/DDL_Ray.py:3-7,62,
144-149; DDL_Ray2.py:147-158; DDL_Ray3.py:67,185-205; RayTest.py), rebuilt for one MI355X node.

    rt = runtime.init(num_workers=8)            # one worker PROCESS per GPU (cpu workers w/o GPUs)
    Remote = rt.remote(RandomModelActor, num_gpus=1)
    a = Remote.remote(args...)                   # actor creation -> handle
    f = a.train.remote()                         # async method call -> ObjectRef (future)
    rt.get([f, ...]); rt.wait([...]); rt.shutdown()

Semantics kept from the reference's use of Ray: an actor lives in one worker process and executes
its calls strictly in submission order (the reference relies on `train.remote()` finishing before
the queued `get_accuracy.remote()`, DDL_Ray2.py:151-154); calls on different workers run
concurrently; `get` blocks; futures passed as arguments are resolved by the runtime, not by the caller (the
submitting thread never waits: the call is parked in its worker's outbox until its inputs exist).  What is different:
workers are pinned one-per-GPU, big tensors never travel (each worker keeps the dataset resident
and receives seeds/indices), transport is a native framed-socket endpoint
(csrc/host/actor_transport.cpp), and a dying worker fails its futures with `WorkerDiedError`
instead of hanging the driver (the reference has no failure handling at all).
"""
from __future__ import annotations

import itertools
import os
import pickle
import subprocess
import sys
import tempfile
import threading
import time
from typing import Any, Dict, List, Optional, Sequence, Tuple

try:
    import cloudpickle as _cp
except ImportError:          # plain pickle handles importable classes/functions
    _cp = pickle

from . import transport as T

TAG_CREATE, TAG_CALL, TAG_TASK, TAG_RESULT, TAG_HELLO, TAG_SHUTDOWN, TAG_RELEASE = 1, 2, 3, 4, 5, 6, 7


class WorkerDiedError(RuntimeError):
    pass


class RemoteError(RuntimeError):
    """An exception raised inside an actor method / task; carries the remote traceback text."""


class GetTimeoutError(TimeoutError):
    pass


class ObjectRef:
    __slots__ = ("id", "_rt", "_ev", "_ok", "_val", "worker", "_waiters")

    def __init__(self, rt, rid, worker):
        self.id, self._rt, self.worker = rid, rt, worker
        self._ev = threading.Event()
        self._ok = None
        self._val = None
        self._waiters = set()       # workers whose outbox head is parked on this result

    def done(self) -> bool:
        return self._ev.is_set()

    def _set(self, ok, val):
        self._ok, self._val = ok, val
        self._ev.set()
        if self._waiters:           # calls parked on this result can go now (dispatcher thread, never this one)
            self._rt._kick(self._waiters)

    def get(self, timeout: Optional[float] = None):
        if not self._ev.wait(timeout):
            raise GetTimeoutError(f"object {self.id} not ready after {timeout}s")
        if self._ok:
            return self._val
        raise self._val

    def __repr__(self):
        return f"ObjectRef({self.id}, worker={self.worker}, done={self.done()})"


class _Method:
    def __init__(self, handle, name):
        self._h, self._name = handle, name

    def remote(self, *args, **kwargs) -> ObjectRef:
        return self._h._rt._submit(self._h._worker, TAG_CALL, (self._h._actor_id, self._name, args, kwargs))


class ActorHandle:
    def __init__(self, rt, actor_id, worker, created: ObjectRef):
        self._rt, self._actor_id, self._worker, self._created = rt, actor_id, worker, created

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return _Method(self, name)

    def __repr__(self):
        return f"ActorHandle(id={self._actor_id}, worker={self._worker})"


class RemoteClass:
    def __init__(self, rt, cls, num_gpus):
        self._rt, self._cls, self._num_gpus = rt, cls, num_gpus

    def remote(self, *args, **kwargs) -> ActorHandle:
        return self._rt._create_actor(self._cls, args, kwargs, self._num_gpus)


class RemoteFunction:
    def __init__(self, rt, fn, num_gpus):
        self._rt, self._fn, self._num_gpus = rt, fn, num_gpus

    def remote(self, *args, **kwargs) -> ObjectRef:
        w = self._rt._pick_worker(self._num_gpus)
        return self._rt._submit(w, TAG_TASK, (self._fn, args, kwargs))


class Runtime:
    def __init__(self, num_workers: Optional[int] = None, devices: Optional[Sequence[str]] = None,
                 connect_timeout: float = 120.0, env: Optional[Dict[str, str]] = None):
        if devices is None:
            ngpu = 0
            try:
                import torch
                ngpu = torch.cuda.device_count()
            except Exception:
                pass
            if num_workers is None:
                num_workers = ngpu if ngpu else min(4, os.cpu_count() or 1)
            devices = [f"cuda:{i % ngpu}" if ngpu else "cpu" for i in range(num_workers)]
        self.devices = list(devices)
        self._dir = tempfile.mkdtemp(prefix="ddl_rt_")
        self._path = os.path.join(self._dir, "driver.sock")
        self._ep = T.Endpoint()
        self._ep.listen(self._path)
        self._lock = threading.Lock()
        self._refs: Dict[int, ObjectRef] = {}
        self._ids = itertools.count(1)
        self._actor_ids = itertools.count(1)
        self._actors_on: Dict[int, int] = {}    # worker index -> actors placed there
        self._conn_of: Dict[int, int] = {}      # worker index -> connection id
        self._worker_of: Dict[int, int] = {}    # connection id -> worker index
        self._dead: set = set()
        self._procs: List[subprocess.Popen] = []
        self._hello = threading.Condition()
        self._closing = False
        # per-worker outbox: calls are SENT in submission order (an actor's mailbox is FIFO, DDL_Ray2.py:151-154), and
        # a call whose arguments contain unresolved futures parks at the head of its worker's outbox -- without
        # blocking the submitting thread (Ray resolves such arguments in its scheduler, not in the driver)
        self._outbox: Dict[int, List[Tuple[ObjectRef, int, Any]]] = {i: [] for i in range(len(devices))}
        self._out_mu = [threading.Lock() for _ in devices]
        self._kicks: List[int] = []
        self._kick_cv = threading.Condition()
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        for i, dev in enumerate(self.devices):
            e = dict(os.environ)
            e.update(env or {})
            e["DDL_WORKER_DEVICE"] = dev
            e["DDL_WORKER_INDEX"] = str(i)
            e["DDL_SYS_PATH"] = os.pathsep.join([root] + [p for p in sys.path if p])
            e["PYTHONPATH"] = os.pathsep.join([root] + ([e["PYTHONPATH"]] if e.get("PYTHONPATH") else []))
            e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(1, len(self.devices)))))
            self._procs.append(subprocess.Popen(
                [sys.executable, "-m", "distributeddeeplearning_amd.parallel.worker", self._path, str(i)],
                env=e))              # workers share the driver's cwd: relative artifact paths agree
        self._thread = threading.Thread(target=self._pump, name="ddl-runtime-events", daemon=True)
        self._thread.start()
        self._disp = threading.Thread(target=self._dispatch, name="ddl-runtime-dispatch", daemon=True)
        self._disp.start()
        deadline = time.time() + connect_timeout
        with self._hello:
            while len(self._conn_of) < len(self.devices):
                left = deadline - time.time()
                if left <= 0 or any(p.poll() is not None for p in self._procs):
                    self.shutdown()
                    raise WorkerDiedError("workers failed to start / connect")
                self._hello.wait(min(left, 0.5))

    # ------------------------------------------------------------------ event pump
    def _pump(self):
        while not self._closing:
            try:
                kind, conn, tag, payload = self._ep.next(0.25)
            except Exception:
                return
            if kind == T.MESSAGE:
                if tag == TAG_HELLO:
                    idx = pickle.loads(payload)
                    with self._hello:
                        self._conn_of[idx] = conn
                        self._worker_of[conn] = idx
                        self._hello.notify_all()
                elif tag == TAG_RESULT:
                    rid, ok, val = pickle.loads(payload)
                    with self._lock:
                        ref = self._refs.pop(rid, None)
                    if ref is not None:
                        ref._set(ok, val if ok else RemoteError(val))
            elif kind == T.CLOSED:
                idx = self._worker_of.get(conn)
                if idx is None:
                    continue
                self._dead.add(idx)
                with self._lock:
                    lost = [r for r in self._refs.values() if r.worker == idx]
                    for r in lost:
                        self._refs.pop(r.id, None)
                for r in lost:
                    r._set(False, WorkerDiedError(f"worker {idx} ({self.devices[idx]}) died"))
                self._kick([idx])           # parked calls of the dead worker fail too

    # ------------------------------------------------------------------ submission
    def _pick_worker(self, num_gpus=0) -> int:
        """Resource-aware placement (Ray places an actor where its `num_gpus` request fits,
        /root/reference/DDL_Ray3.py:67): the live worker with the fewest unfinished requests, then the fewest
        actors, wins.  With early stopping (DDL.py:95) members finish at very different times, so a blind
        round-robin would keep stacking new actors behind a busy worker while another GPU idles."""
        alive = [i for i in range(len(self.devices)) if i not in self._dead
                 and (not num_gpus or self.devices[i].startswith("cuda"))]
        if not alive:
            alive = [i for i in range(len(self.devices)) if i not in self._dead]
        if not alive:
            raise WorkerDiedError("no live workers")
        with self._lock:
            pending = {i: 0 for i in alive}
            for r in self._refs.values():
                if r.worker in pending:
                    pending[r.worker] += 1
        return min(alive, key=lambda i: (pending[i], self._actors_on.get(i, 0), i))

    def worker_load(self) -> Dict[int, Tuple[int, int]]:
        """{worker: (unfinished requests, actors placed)} -- what `_pick_worker` ranks by."""
        with self._lock:
            out = {i: [0, self._actors_on.get(i, 0)] for i in range(len(self.devices)) if i not in self._dead}
            for r in self._refs.values():
                if r.worker in out:
                    out[r.worker][0] += 1
        return {k: tuple(v) for k, v in out.items()}

    def _resolve(self, obj):
        """Replace every (finished) future in `obj` by its value; raises the dependency's error if it failed."""
        if isinstance(obj, ObjectRef):
            return obj.get()
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._resolve(o) for o in obj)
        if isinstance(obj, dict):
            return {k: self._resolve(v) for k, v in obj.items()}
        return obj

    def _pending_deps(self, obj, out):
        if isinstance(obj, ObjectRef):
            if not obj.done():
                out.append(obj)
        elif isinstance(obj, (list, tuple)):
            for o in obj:
                self._pending_deps(o, out)
        elif isinstance(obj, dict):
            for o in obj.values():
                self._pending_deps(o, out)
        return out

    def _kick(self, workers):
        with self._kick_cv:
            self._kicks.extend(workers)
            self._kick_cv.notify()

    def _dispatch(self):
        """Re-drains outboxes whose parked head call became sendable (woken by ObjectRef._set)."""
        while not self._closing:
            with self._kick_cv:
                while not self._kicks and not self._closing:
                    self._kick_cv.wait(0.25)
                todo, self._kicks = set(self._kicks), []
            for w in todo:
                self._drain(w)

    def _drain(self, worker: int):
        """Send this worker's queued calls, in order, up to the first one that still waits for an argument."""
        with self._out_mu[worker]:
            box = self._outbox[worker]
            while box:
                ref, tag, body = box[0]
                deps = self._pending_deps(body, [])
                if deps:
                    for d in deps:
                        d._waiters.add(worker)
                    if all(d.done() for d in deps):     # finished between the scan and the registration
                        continue
                    return
                box.pop(0)
                if ref.done():              # already failed (its worker died while it was parked)
                    continue
                if worker in self._dead:
                    with self._lock:
                        self._refs.pop(ref.id, None)
                    ref._set(False, WorkerDiedError(f"worker {worker} is dead"))
                    continue
                try:
                    payload = _cp.dumps((ref.id, self._resolve(body)))
                except Exception as e:      # noqa: BLE001 - a failed dependency fails the call that needed it
                    with self._lock:
                        self._refs.pop(ref.id, None)
                    ref._set(False, e)
                    continue
                try:
                    self._ep.send(self._conn_of[worker], tag, payload)
                except T.TransportError as e:
                    with self._lock:
                        self._refs.pop(ref.id, None)
                    ref._set(False, WorkerDiedError(f"worker {worker}: {e}"))

    def _submit(self, worker: int, tag: int, body) -> ObjectRef:
        rid = next(self._ids)
        ref = ObjectRef(self, rid, worker)
        if worker in self._dead:
            ref._set(False, WorkerDiedError(f"worker {worker} is dead"))
            return ref
        with self._lock:
            self._refs[rid] = ref
        with self._out_mu[worker]:
            self._outbox[worker].append((ref, tag, body))
        self._drain(worker)
        return ref

    def _create_actor(self, cls, args, kwargs, num_gpus) -> ActorHandle:
        w = self._pick_worker(num_gpus)
        self._actors_on[w] = self._actors_on.get(w, 0) + 1
        aid = next(self._actor_ids)
        created = self._submit(w, TAG_CREATE, (aid, cls, args, kwargs))
        return ActorHandle(self, aid, w, created)

    # ------------------------------------------------------------------ public API
    def remote(self, target=None, num_gpus: int = 0):
        """`rt.remote(Cls)` / `rt.remote(fn)` / decorator form `@rt.remote(num_gpus=1)`."""
        if target is None:
            return lambda t: self.remote(t, num_gpus)
        if isinstance(target, type):
            return RemoteClass(self, target, num_gpus)
        return RemoteFunction(self, target, num_gpus)

    def get(self, refs, timeout: Optional[float] = None):
        if isinstance(refs, ObjectRef):
            return refs.get(timeout)
        if isinstance(refs, ActorHandle):
            return refs._created.get(timeout)
        deadline = None if timeout is None else time.time() + timeout
        out = []
        for r in refs:
            out.append(self.get(r, None if deadline is None else max(0.0, deadline - time.time())))
        return out

    def wait(self, refs: Sequence[ObjectRef], num_returns: int = 1, timeout: Optional[float] = None):
        deadline = None if timeout is None else time.time() + timeout
        while True:
            ready = [r for r in refs if r.done()]
            if len(ready) >= num_returns or (deadline is not None and time.time() >= deadline):
                return ready, [r for r in refs if not r.done()]
            time.sleep(0.002)

    def release(self, handle: ActorHandle):
        self._actors_on[handle._worker] = max(0, self._actors_on.get(handle._worker, 0) - 1)
        self._submit(handle._worker, TAG_RELEASE, (handle._actor_id,))

    def alive_workers(self) -> List[int]:
        return [i for i in range(len(self.devices)) if i not in self._dead]

    def shutdown(self):
        if self._closing:
            return
        for idx, conn in list(self._conn_of.items()):
            if idx not in self._dead:
                try:
                    self._ep.send(conn, TAG_SHUTDOWN, b"")
                except Exception:
                    pass
        for p in self._procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()          # exact child PID, never a pattern
        self._closing = True
        with self._kick_cv:
            self._kick_cv.notify_all()
        self._thread.join(timeout=2)
        self._disp.join(timeout=2)
        self._ep.close()
        try:
            os.rmdir(self._dir)
        except OSError:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.shutdown()


_GLOBAL: Optional[Runtime] = None


def init(num_workers: Optional[int] = None, devices: Optional[Sequence[str]] = None, **kw) -> Runtime:
    """Module-level runtime (the `ray.init(...)` moment of the reference scripts)."""
    global _GLOBAL
    if _GLOBAL is None:
        _GLOBAL = Runtime(num_workers, devices, **kw)
    return _GLOBAL


def shutdown():
    global _GLOBAL
    if _GLOBAL is not None:
        _GLOBAL.shutdown()
        _GLOBAL = None


def worker_device() -> str:
    """Inside a worker: the device this process is pinned to ('cuda:3' / 'cpu')."""
    return os.environ.get("DDL_WORKER_DEVICE", "cpu")


def worker_index() -> int:
    return int(os.environ.get("DDL_WORKER_INDEX", "0"))
