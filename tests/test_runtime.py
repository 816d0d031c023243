"""Actor runtime (the Ray replacement): ordering, parallelism, errors, worker death."""
import os
import time

import pytest

from distributeddeeplearning_amd.parallel import runtime
from distributeddeeplearning_amd.parallel import transport as T


class Counter:
    def __init__(self, start):
        self.v, self.log = start, []

    def add(self, k, delay=0.0):
        time.sleep(delay)
        self.v += k
        self.log.append(k)
        return self.v

    def state(self):
        return self.v, self.log, os.getpid(), runtime.worker_device(), runtime.worker_index()

    def boom(self):
        raise ValueError("nope")

    def die(self):
        os._exit(3)


def _sleep(s):
    time.sleep(s)
    return os.getpid()


@pytest.fixture(scope="module")
def rt():
    r = runtime.Runtime(3, ["cpu"] * 3)
    yield r
    r.shutdown()


def test_transport_frames_roundtrip(tmp_path):
    a, b = T.Endpoint(), T.Endpoint()
    a.listen(str(tmp_path / "s.sock"))
    c = b.connect(str(tmp_path / "s.sock"), 5.0)
    kind, conn_a, _, _ = a.next(5.0)
    assert kind == T.CONNECTED
    big = os.urandom(3_000_000)
    b.send(c, 42, big)
    b.send(c, 43, b"")
    kind, _, tag, payload = a.next(5.0)
    assert (kind, tag) == (T.MESSAGE, 42) and payload == big
    kind, _, tag, payload = a.next(5.0)
    assert (kind, tag, payload) == (T.MESSAGE, 43, b"")
    a.send(conn_a, 7, b"pong")
    assert b.next(5.0)[3] == b"pong"
    b.close()
    assert a.next(5.0)[0] == T.CLOSED           # peer death is an event, not a hang
    a.close()


def test_tasks_run_in_parallel_like_raytest(rt):
    f = rt.remote(_sleep)
    rt.get([f.remote(0.0) for _ in range(3)])
    t0 = time.time()
    pids = rt.get([f.remote(0.5) for _ in range(3)])
    assert time.time() - t0 < 1.2 and len(set(pids)) == 3           # 0.5 s if parallel, 1.5 s if serial


def test_actor_calls_keep_submission_order_and_state(rt):
    C = rt.remote(Counter, num_gpus=1)
    a, b = C.remote(10), C.remote(100)
    refs = [a.add.remote(i, 0.01) for i in range(6)]                 # fire-and-forget, DDL_Ray2-style
    last = a.state.remote()                                          # queued behind the adds
    v, log, pid_a, dev, idx = rt.get(last)
    assert (v, log) == (25, list(range(6))) and rt.get(refs) == [10, 11, 13, 16, 20, 25]
    assert dev == "cpu" and rt.get(b.state.remote())[2] != pid_a     # actors placed round-robin
    assert rt.get(a.add.remote(b.add.remote(1))) == 25 + 101         # a future as an argument
    ready, pending = rt.wait([a.add.remote(0, 0.3)], num_returns=1, timeout=0.01)
    assert pending and not ready


def test_future_arguments_do_not_block_the_submitting_thread(rt):
    """A future passed as an argument is resolved by the runtime (Ray does it in its scheduler): the driver keeps
    submitting while the producer still runs -- async mode builds transfer children from parents that are still
    training (/root/reference/DDL_Ray3.py:195-205).  The consumer's mailbox stays FIFO behind the parked call, and a
    failed producer fails exactly the calls that needed it."""
    C = rt.remote(Counter, num_gpus=1)
    a, b = C.remote(0), C.remote(1000)
    rt.get([a.state.remote(), b.state.remote()])
    t0 = time.time()
    slow = a.add.remote(5, 0.6)                       # producer: 0.6 s on a's worker
    parked = b.add.remote(slow)                       # needs the producer's result
    behind = b.add.remote(1)                          # queued behind the parked call in b's mailbox
    other = a.add.remote(1)                           # a's own mailbox is not held up by b's parked call
    assert time.time() - t0 < 0.3                     # nothing above waited for the producer
    assert not parked.done() and not behind.done()
    assert rt.get(parked) == 1005 and rt.get(behind) == 1006 and rt.get(other) == 6
    assert rt.get(b.state.remote())[1] == [5, 1]      # order kept: the parked call ran first
    bad = a.boom.remote()
    needs_bad = b.add.remote(bad)
    after = b.add.remote(2)
    with pytest.raises(runtime.RemoteError):
        rt.get(needs_bad)
    assert rt.get(after) == 1008                      # the mailbox moved on


def test_remote_exception_and_timeout(rt):
    a = rt.remote(Counter).remote(0)
    with pytest.raises(runtime.RemoteError, match="ValueError: nope"):
        rt.get(a.boom.remote())
    assert rt.get(a.add.remote(1)) == 1                              # actor survives its exception
    with pytest.raises(runtime.GetTimeoutError):
        rt.get(a.add.remote(1, 1.0), timeout=0.05)


def test_worker_death_fails_futures_and_pool_survives():
    with runtime.Runtime(2, ["cpu"] * 2) as r:
        C = r.remote(Counter)
        a, b = C.remote(0), C.remote(0)
        doomed = [b.die.remote(), b.add.remote(1)]
        for f in doomed:
            with pytest.raises(runtime.WorkerDiedError):
                r.get(f, timeout=20)
        assert r.alive_workers() == [a._worker]
        assert r.get(a.add.remote(5)) == 5
        c = C.remote(1)                                              # new actors avoid the dead worker
        assert c._worker == a._worker and r.get(c.add.remote(1)) == 2


def test_placement_prefers_the_least_loaded_worker():
    """Actors go where the fewest requests are pending (Ray places by free resources, DDL_Ray3.py:67), not round-robin."""
    r = runtime.Runtime(3, ["cpu"] * 3)
    try:
        C = r.remote(Counter)
        first = [C.remote(0) for _ in range(3)]
        assert sorted(a._worker for a in first) == [0, 1, 2]            # empty pool: spread by actor count
        busy = [a for a in first if a._worker in (0, 1)]
        slow = [a.add.remote(1, 1.5) for a in busy]                     # workers 0 and 1 are busy for a while
        time.sleep(0.2)
        idle = next(a._worker for a in first if a._worker == 2)
        load = r.worker_load()
        assert load[0][0] >= 1 and load[1][0] >= 1 and load[2][0] == 0
        late = [C.remote(0) for _ in range(2)]
        assert late[0]._worker == idle                                  # round-robin would have picked worker 0
        r.get([a._created for a in late] + slow)
    finally:
        r.shutdown()
