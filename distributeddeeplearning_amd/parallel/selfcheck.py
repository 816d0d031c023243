"""Start-up self-tests of the data-parallel path, agreed across ranks.

The development box has one GPU, the scaling run has eight: whatever cannot be validated here must
validate itself there, cheaply, before the timed region -- and every rank must reach the SAME verdict,
otherwise ranks would replay different graphs and dead-lock in their collectives.

`captured_collective_ok(dp)`: is an all-reduce captured into a hipGraph replayed correctly?  A captured
RCCL collective is what lets a whole DP training step be ONE graph (`--graph full`) instead of graph
segments around eagerly issued collectives (4 segment boundaries of ~14 us each on the DCGAN step,
docs/DESIGN.md).  The test captures an all-reduce of a rank-dependent buffer, replays it twice on
different contents and compares with the eager result, then takes the MAX of the failure flags.

`ranks_seen(dp)`: all-gather of the ranks, for the benchmark record (proof that the communicator really
spanned N processes).
"""
from __future__ import annotations

import time

import torch
import torch.distributed as dist

from ..utils import hipgraph


def _agree(failed: bool, device) -> bool:
    """True iff NO rank failed (all-reduce MAX of the failure flag over the default group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return not failed
    f = torch.tensor([1.0 if failed else 0.0], device=device)
    dist.all_reduce(f, op=dist.ReduceOp.MAX)
    return float(f.item()) == 0.0


def _wait(stream, timeout_s: float) -> bool:
    """Wait for `stream` without blocking forever: a wedged collective must fail the test, not hang the job."""
    ev = torch.cuda.Event()
    ev.record(stream)
    t0 = time.time()
    while not ev.query():
        if time.time() - t0 > timeout_s:
            return False
        time.sleep(0.001)
    return True


def captured_collective_ok(dp, device, numel: int = 1 << 18, timeout_s: float = 30.0, verbose: bool = False) -> bool:
    """See module docstring.  Collective-safe: every rank executes the same sequence of collectives whatever
    happens locally (a rank whose capture failed still takes part in the agreement all-reduces, and nobody
    replays unless everybody captured)."""
    device = torch.device(device)
    if device.type != "cuda":
        return False
    rank = getattr(dp, "rank", 0)
    base = (torch.arange(numel, device=device, dtype=torch.float32) % 251.0) * float(rank + 1) + float(rank)
    eager = base.clone()
    dp.all_reduce_sum_(eager)               # also warms the communicator up before any capture
    static = base.clone()
    graph, err = None, None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dp.all_reduce_sum_(static)      # same call once on a side stream, as capture warm-ups do
        torch.cuda.current_stream().wait_stream(side)
        hipgraph.quiesce()
        graph, _ = hipgraph.capture(lambda: dp.all_reduce_sum_(static))
    except Exception as e:      # noqa: BLE001 - the verdict is "no", whatever the reason
        err = e
    if not _agree(graph is None, device):
        if verbose and err is not None:
            print(f"[selfcheck rank {rank}] capture of a collective failed: {err!r}", flush=True)
        torch.cuda.synchronize()
        return False
    ok = True
    try:
        for mult in (1.0, 3.0):             # two replays on different contents: the graph really re-executes
            static.copy_(base * mult)
            graph.replay()
            if not _wait(torch.cuda.current_stream(), timeout_s):
                ok = False
                break
            ok = ok and torch.equal(static, eager * mult)
    except Exception as e:      # noqa: BLE001
        err, ok = e, False
    agreed = _agree(not ok, device)
    if verbose and not agreed:
        print(f"[selfcheck rank {rank}] captured all-reduce replay local_ok={ok} err={err!r}", flush=True)
    return agreed


def ranks_seen(dp, device) -> list:
    """Sorted list of ranks that answered an all-gather through `dp` (the benchmark record's proof of N)."""
    if dp is None:
        return [0]
    device = torch.device(device)
    v = torch.tensor([float(dp.rank)], device=device)
    got = dp.all_gather_scalars(v).reshape(-1).cpu().tolist()
    return sorted(int(x) for x in got)
