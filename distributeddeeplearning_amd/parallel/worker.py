"""Worker process of the actor runtime: `python -m distributeddeeplearning_amd.parallel.worker
<driver socket> <index>`.  Hosts any number of actors, executes messages strictly FIFO (so calls
on one actor keep their submission order, which the reference relies on: DDL_Ray2.py:151-154) and
pushes every result back to the driver as soon as it exists."""
from __future__ import annotations

import os
import pickle
import sys
import traceback


def main():
    path, index = sys.argv[1], int(sys.argv[2])
    for p in reversed(os.environ.get("DDL_SYS_PATH", "").split(os.pathsep)):
        if p and p not in sys.path:
            sys.path.insert(0, p)
    try:
        import cloudpickle as cp
    except ImportError:
        cp = pickle
    from distributeddeeplearning_amd.parallel import runtime as R
    from distributeddeeplearning_amd.parallel import transport as T

    ep = T.Endpoint()
    conn = ep.connect(path, 60.0)
    ep.send(conn, R.TAG_HELLO, pickle.dumps(index))
    actors = {}

    def reply(rid, ok, val):
        try:
            data = cp.dumps((rid, ok, val))
        except Exception as e:                       # unpicklable result
            data = pickle.dumps((rid, False, f"result of call {rid} is not picklable: {e!r}"))
        ep.send(conn, R.TAG_RESULT, data)

    while True:
        kind, _, tag, payload = ep.next(1.0)
        if kind == T.CLOSED:
            break                                    # driver is gone: exit quietly
        if kind != T.MESSAGE:
            continue
        if tag == R.TAG_SHUTDOWN:
            break
        rid = -1
        try:
            rid, body = pickle.loads(payload)
            if tag == R.TAG_CREATE:
                aid, cls, args, kwargs = body
                actors[aid] = cls(*args, **kwargs)
                reply(rid, True, None)
            elif tag == R.TAG_CALL:
                aid, name, args, kwargs = body
                reply(rid, True, getattr(actors[aid], name)(*args, **kwargs))
            elif tag == R.TAG_TASK:
                fn, args, kwargs = body
                reply(rid, True, fn(*args, **kwargs))
            elif tag == R.TAG_RELEASE:
                actors.pop(body[0], None)
                reply(rid, True, None)
        except BaseException as e:                   # report, keep serving (SystemExit included)
            reply(rid, False, f"{type(e).__name__}: {e}\n{traceback.format_exc()}")
            if isinstance(e, KeyboardInterrupt):
                break
    ep.close()


if __name__ == "__main__":
    main()
