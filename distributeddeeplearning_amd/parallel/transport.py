"""ctypes face of csrc/host/actor_transport.cpp: an `Endpoint` exchanging tagged frames with peers
over Unix-domain sockets; a native epoll thread reassembles frames into one event queue."""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional, Tuple

_LIB = Path(__file__).resolve().parent.parent / "_C" / "libddl_host.so"
if not _LIB.exists():
    raise ImportError(f"{_LIB} is missing: run `python build.py --only host`")
_lib = C.CDLL(str(_LIB))
_lib.ddl_ep_last_error.restype = C.c_char_p
_lib.ddl_ep_create.restype = C.c_void_p
_lib.ddl_ep_listen.argtypes = [C.c_void_p, C.c_char_p]
_lib.ddl_ep_connect.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
_lib.ddl_ep_send.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64]
_lib.ddl_ep_next.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
_lib.ddl_ep_take.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
_lib.ddl_ep_num_conns.argtypes = [C.c_void_p]
_lib.ddl_ep_close_conn.argtypes = [C.c_void_p, C.c_int]
_lib.ddl_ep_destroy.argtypes = [C.c_void_p]

TIMEOUT, MESSAGE, CONNECTED, CLOSED = 0, 1, 2, 3


class TransportError(IOError):
    pass


def _err() -> TransportError:
    return TransportError(_lib.ddl_ep_last_error().decode("utf8", "replace"))


class Endpoint:
    def __init__(self):
        self._h = _lib.ddl_ep_create()
        if not self._h:
            raise _err()

    def listen(self, path: str):
        if _lib.ddl_ep_listen(self._h, path.encode()) < 0:
            raise _err()

    def connect(self, path: str, timeout: float = 30.0) -> int:
        c = _lib.ddl_ep_connect(self._h, path.encode(), int(timeout * 1000))
        if c < 0:
            raise _err()
        return c

    def send(self, conn: int, tag: int, payload: bytes):
        buf = (C.c_char * len(payload)).from_buffer_copy(payload) if payload else None
        if _lib.ddl_ep_send(self._h, conn, tag, buf, len(payload)) < 0:
            raise _err()

    def next(self, timeout: float = 1.0) -> Tuple[int, int, int, Optional[bytes]]:
        """-> (kind, conn, tag, payload).  kind is TIMEOUT / MESSAGE / CONNECTED / CLOSED."""
        conn, tag, size = C.c_int(), C.c_uint64(), C.c_uint64()
        kind = _lib.ddl_ep_next(self._h, int(timeout * 1000), C.byref(conn), C.byref(tag), C.byref(size))
        payload = None
        if kind == MESSAGE:
            buf = C.create_string_buffer(max(size.value, 1))
            if _lib.ddl_ep_take(self._h, buf, size.value) < 0:
                raise _err()
            payload = buf.raw[:size.value]
        return kind, conn.value, tag.value, payload

    def num_conns(self) -> int:
        return _lib.ddl_ep_num_conns(self._h)

    def close_conn(self, conn: int):
        _lib.ddl_ep_close_conn(self._h, conn)

    def close(self):
        if self._h:
            _lib.ddl_ep_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
