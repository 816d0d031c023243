"""Python face of csrc/host/h5_min.cpp: read / write the Keras `save_weights` HDF5 layout.

Layout (written by /root/reference/DDL_Normal4.py:172): root attrs `layer_names`,
`backend`, `keras_version`; one group per layer with attr `weight_names`
(e.g. `conv2d_1/kernel:0`), float32 datasets at `/<layer>/<layer>/<weight>:0`.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Dict, List, Tuple

import numpy as np

_LIB = Path(__file__).resolve().parent.parent / "_C" / "libddl_host.so"
if not _LIB.exists():
    raise ImportError(f"{_LIB} is missing: run `python build.py --only host`")
_lib = C.CDLL(str(_LIB))
_lib.h5min_last_error.restype = C.c_char_p
_lib.h5min_open.restype = C.c_void_p
_lib.h5min_open.argtypes = [C.c_char_p]
_lib.h5min_close.argtypes = [C.c_void_p]
for _n in ("h5min_list", "h5min_attr_names"):
    getattr(_lib, _n).argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
_lib.h5min_is_group.argtypes = [C.c_void_p, C.c_char_p]
_lib.h5min_attr_strings.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
_lib.h5min_dataset_info.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
_lib.h5min_read_f32.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
_lib.h5min_create.restype = C.c_void_p
_lib.h5min_create.argtypes = [C.c_char_p]
_lib.h5min_add_group.argtypes = [C.c_void_p, C.c_char_p]
_lib.h5min_add_attr_strings.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
_lib.h5min_add_dataset_f32.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_int, C.c_void_p]
_lib.h5min_finalize.argtypes = [C.c_void_p]


class H5Error(IOError):
    pass


def _err():
    return H5Error(_lib.h5min_last_error().decode("utf8", "replace"))


class H5File:
    """Read-only view of an HDF5 file (whole file is parsed at open)."""

    def __init__(self, path):
        self._h = _lib.h5min_open(str(path).encode())
        if not self._h:
            raise _err()

    def close(self):
        if self._h:
            _lib.h5min_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _text(self, fn, *args) -> List[str]:
        n = fn(self._h, *args, None, 0)
        if n < 0:
            raise _err()
        buf = C.create_string_buffer(n + 1)
        fn(self._h, *args, buf, n + 1)
        s = buf.value.decode("utf8")
        return s.split("\n") if s else []

    def list(self, path="/") -> List[str]:
        return self._text(_lib.h5min_list, path.encode())

    def is_group(self, path) -> bool:
        r = _lib.h5min_is_group(self._h, path.encode())
        if r < 0:
            raise _err()
        return bool(r)

    def attr_names(self, path="/") -> List[str]:
        return self._text(_lib.h5min_attr_names, path.encode())

    def attr_strings(self, path, name) -> List[str]:
        return self._text(_lib.h5min_attr_strings, path.encode(), name.encode())

    def shape(self, path) -> Tuple[int, ...]:
        dims = (C.c_uint64 * 8)()
        dt = C.c_int()
        r = _lib.h5min_dataset_info(self._h, path.encode(), dims, C.byref(dt))
        if r < 0:
            raise _err()
        return tuple(int(dims[i]) for i in range(r))

    def read(self, path) -> np.ndarray:
        shp = self.shape(path)
        out = np.empty(shp, dtype=np.float32)
        if _lib.h5min_read_f32(self._h, path.encode(), out.ctypes.data_as(C.c_void_p), out.size) < 0:
            raise _err()
        return out


def read_keras_weights(path) -> Tuple[List[str], Dict[Tuple[str, str], np.ndarray], Dict[str, str]]:
    """-> (layer_names, {(layer, weight_name_without_':0'): array}, root string attrs)."""
    with H5File(path) as f:
        layers = f.attr_strings("/", "layer_names")
        meta = {}
        for a in f.attr_names("/"):
            if a != "layer_names":
                try:
                    meta[a] = "\n".join(f.attr_strings("/", a))
                except H5Error:
                    pass
        weights = {}
        for ln in layers:
            for wn in f.attr_strings("/" + ln, "weight_names"):
                arr = f.read(f"/{ln}/{wn}")
                short = wn.split("/")[-1]
                if short.endswith(":0"):
                    short = short[:-2]
                weights[(ln, short)] = arr
    return layers, weights, meta


def write_keras_weights(path, layer_names: List[str], weights: List[Tuple[str, str, np.ndarray]],
                        backend="tensorflow", keras_version="2.2.4"):
    """`weights`: [(layer, weight_name, array)] in Keras order; every layer of `layer_names` gets
    a group (weightless layers an empty `weight_names`)."""
    h = _lib.h5min_create(str(path).encode())
    if not h:
        raise _err()

    def chk(r):
        if r < 0:
            raise _err()

    chk(_lib.h5min_add_attr_strings(h, b"/", b"layer_names", "\n".join(layer_names).encode(), 0))
    chk(_lib.h5min_add_attr_strings(h, b"/", b"backend", backend.encode(), 1))
    chk(_lib.h5min_add_attr_strings(h, b"/", b"keras_version", keras_version.encode(), 1))
    by_layer: Dict[str, List[Tuple[str, np.ndarray]]] = {ln: [] for ln in layer_names}
    for ln, wn, arr in weights:
        by_layer[ln].append((wn, arr))
    for ln in layer_names:
        chk(_lib.h5min_add_group(h, f"/{ln}".encode()))
        names = [f"{ln}/{wn}:0" for wn, _ in by_layer[ln]]
        chk(_lib.h5min_add_attr_strings(h, f"/{ln}".encode(), b"weight_names", "\n".join(names).encode(), 0))
        for wn, arr in by_layer[ln]:
            a = np.ascontiguousarray(arr, dtype=np.float32)
            dims = (C.c_uint64 * max(a.ndim, 1))(*a.shape)
            chk(_lib.h5min_add_dataset_f32(h, f"/{ln}/{ln}/{wn}:0".encode(), dims, a.ndim,
                                           a.ctypes.data_as(C.c_void_p)))
    chk(_lib.h5min_finalize(h))
