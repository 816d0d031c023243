#!/usr/bin/env python3
"""In-tree native build: `python build.py [--force] [--only hip|host|comm]`.

Produces, under distributeddeeplearning_amd/_C/ :
  libddl_hip.so   every gfx950 kernel in csrc/hip/*.hip           (hipcc --offload-arch=gfx950)
  libddl_host.so  CPU-only runtime: HDF5 subset, actor transport   (g++)
  libddl_comm.so  RCCL communicator + xGMI peer all-reduce         (hipcc, links librccl)
`python build.py --sanitize asan|ubsan|tsan` builds the host C++ with a CPU sanitizer into
build/sanitize/host_sanitize.<kind> (HDF5 mutation fuzzer + transport stress; tests/test_sanitizers.py runs them).

hipcc cross-compiles without a GPU, so this runs in the CPU build container; the .so files are
git-ignored but travel to the GPU box with the tree.  Objects are cached by source mtime.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OUT = ROOT / "distributeddeeplearning_amd" / "_C"
OBJ = ROOT / "build" / "obj"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
HIPCC = str(ROCM / "bin" / "hipcc")
ARCH = "gfx950"


def _newer(src: Path, dst: Path, deps=()) -> bool:
    if not dst.exists():
        return True
    t = dst.stat().st_mtime
    return any(p.stat().st_mtime > t for p in (src, *deps))


def _run(cmd):
    print(" ".join(str(c) for c in cmd), flush=True)
    subprocess.run([str(c) for c in cmd], check=True)


def build_hip(force=False, jobs=4):
    srcs = sorted((CSRC / "hip").glob("*.hip"))
    hdrs = sorted((CSRC / "hip").glob("*.h"))
    OBJ.mkdir(parents=True, exist_ok=True)
    objs, todo = [], []
    for s in srcs:
        o = OBJ / (s.stem + ".o")
        objs.append(o)
        if force or _newer(s, o, hdrs):
            todo.append([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                         "-fvisibility=hidden", "-c", s, "-o", o])
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(_run, todo))
    lib = OUT / "libddl_hip.so"
    if force or todo or not lib.exists():
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


def build_host(force=False):
    srcs = sorted((CSRC / "host").glob("*.cpp"))
    if not srcs:
        return None
    hdrs = sorted((CSRC / "host").glob("*.h"))
    lib = OUT / "libddl_host.so"
    if force or any(_newer(s, lib, hdrs) for s in srcs):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-pthread", "-o", lib, *srcs])
    return lib


def build_comm(force=False):
    srcs = sorted((CSRC / "comm").glob("*.cpp")) + sorted((CSRC / "comm").glob("*.hip"))
    if not srcs:
        return None
    hdrs = sorted((CSRC / "comm").glob("*.h")) + sorted((CSRC / "hip").glob("*.h"))     # p2p shares optim_update.h
    lib = OUT / "libddl_comm.so"
    if force or any(_newer(s, lib, hdrs) for s in srcs):
        _run([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
              "-Wno-unused-value", f"-I{ROCM / 'include'}", "-o", lib, *srcs, f"-L{ROCM / 'lib'}", "-lrccl"])
    return lib


SANITIZERS = {"asan": ["-fsanitize=address", "-fno-omit-frame-pointer"],
              "ubsan": ["-fsanitize=undefined", "-fno-sanitize-recover=undefined"],
              "tsan": ["-fsanitize=thread"]}


def build_sanitized(kind: str) -> Path:
    """CPU-only sanitizer build of the host runtime (HDF5 subset + actor transport) linked with its native test
    driver csrc/host/tests/host_sanitize.cpp -> build/sanitize/host_sanitize.<kind>.  Never touches the GPU code."""
    if kind not in SANITIZERS:
        raise ValueError(f"unknown sanitizer {kind!r}; choose from {sorted(SANITIZERS)}")
    out_dir = ROOT / "build" / "sanitize"
    out_dir.mkdir(parents=True, exist_ok=True)
    exe = out_dir / f"host_sanitize.{kind}"
    srcs = sorted((CSRC / "host").glob("*.cpp")) + [CSRC / "host" / "tests" / "host_sanitize.cpp"]
    if any(_newer(s, exe) for s in srcs):
        _run(["g++", "-O1", "-g", "-std=c++17", "-pthread", *SANITIZERS[kind], "-o", exe, *srcs])
    return exe


def build_all(force=False, only=None):
    OUT.mkdir(parents=True, exist_ok=True)
    libs = []
    if only in (None, "hip"):
        libs.append(build_hip(force))
    if only in (None, "host"):
        libs.append(build_host(force))
    if only in (None, "comm"):
        libs.append(build_comm(force))
    return [l for l in libs if l]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--only", choices=["hip", "host", "comm"])
    ap.add_argument("--sanitize", choices=sorted(SANITIZERS),
                    help="build the host C++ (h5_min, actor_transport) with a CPU sanitizer plus its test driver")
    a = ap.parse_args()
    if a.sanitize:
        print("built", build_sanitized(a.sanitize))
        sys.exit(0)
    for l in build_all(a.force, a.only):
        print("built", l)
