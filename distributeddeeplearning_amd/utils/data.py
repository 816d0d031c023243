"""Data layer: MNIST (if present on disk) or a synthetic MNIST-shaped stand-in, device resident.

Mirrors the reference's `Data` class (/root/reference/DDL.py:117-130, DDL_Normal3.py:160-173):
raw uint8 pixels (NO normalisation), one-hot float labels, and `get_random_sample()` = a fresh
random permutation of the 60k training images split into train / validation.

MI355X-first differences: the reference pickles ~146 MB of shuffled images to
every Ray actor.  Here the uint8 dataset sits once in each GPU's HBM (47 MB) and a "sample" is
just an int64 permutation; the gather + uint8->bf16 cast + channel replicate + pad happens in one
kernel when a batch is formed (`Network.prepare_input(images, index=...)`).

There is no network in the build environment, so when no MNIST files are found the loader
synthesises a learnable digit dataset (rendered glyphs + jitter + noise) of identical shape/dtype.
"""
from __future__ import annotations

import gzip
import os
import struct
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Tuple

import numpy as np
import torch

_SEARCH = ("~/.keras/datasets/mnist.npz", "./data/mnist.npz", "/data/mnist.npz", "./data/mnist", "/data/mnist")


def one_hot(labels: torch.Tensor, dim: int) -> torch.Tensor:
    """/root/reference/DDL.py:5-10 (float one-hot), vectorised."""
    return torch.nn.functional.one_hot(labels.long().reshape(-1), dim).float()


def _read_idx(path: Path) -> np.ndarray:
    op = gzip.open if path.suffix == ".gz" else open
    with op(path, "rb") as f:
        magic, = struct.unpack(">I", f.read(4))
        nd = magic & 0xFF
        dims = struct.unpack(">" + "I" * nd, f.read(4 * nd))
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(dims)


def _try_load_mnist(root: Optional[str]):
    cands = [root] if root else []
    cands += [os.environ.get("DDL_MNIST", "")] + list(_SEARCH)
    for c in cands:
        if not c:
            continue
        p = Path(os.path.expanduser(c))
        if p.is_file() and p.suffix == ".npz":
            with np.load(p, allow_pickle=False) as z:
                return (z["x_train"], z["y_train"]), (z["x_test"], z["y_test"])
        if p.is_dir():
            def pick(stem):
                for ext in ("", ".gz"):
                    q = p / (stem + ext)
                    if q.exists():
                        return q
                return None
            files = [pick(s) for s in ("train-images-idx3-ubyte", "train-labels-idx1-ubyte",
                                       "t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte")]
            if all(files):
                a = [_read_idx(f) for f in files]
                return (a[0], a[1]), (a[2], a[3])
    return None


_GLYPHS = {  # 5x7 bitmaps
    0: "01110 10001 10011 10101 11001 10001 01110", 1: "00100 01100 00100 00100 00100 00100 01110",
    2: "01110 10001 00001 00010 00100 01000 11111", 3: "11110 00001 00001 01110 00001 00001 11110",
    4: "00010 00110 01010 10010 11111 00010 00010", 5: "11111 10000 11110 00001 00001 10001 01110",
    6: "00110 01000 10000 11110 10001 10001 01110", 7: "11111 00001 00010 00100 01000 01000 01000",
    8: "01110 10001 10001 01110 10001 10001 01110", 9: "01110 10001 10001 01111 00001 00010 01100",
}


def synthetic_mnist(n_train: int = 60000, n_test: int = 10000, seed: int = 0):
    """MNIST-shaped (uint8 28x28, labels 0..9) learnable stand-in: upscaled 5x7 glyphs with random
    shift, thickness, intensity and pixel noise."""
    rng = np.random.default_rng(seed)
    base = np.zeros((10, 7, 5), np.float32)
    for d, s in _GLYPHS.items():
        base[d] = np.array([[int(c) for c in row] for row in s.split()], np.float32)
    big = np.kron(base, np.ones((3, 3), np.float32))          # 10 x 21 x 15

    def make(n):
        y = rng.integers(0, 10, n)
        x = np.zeros((n, 28, 28), np.float32)
        dy = rng.integers(0, 8, n)
        dx = rng.integers(0, 14, n)
        for i in range(n):
            x[i, dy[i]:dy[i] + 21, dx[i]:dx[i] + 15] = big[y[i]]
        x *= rng.uniform(0.6, 1.0, (n, 1, 1)).astype(np.float32)
        x += rng.normal(0, 0.08, x.shape).astype(np.float32)
        return (np.clip(x, 0, 1) * 255).astype(np.uint8), y.astype(np.uint8)

    return make(n_train), make(n_test)


@dataclass
class Sample:
    """One member's private train/validation split: index tensors into the resident dataset."""
    train_idx: torch.Tensor
    val_idx: torch.Tensor


class Data:
    def __init__(self, device="cpu", channels: int = 3, root: Optional[str] = None, seed: int = 0,
                 n_train: int = 60000, n_test: int = 10000, synthetic: Optional[bool] = None):
        loaded = None if synthetic else _try_load_mnist(root)
        if loaded is None:
            if synthetic is False:
                raise FileNotFoundError("MNIST not found (set DDL_MNIST or pass root=)")
            loaded = synthetic_mnist(n_train, n_test, seed)
            self.source = "synthetic"
        else:
            self.source = "mnist"
        (tx, ty), (vx, vy) = loaded
        self.device = torch.device(device)
        self.channels = channels
        # resident uint8 images [N,28,28,1]; channel replication happens in the batch kernel
        self.train_x = torch.from_numpy(np.ascontiguousarray(tx[:n_train])).unsqueeze(-1).to(self.device)
        self.test_x = torch.from_numpy(np.ascontiguousarray(vx[:n_test])).unsqueeze(-1).to(self.device)
        self.train_labels = torch.from_numpy(np.ascontiguousarray(ty[:n_train]).astype(np.int64)).to(self.device)
        self.test_labels = torch.from_numpy(np.ascontiguousarray(vy[:n_test]).astype(np.int64)).to(self.device)
        self.train_y = one_hot(self.train_labels, 10)
        self.test_y = one_hot(self.test_labels, 10)
        self.gen = torch.Generator().manual_seed(seed + 1)

    @property
    def input_shape(self):
        return (28, 28, self.channels)

    def get_random_sample(self, data_train_size: int, seed: Optional[int] = None) -> Sample:
        """Random permutation of the training set -> first `data_train_size` train, rest validation."""
        gen = self.gen if seed is None else torch.Generator().manual_seed(seed)
        perm = torch.randperm(self.train_x.shape[0], generator=gen).to(self.device)
        return Sample(perm[:data_train_size].contiguous(), perm[data_train_size:].contiguous())

    get_random_sample_dataset = get_random_sample      # DDL_Normal4.py:189 name
