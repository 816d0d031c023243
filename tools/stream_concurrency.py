"""Do small kernels issued on different HIP streams (eagerly or as hipGraph replays) overlap on this GPU?
Prints wall time for the same total work on 1, 2, 4 and 8 streams."""
import sys
import time

import torch

dev = torch.device("cuda")
N_K = 200          # kernels per chain


def chain(x):
    for _ in range(N_K):
        x.mul_(1.0001)


def run(n_streams, mode, total_chains=8, size=4096):
    xs = [torch.ones(size, device=dev) for _ in range(total_chains)]
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    graphs = []
    if mode == "graph":
        for x in xs:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                chain(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                chain(x)
            graphs.append(g)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for i, x in enumerate(xs):
            with torch.cuda.stream(streams[i % n_streams]):
                if mode == "graph":
                    graphs[i].replay()
                else:
                    chain(x)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


for size in (4096, 1 << 22):
    for mode in ("eager", "graph"):
        res = [run(n, mode, size=size) for n in (1, 2, 4, 8)]
        print(f"size {size:8d} {mode:5s}: 8 chains x {N_K} kernels on 1/2/4/8 streams: " + " ".join(f"{r:7.2f} ms" for r in res), flush=True)
