#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (results.db or *_kernel_trace.csv) into a per-kernel table.

    python tools/rocprof_summary.py gpurun_out/prof2 [--skip-first N] > profiles/xyz.md
"""
import csv
import glob
import re
import sqlite3
import sys
from collections import defaultdict


def load(path):
    rows = []          # (name, start, end)
    dbs = glob.glob(f"{path}/**/*_results.db", recursive=True)
    if dbs:
        db = sqlite3.connect(dbs[0])
        cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
        name = "name" if "name" in cols else "kernel_name"
        grid = "grid_x, grid_y, grid_z, workgroup_x" if "grid_x" in cols else "0, 0, 0, 0"
        q = "queue || ' / ' || stream" if "queue" in cols and "stream" in cols else "''"
        for n, s, e, gx, gy, gz, wx, qi in db.execute(
                f"select {name}, start, end, {grid}, {q} from kernels order by start"):
            rows.append((n, s, e, (gx, gy, gz, wx), qi))
        return rows
    for f in glob.glob(f"{path}/**/*_kernel_trace.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), None, int(r.get("Queue_Id", 0) or 0)))
    rows.sort(key=lambda r: r[1])
    return rows


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void ", "")
    return n[:70]


def main():
    path = sys.argv[1]
    rows = load(path)
    if not rows:
        sys.exit("no kernel records found")
    steps = None
    if "--steady" in sys.argv:
        # keep the last few training steps (optimizer kernel to optimizer kernel): warm-up, autotuning
        # and graph capture are over by then.  The fused optimizer kernel runs twice per GAN step and
        # once per network of the 5-replica classifier step.
        per = 2 if "--gan" in sys.argv else 5
        opt = [i for i, r in enumerate(rows) if "optim_kernel" in r[0]]
        total_steps = len(opt) // per
        steps = max(1, min(8, total_steps // 2))
        rows = rows[opt[-steps * per - 1] + 1: opt[-1] + 1]
    if "--timeline" in sys.argv:
        # the last complete training step as a timeline: start / end relative to the step's first kernel, the hardware
        # queue of each dispatch, and which other dispatches it ran beside (kernels of different HIP streams land on
        # different queues: a collective on the communication stream shows up overlapping the backward kernels)
        # step boundary: the per-step noise draw (GAN) -- the fused all-reduce + optimizer path has no optimizer kernel
        mark = [i for i, r in enumerate(rows) if "philox_uniform_kernel" in r[0]]
        if len(mark) < 2:
            sys.exit("no step markers")
        lo, hi = mark[-2], mark[-1]
        t0 = rows[lo][1]
        print("| # | queue | kernel | start us | end us | runs beside # |")
        print("|---:|---:|---|---:|---:|---|")
        over_us = 0.0
        for i in range(lo, hi):
            n, s_, e_, g, q = rows[i]
            beside = [j - lo for j in range(lo, hi) if j != i and rows[j][1] < e_ and rows[j][2] > s_]
            if beside and q != rows[lo][4]:
                over_us += (min(e_, max(rows[j + lo][2] for j in beside)) - max(s_, min(rows[j + lo][1] for j in beside))) / 1e3
            print(f"| {i - lo} | {q} | `{short(n)}` | {(s_ - t0)/1e3:.1f} | {(e_ - t0)/1e3:.1f} | "
                  f"{', '.join(map(str, beside))} |")
        busy = sum(r[2] - r[1] for r in rows[lo:hi])
        print(f"\nkernel time {busy/1e3:.1f} us, span {(rows[hi-1][2] - t0)/1e3:.1f} us; "
              f"time of side-queue dispatches spent beside a main-queue kernel: {over_us:.1f} us")
        return
    if "--sequence" in sys.argv:
        # the last complete training step in launch order (optimizer kernel to optimizer kernel)
        per = 2 if "--gan" in sys.argv else 5
        opt = [i for i, r in enumerate(rows) if "optim_kernel" in r[0]]
        lo, hi = opt[-per - 1] + 1, opt[-1] + 1
        print("| # | kernel | us | gap us | grid (threads) | block |")
        print("|---:|---|---:|---:|---|---:|")
        tot = 0
        for i in range(lo, hi):
            n, s_, e_, g = rows[i][:4]
            tot += e_ - s_
            gs = "x".join(str(v) for v in g[:3]) if g else ""
            print(f"| {i - lo} | `{short(n)}` | {(e_ - s_)/1e3:.1f} | {(s_ - rows[i-1][2])/1e3:.1f} | {gs} | {g[3] if g else ''} |")
        print(f"\nkernel time {tot/1e3:.1f} us, span {(rows[hi-1][2] - rows[lo][1])/1e3:.1f} us")
        return
    agg = defaultdict(lambda: [0, 0, 10**18, 0])
    for n, s, e, *_ in rows:
        a = agg[short(n)]
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    span = rows[-1][2] - rows[0][1]
    print(f"kernels: {len(rows)} dispatches, {tot/1e6:.3f} ms busy, {span/1e6:.3f} ms first-to-last span")
    if steps:
        print(f"steady-state window: {steps} training steps -> {tot/1e6/steps:.3f} ms of kernel time per step, "
              f"{len(rows)/steps:.0f} dispatches per step")
    print()
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        extra = f" {a[1]/1e3/steps:.1f} us/step" if steps else ""
        print(f"| `{n}` | {a[0]} | {a[1]/1e3:.1f} | {a[1]/a[0]/1e3:.2f} | {a[2]/1e3:.2f} | {a[3]/1e3:.2f} | {100*a[1]/tot:.1f}{extra} |")


if __name__ == "__main__":
    main()
