"""Summarise rocprofv3 --pmc runs of bench.py (eager): per kernel, mean counter values per dispatch over the
last `--last N` dispatches of each run (whole timed steps: warm-up and plan autotuning come first), HBM traffic
and the derived rates; with `--steps S` (N = S whole steps) also the HBM bytes of one step.
usage: pmc_summary.py [--last N] [--steps S] <run dir> [<run dir> ...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:60]


LAST = 285
STEPS = 0
TOTALS = defaultdict(float)         # counter -> sum over the window (first run that collected it)


def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            rows += list(csv.DictReader(fh))
    if not rows:
        raise SystemExit(f"no counter_collection.csv under {d}")
    by_disp = defaultdict(dict)
    meta = {}
    for r in rows:
        did = int(r["Dispatch_Id"])
        by_disp[did][r["Counter_Name"]] = float(r["Counter_Value"])
        if did not in meta:
            dur = None
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            meta[did] = (short(r["Kernel_Name"]), dur)
    ids = sorted(by_disp)
    ids = ids[-LAST:]
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    run_tot = defaultdict(float)
    for did in ids:
        k, dur = meta[did]
        cnt[k] += 1
        for c, v in by_disp[did].items():
            agg[k][c] += v
            run_tot[c] += v
        if dur is not None:
            agg[k]["_us"] += dur
    for c, v in run_tot.items():
        TOTALS.setdefault(c, v)
    return {k: {c: v / cnt[k] for c, v in cs.items()} | {"_calls": cnt[k]} for k, cs in agg.items()}


def main():
    global LAST
    argv = sys.argv[1:]
    global STEPS
    while argv and argv[0] in ("--last", "--steps"):
        if argv[0] == "--last":
            LAST = int(argv[1])
        else:
            STEPS = int(argv[1])
        argv = argv[2:]
    merged = defaultdict(dict)
    for d in argv:
        for k, cs in load(d).items():
            for c, v in cs.items():
                merged[k].setdefault(c, v)
    cols = ["_calls", "_us", "FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
            "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY"]
    print("per-dispatch means over the timed steps; `us` is the dispatch time UNDER counter collection "
          "(serialised, slower than in the kernel trace); HBM GB/s = (FETCH_SIZE + WRITE_SIZE) KB / us; "
          "mfma = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs), MFMA-pipe busy cycles per CU per GPU-active cycle "
          "(a dense back-to-back MFMA loop reads 0.5 on this scale -- calibrated in round 3 -- so 0.19 is 38 % of the matrix pipes' time); lds conflict % = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
          "wait % = SQ_WAIT_ANY / SQ_WAVE_CYCLES\n")
    print("| kernel | calls | us | read MB | write MB | HBM GB/s | mfma | lds conflict % | wait % | waves |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    order = sorted(merged, key=lambda k: -merged[k].get("_us", 0) * merged[k].get("_calls", 0))
    for k in order:
        m = merged[k]
        us = m.get("_us", 0.0)
        has = lambda c: c in m
        f = lambda v, fmt: format(v, fmt) if v is not None else "-"
        rd = m["FETCH_SIZE"] / 1024 if has("FETCH_SIZE") else None
        wr = m["WRITE_SIZE"] / 1024 if has("WRITE_SIZE") else None
        gbs = (rd + wr) * 1e-3 / (us * 1e-6) if us and rd is not None and wr is not None else None
        mf = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 256) \
            if has("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("GRBM_GUI_ACTIVE") else None
        lc = 100 * m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") else None
        wt = 100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None
        print(f"| `{k}` | {m['_calls']} | {us:.1f} | {f(rd, '.2f')} | {f(wr, '.2f')} | {f(gbs, '.0f')} | {f(mf, '.2f')} | "
              f"{f(lc, '.1f')} | {f(wt, '.0f')} | {f(m.get('SQ_WAVES'), '.0f')} |")
    if STEPS and "FETCH_SIZE" in TOTALS and "WRITE_SIZE" in TOTALS:
        rd, wr = TOTALS["FETCH_SIZE"] / 1024 / STEPS, TOTALS["WRITE_SIZE"] / 1024 / STEPS
        print(f"\n**HBM traffic per step** ({LAST // STEPS} dispatches): read {rd:.0f} MB + write {wr:.0f} MB = "
              f"{(rd + wr) / 1000:.2f} GB")
    print("\n`-`: not defined for that kernel (no LDS traffic), or the counter's run used a different kernel variant "
          "for that problem (tools/pmc_step.sh tunes the plans once, without counters, so that all passes agree).")

if __name__ == "__main__":
    main()
