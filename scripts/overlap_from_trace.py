"""Reduce a `rocprofv3 --kernel-trace` CSV of bench.py to what the timed run looks like on the GPU:
per kernel its mean duration and start-to-start spacing, how many kernels are resident together, and which
hardware queues carried them.  VERDICT r2 item 2: the committed kernel statistics were captured on ONE stream,
the bench issues on 16 -- this shows the overlap instead of arguing it.

usage: python scripts/overlap_from_trace.py <kernel_trace.csv> [--kernel SUBSTR] [--last N] [--json OUT]
  --kernel : the dominant kernel (default: the one with the largest total time)
  --last   : only the last N launches of that kernel (default 2000: the steady state, past the warm-up)
"""
import argparse
import csv
import json
import sys
from collections import defaultdict

import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--kernel", default=None)
    ap.add_argument("--last", type=int, default=2000)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    rows = []
    with open(a.trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]))
    tot = defaultdict(int)
    for name, s, e, _ in rows:
        tot[name] += e - s
    dom = max(tot, key=tot.get) if a.kernel is None else next(n for n in tot if a.kernel in n)
    sel = sorted([(s, e, q) for name, s, e, q in rows if name == dom])[-a.last:]
    s = np.array([x[0] for x in sel], dtype=np.float64)
    e = np.array([x[1] for x in sel], dtype=np.float64)
    t_lo, t_hi = s.min(), e.max()
    # everything that ran inside the window, for the residency figures
    inside = [(n, x, y) for n, x, y, _ in rows if y > t_lo and x < t_hi]
    ev = []
    for n, x, y in inside:
        ev.append((max(x, t_lo), 1, n == dom))
        ev.append((min(y, t_hi), -1, n == dom))
    ev.sort()
    cur_all = cur_dom = 0
    last = t_lo
    hist_all, hist_dom = defaultdict(float), defaultdict(float)
    for t, d, is_dom in ev:
        hist_all[cur_all] += t - last
        hist_dom[cur_dom] += t - last
        last = t
        cur_all += d
        if is_dom:
            cur_dom += d
    span = t_hi - t_lo
    others = defaultdict(list)
    for n, x, y in inside:
        if n != dom:
            others[n].append(y - x)
    out = {
        "kernel": dom,
        "launches": len(sel),
        "mean_duration_us": float((e - s).mean() / 1e3),
        "std_duration_us": float((e - s).std() / 1e3),
        "p10_p50_p90_duration_us": [float(np.percentile(e - s, p) / 1e3) for p in (10, 50, 90)],
        "mean_start_to_start_us": float(np.diff(np.sort(s)).mean() / 1e3),
        "window_us": float(span / 1e3),
        "launches_per_window_step_us": float(span / 1e3 / len(sel)),
        "fraction_of_time_with_n_of_this_kernel_resident": {str(k): round(v / span, 4) for k, v in sorted(hist_dom.items())},
        "fraction_of_time_with_n_kernels_resident": {str(k): round(v / span, 4) for k, v in sorted(hist_all.items())},
        "mean_kernels_resident": float(sum(k * v for k, v in hist_all.items()) / span),
        "hardware_queues": sorted({q for _, _, q in sel}),
        "other_kernels_in_window": {n[:80]: {"calls": len(v), "mean_us": float(np.mean(v) / 1e3)} for n, v in others.items()},
    }
    try:  # tie the record to the kernel sources it was taken on (bench.py drops stale records)
        from bench import kernel_source_hash

        out["kernel_source_hash"] = kernel_source_hash()
    except Exception:  # noqa: BLE001
        pass
    txt = json.dumps(out, indent=1)
    print(txt)
    if a.json:
        with open(a.json, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    sys.exit(main())
