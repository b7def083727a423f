#!/usr/bin/env python3
"""Whole-solve kernel timeline from a rocprofv3 --kernel-trace CSV, runs of the same kernel folded into one line (count, busy
time, gaps inside the run), a marker line at every sweep; the LAST solve of the trace (a solve starts at the max-abs pass of the
set-up).  For the large configurations, where tools/timeline_full.py prints thousands of lines.
    python tools/timeline_solve.py <kernel_trace.csv> [solve_index_from_end=1]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "tlsq::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("tlsq::", "")
    m = re.match(r"([A-Za-z_0-9]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:46]


starts = [i for i, r in enumerate(rows) if "k_maxabs" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo = starts[-back]
hi = starts[-back + 1] if back > 1 else len(rows)
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
run = None   # [name, start, count, busy, gaps]
sweep = 0
tot_busy = 0


def flush():
    if run:
        print(f"{(run[1] - t0) / 1e3:10.1f} us  {run[0]:46s} x{run[2]:<4d} busy {run[3] / 1e3:9.1f}  gaps {run[4] / 1e3:8.1f}")


for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = short(r["Kernel_Name"])
    if "k_zsweep" in name or "k_first_shrink" in name or "k_update_shrink" in name or "k_fused_zgram" in name:
        flush()
        run = None
        sweep += 1
        print(f"---- sweep {sweep} at {(s - t0) / 1e3:.1f} us")
    gap = max(0, s - prev_end)
    if run and run[0] == name:
        run[2] += 1
        run[3] += e - s
        run[4] += gap
    else:
        flush()
        run = [name, s, 1, e - s, gap]
    tot_busy += e - s
    prev_end = max(prev_end, e)
flush()
print(f"total {(prev_end - t0) / 1e3:.1f} us wall, {tot_busy / 1e3:.1f} us busy")
