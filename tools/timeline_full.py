#!/usr/bin/env python3
"""Whole-solve kernel timeline from a rocprofv3 --kernel-trace CSV: every kernel of the LAST solve in the trace (a solve
starts at k_maxabs), with the gap to its predecessor, and totals for set-up / first iteration / steady iterations / tail.
    python tools/timeline_full.py <kernel_trace.csv> [solve_index_from_end=1]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "tlsq::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_maxabs" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo = starts[-back]
hi = starts[-back + 1] if back > 1 else len(rows)
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
sweeps = 0
sect = "setup"
tot = {}
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tlsq::", "")[:44]
    if "k_first_shrink" in name:
        sect = "iter1"
    elif "k_zsweep" in name or "k_update_shrink" in name:
        sweeps += 1
        sect = f"it{sweeps + 1:02d}"
    elif "k_final_e" in name:
        sect = "tail"
    print(f"{sect:6s} {(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {name}")
    b = tot.setdefault(sect, [s, e, 0])
    b[1] = e
    b[2] += e - s
    prev_end = e
print("---- sections (wall us, busy us)")
for k, (s, e, busy) in tot.items():
    print(f"{k:6s} start {(s - t0) / 1e3:9.1f}  wall {(e - s) / 1e3:8.1f}  busy {busy / 1e3:8.1f}")
print(f"solve: {(prev_end - t0) / 1e3:.1f} us wall")
