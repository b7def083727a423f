#!/usr/bin/env python3
"""Compressed kernel timeline of the LAST solve in a rocprofv3 --kernel-trace CSV: consecutive launches of the same kernel are
merged into one line (count, busy time, gaps), a marker line is printed at every sweep (= ALM iteration boundary).
    python tools/dbg/trace_runs.py <kernel_trace.csv> [first-kernel-substring-of-a-solve, default k_maxabs]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else "k_maxabs"
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
lo = starts[-1] if starts else 0
rows = rows[lo:]
short = lambda n: n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("tlsq::", "")[:44]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
cur = None
def flush():
    if cur:
        print(f"{(cur['t'] - t0) / 1e3:10.1f} us  {cur['name']:44s} x{cur['n']:<4d} busy {cur['busy'] / 1e3:9.1f}  gaps {cur['gap'] / 1e3:8.1f}")
it = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = short(r["Kernel_Name"])
    if "zsweep" in n or "first_shrink" in n or "fused_zgram" in n:
        flush(); cur = None
        it += 1
        print(f"---- sweep {it} at {(s - t0) / 1e3:.1f} us")
    if cur and cur["name"] == n:
        cur["n"] += 1; cur["busy"] += e - s; cur["gap"] += max(0, s - prev_end)
    else:
        flush()
        cur = dict(name=n, n=1, busy=e - s, gap=max(0, s - prev_end), t=s)
    prev_end = max(prev_end, e)
flush()
print(f"total {(prev_end - t0) / 1e3:.1f} us")
