#!/usr/bin/env python3
"""Kernel timeline between two solves (the end of one solve's loop, the set-up of the next one up to its first sweep) from a
rocprofv3 --kernel-trace CSV of tools/c2_debug.py.    python tools/timeline_setup.py <kernel_trace.csv> [solve_index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "tlsq::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
firsts = [i for i, r in enumerate(rows) if "k_first_shrink" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(firsts) - 2
i1 = firsts[which]
# back to the last sweep of the previous solve
i0 = max(i for i in range(i1) if "k_zsweep" in rows[i]["Kernel_Name"]) if which > 0 else 0
i2 = min(i for i in range(i1, len(rows)) if "k_zsweep" in rows[i]["Kernel_Name"])
t0 = int(rows[i0]["Start_Timestamp"])
prev = t0
for r in rows[i0:i2 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tlsq::", "")[:44]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev) / 1e3:7.1f} gap  {(e - s) / 1e3:7.1f} us  {name}")
    prev = e
