#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --stats kernel_stats.csv:  python tools/kstats.py <kernel_stats.csv> [solves=1] [rows=16]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
solves = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    n = r["Name"].split("(")[0].replace("void ", "").replace("tlsq::", "")[:50]
    print("  %-50s %6s x %9.1f us  %8.2f ms/solve  %s%%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                             float(r["TotalDurationNs"]) / 1e6 / solves, r["Percentage"]))
