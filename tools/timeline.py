#!/usr/bin/env python3
"""Print the kernel timeline of one ALM iteration from a rocprofv3 --kernel-trace CSV (gaps show host round trips).
    python tools/timeline.py <kernel_trace.csv> [iteration_index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "tlsq::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# iterations are delimited by the Gram kernel of Z (k_gemm_f64<true, true...>) that follows a sweep
marks = [i for i, r in enumerate(rows) if "k_update_shrink" in r["Kernel_Name"] or "k_rebuild_update_shrink" in r["Kernel_Name"] or "k_zsweep" in r["Kernel_Name"]]
it = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
lo, hi = marks[it] , marks[it + 1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tlsq::", "")[:40]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {name}")
    busy += e - s
    prev_end = e
print(f"iteration: {(prev_end - t0) / 1e3:.1f} us wall, {busy / 1e3:.1f} us busy")
