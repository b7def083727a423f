#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from a rocprofv3 counter pass (SURVEY 8d "Counters"):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...
usage: summarize_mfma.py <dir> <out_csv>
Columns: kernel, dispatches, average SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES per launch (summed over the shader engines
as rocprofv3 reports them) and their ratio.  Only tlsq:: kernels are kept."""
import csv, glob, os, sys
from collections import defaultdict


def main():
    d, out = sys.argv[1:3]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if "tlsq::" not in row["Kernel_Name"]:
                    continue
                a = acc[row["Kernel_Name"]][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    rows = []
    for k, c in acc.items():
        n = max(v[0] for v in c.values())
        mf = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / max(c["SQ_VALU_MFMA_BUSY_CYCLES"][0], 1)
        bz = c["SQ_BUSY_CYCLES"][1] / max(c["SQ_BUSY_CYCLES"][0], 1)
        rows.append((k, n, int(mf), int(bz), round(mf / bz, 4) if bz else 0.0))
    rows.sort(key=lambda r: -r[2] * r[1])
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches", "SQ_VALU_MFMA_BUSY_CYCLES_avg", "SQ_BUSY_CYCLES_avg", "ratio"])
        w.writerows(rows)
    for r in rows[:8]:
        print(r[0][:70], r[1], r[2], r[3], r[4])


if __name__ == "__main__":
    main()
