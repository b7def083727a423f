#!/bin/bash
# rocprofv3 kernel stats of one python tool:  bash tools/dbg/prof_cmd.sh <tag> <script.py> [args...]   -> gpurun_out/<tag>_kernel_stats.csv
set -u
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag" -- python3 "$@" > "$out/$tag.log" 2>&1
cp "$(find "$out/$tag" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
rm -rf "$out/$tag"
python3 - "$out/${tag}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void tlsq::", "").replace("tlsq::", "").split("(")[0][:60]
    print(f"{n:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us")
PY
