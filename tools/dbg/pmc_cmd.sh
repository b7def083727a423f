#!/bin/bash
# PMC counters of one python tool, per kernel (average per dispatch):
#   bash tools/dbg/pmc_cmd.sh <tag> "<COUNTER1 COUNTER2 ...>" <kernel-substring> <script.py> [args...]
set -u
tag=$1; ctrs=$2; pat=$3; shift 3
export TMPDIR=/tmp
out=$PWD/gpurun_out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out/$tag" -- python3 "$@" > "$out/$tag.log" 2>&1
f=$(find "$out/$tag" -name '*counter_collection.csv' | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    if sys.argv[2] not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(k[:80], "dispatches", len(n[k]))
    for c, v in sorted(acc[k].items()):
        print(f"   {c:32s} {v / len(n[k]):18.0f}")
PY
rm -rf "$out/$tag"
