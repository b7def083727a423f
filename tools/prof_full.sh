#!/bin/bash
# whole-solve kernel timeline of the C2 solve (GPU box, from the repo root):  bash tools/prof_full.sh <tag>
set -u
tag=${1:-c2full}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_trace" -- python3 $PWD/tools/c2_debug.py > "$out/${tag}_run.log" 2>&1
cp "$(find "$out/${tag}_trace" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
python3 tools/timeline_full.py "$(find "$out/${tag}_trace" -name '*kernel_trace.csv' | head -1)" 2 > "$out/${tag}_timeline_full.txt" 2>&1
find "$out/${tag}_trace" -name '*kernel_trace.csv' -delete
tail -4 "$out/${tag}_run.log"
tail -40 "$out/${tag}_timeline_full.txt"
