#!/bin/bash
# kernel stats + timeline of one iteration of the C2 solve (run on the GPU box through gpurun, from the repo root)
#   bash tools/prof_c2.sh <tag> [iteration]
set -u
tag=${1:-c2}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_trace" -- python3 $PWD/tools/c2_debug.py > "$out/${tag}_run.log" 2>&1
cp "$(find "$out/${tag}_trace" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
python3 tools/timeline.py "$(find "$out/${tag}_trace" -name '*kernel_trace.csv' | head -1)" ${2:-} > "$out/${tag}_timeline.txt" 2>&1
find "$out/${tag}_trace" -name '*kernel_trace.csv' -delete
tail -3 "$out/${tag}_run.log"
cat "$out/${tag}_timeline.txt"
