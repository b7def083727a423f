#!/bin/bash
# rocprofv3 kernel trace of C5 (65536 x 4096 fp32, rank 64) and the compressed timeline of the timed solve.
#   bash tools/dbg/c5_trace.sh <tag> [--randomized]
set -u
tag=${1:-c5t}; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag" -- python3 $PWD/tools/large_case.py 65536 4096 64 --f32 --no-hist "$@" > "$out/$tag.log" 2>&1
grep " iters=" "$out/$tag.log"
tr=$(find "$out/$tag" -name '*kernel_trace.csv' | head -1)
python3 tools/dbg/trace_runs.py "$tr" > "$out/${tag}_timeline.txt"
cp "$(find "$out/$tag" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
rm -rf "$out/$tag"
tail -3 "$out/${tag}_timeline.txt"
