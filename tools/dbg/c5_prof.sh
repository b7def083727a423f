#!/bin/bash
# kernel timeline of the C5 solve (65536 x 4096 fp32 rank 64), randomized or exact:  bash tools/dbg/c5_prof.sh <tag> [--randomized]
set -u
tag=${1:-c5}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_trace" -- python3 $PWD/tools/large_case.py 65536 4096 64 --f32 --no-hist ${2:-} > "$out/${tag}_run.log" 2>&1
cp "$(find "$out/${tag}_trace" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
python3 tools/timeline_solve.py "$(find "$out/${tag}_trace" -name '*kernel_trace.csv' | head -1)" 1 > "$out/${tag}_timeline.txt" 2>&1
find "$out/${tag}_trace" -name '*kernel_trace.csv' -delete
grep "iters=" "$out/${tag}_run.log"
