#!/bin/bash
# kernel timeline of the C2 solve with the returned decomposition (GPU box, repo root): bash tools/dbg/ws_prof.sh <tag>
set -u
tag=${1:-ws}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p "$out"
python3 tools/dbg/ws_profile.py 4 > "$out/${tag}_plain_run.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_trace" -- python3 $PWD/tools/dbg/ws_profile.py 3 > "$out/${tag}_run.log" 2>&1
cp "$(find "$out/${tag}_trace" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
python3 tools/timeline_full.py "$(find "$out/${tag}_trace" -name '*kernel_trace.csv' | head -1)" 2 > "$out/${tag}_timeline_full.txt" 2>&1
find "$out/${tag}_trace" -name '*kernel_trace.csv' -delete
cat "$out/${tag}_plain_run.log"
