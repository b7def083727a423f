#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: kernel stats + the two PMC passes of one bench solve.
#   bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>_{stats,fetch,write,mfma}/ + <tag>_kernel_stats.csv, <tag>_pmc_hbm.csv,
#                                             <tag>_pmc_sweeps.json, <tag>_pmc_mfma.csv
set -u
tag=${1:-prof}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p "$out"
B="python3 $PWD/bench.py --steps 1 --warmup 0 --cpu-iters 0 --no-c4 --no-c5 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -- $B > "$out/${tag}_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/${tag}_fetch" -- $B > "$out/${tag}_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/${tag}_write" -- $B > "$out/${tag}_write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$out/${tag}_mfma" -- $B > "$out/${tag}_mfma.log" 2>&1
python3 tools/summarize_pmc.py "$out/${tag}_fetch" "$out/${tag}_write" "$out/${tag}_pmc_hbm.csv" "$out/${tag}_pmc_sweeps.json"
python3 tools/summarize_mfma.py "$out/${tag}_mfma" "$out/${tag}_pmc_mfma.csv"
cp "$(find "$out/${tag}_stats" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
# keep the merge small: raw traces are not needed once summarised
find "$out/${tag}_stats" "$out/${tag}_fetch" "$out/${tag}_write" "$out/${tag}_mfma" -name '*kernel_trace.csv' -delete
