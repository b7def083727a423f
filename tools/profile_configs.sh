#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: rocprofv3 kernel stats of one solve of C3, C4 (one GPU) and C5.
#   bash tools/profile_configs.sh <tag>   -> gpurun_out/<tag>_{c3,c4,c5}_kernel_stats.csv
set -u
tag=${1:-prof}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p "$out"
run() {   # name, program args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_${name}" -- python3 "$@" > "$out/${tag}_${name}.log" 2>&1
  cp "$(find "$out/${tag}_${name}" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_${name}_kernel_stats.csv"
  find "$out/${tag}_${name}" -name '*kernel_trace.csv' -delete
  grep -E " iters=|^N=" "$out/${tag}_${name}.log"
}
run c5 $PWD/tools/large_case.py 65536 4096 64 --f32 --no-hist
run c4 $PWD/tools/large_case.py 200000 512 16 --no-hist
run c3 $PWD/tools/scale_lowrankfilter.py --no-hist
