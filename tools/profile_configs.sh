#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: rocprofv3 kernel stats of one solve of C3, C4 (one GPU) and C5, and the
# HBM traffic of their kernels (PMC FETCH_SIZE / WRITE_SIZE in separate passes, FETCH doubled on gfx950).
#   bash tools/profile_configs.sh <tag>   -> gpurun_out/<tag>_{c3,c4,c5,c5r}_kernel_stats.csv (c5r: svd = randomized), <tag>_{c3,c4,c5}_pmc_hbm.csv
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
pmc() {   # name, program args...
  local name=$1; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/${tag}_${name}_fetch" -- python3 "$@" > "$out/${tag}_${name}_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/${tag}_${name}_write" -- python3 "$@" > "$out/${tag}_${name}_write.log" 2>&1
  python3 tools/summarize_pmc.py "$out/${tag}_${name}_fetch" "$out/${tag}_${name}_write" "$out/${tag}_${name}_pmc_hbm.csv" | head -6
  rm -rf "$out/${tag}_${name}_fetch" "$out/${tag}_${name}_write"
}
run c5 $PWD/tools/large_case.py 65536 4096 64 --f32 --no-hist
run c5r $PWD/tools/large_case.py 65536 4096 64 --f32 --no-hist --randomized
run c4 $PWD/tools/large_case.py 200000 512 16 --no-hist
run c3 $PWD/tools/scale_lowrankfilter.py --no-hist
pmc c5 $PWD/tools/large_case.py 65536 4096 64 --f32 --no-hist
pmc c5r $PWD/tools/large_case.py 65536 4096 64 --f32 --no-hist --randomized
pmc c4 $PWD/tools/large_case.py 200000 512 16 --no-hist
pmc c3 $PWD/tools/scale_lowrankfilter.py --no-hist
