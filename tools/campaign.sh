#!/bin/bash
# fuzz campaign on the GPU box (from the repo root): bash tools/campaign.sh <outdir> [quick]
set -u
out=${1:-gpurun_out/campaign}
mkdir -p "$out"
for s in 0 7 11 202 505; do python3 tools/fuzz_parity.py $s 300 > "$out/parity_$s.log" 2>&1; tail -1 "$out/parity_$s.log"; done
for s in 1 2; do python3 tools/fuzz_parity.py $s 40 --big > "$out/big_$s.log" 2>&1; tail -1 "$out/big_$s.log"; done
for s in 0 1 2 3; do python3 tools/fuzz_lrf.py $s 120 > "$out/lrf_$s.log" 2>&1; tail -1 "$out/lrf_$s.log"; done
for s in 0 4 5 7 20 21; do python3 tools/fuzz_misc.py $s 120 > "$out/misc_$s.log" 2>&1; tail -1 "$out/misc_$s.log"; done
python3 tools/fuzz_ga.py > "$out/ga.log" 2>&1; tail -1 "$out/ga.log"
