#!/bin/bash
out=gpurun_out/r06d_campaign
mkdir -p $out
for s in 1 2 3 4 99 1234; do python3 tools/fuzz_parity.py $s 300 > "$out/parity_$s.log" 2>&1; tail -1 "$out/parity_$s.log"; done
for s in 3 4; do python3 tools/fuzz_parity.py $s 40 --big > "$out/big_$s.log" 2>&1; tail -1 "$out/big_$s.log"; done
for s in 4 5 6; do python3 tools/fuzz_lrf.py $s 120 > "$out/lrf_$s.log" 2>&1; tail -1 "$out/lrf_$s.log"; done
for s in 1 2 3 8 9; do python3 tools/fuzz_misc.py $s 120 > "$out/misc_$s.log" 2>&1; tail -1 "$out/misc_$s.log"; done
