#!/bin/bash
# Every row of BASELINE.md §5 in one go (run on the GPU box from the repo root, ~3 GPU-minutes):
#   bash tools/bench_all.sh > gpurun_out/all_configs.txt
# C1 = smoke(), C2 = bench.py, C3 = scale_lowrankfilter, C4 on one GPU and C5 = large_case, plus the widened rows.
set -u
F='^RCCL|^HIP ver|^ROCm|^Hostname|^Librccl|amdgpu.ids'
echo "== C1 500x50 fp64 (smoke)";            python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -Ev "$F" | tail -1
echo "== C2 20000x512 fp64 (bench.py)";      python bench.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; m = d['roofline_mfma']
print(f\"{d['value']:.1f} iters/s  sweep {r['achieved']:.0f} GB/s (frac {r['frac']:.3f}, copy ceiling {r.get('measured_copy_ceiling_GBps') or 0:.0f})  Gram {m['achieved']:.1f} TF (frac {m['frac']:.3f})  cpu {d['cpu_baseline']['value']:.2f} iters/s\")"
echo "== C3 lowrankfilter N=1e7 n=256";      python tools/scale_lowrankfilter.py --no-hist 2>&1 | grep "^N="
echo "== C4 200000x512 fp64 on one GPU";     python tools/large_case.py 200000 512 16 --no-hist 2>&1 | grep " iters="
echo "== C5 65536x4096 fp32 rank 64";        python tools/large_case.py 65536 4096 64 --f32 --no-hist 2>&1 | grep " iters="
echo "== C5 with svd = randomized (BASELINE config 5's algorithm)"; python tools/large_case.py 65536 4096 64 --f32 --no-hist --randomized 2>&1 | grep " iters="
echo "== large mode 16384x8192 fp32 rank 40"; python tools/large_case.py 16384 8192 40 --f32 --no-hist 2>&1 | grep " iters="
echo "== batched rtls 50x(3+1)";             python tools/bench_batched.py 2>&1 | grep -Ev "$F" | tail -1 | cut -c1-260
echo "== batched rtls 500x(5+1)";            python tools/bench_batched.py --M 500 --n 5 --batch 4000 --cpu-problems 50 2>&1 | grep -Ev "$F" | tail -1 | cut -c1-260
echo "== batched rtls 50x(3+1) fp32";        python tools/bench_batched.py --f32 2>&1 | grep -Ev "$F" | tail -1 | cut -c1-260
echo "== noisy data 20000x512 (which solver serves the SVD steps)"; python tools/noisy_case.py 2>&1 | grep "^noise="
echo "== rpca_ga";                           python tools/bench_ga.py --cases 10x40,10x1000,10x1000000,64x1000000,512x200000,2048x100000,4096x50000 2>&1 | grep "^d="
