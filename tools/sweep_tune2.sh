for g in 2048 4096 8192 16384 32768; do
    echo "grid=$g: $(TLSQ_SWEEP_GRID=$g python bench.py --cpu-iters 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k:round(v,4) for k,v in d['phases_ms_per_iter'].items()})")"
done
