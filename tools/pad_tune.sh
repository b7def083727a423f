for pad in 8 6 4 3; do
  echo "pad=$pad: $(TLSQ_PAD=$pad python bench.py --cpu-iters 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['phases_ms_per_iter']['eig'],4), d['svd_step'], d['config']['iters_per_solve'])")"
  echo "   fuzz: $(TLSQ_PAD=$pad timeout 200 python -u tools/fuzz_parity.py 13 150 2>&1 | tail -1)"
done
