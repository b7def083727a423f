for cfg in "1 64" "1 32" "1 16" "2 64" "2 32" "2 16" "2 8"; do
  set -- $cfg
  echo "rows=$1 ct=$2: $(TLSQ_RUS_ROWS=$1 TLSQ_RUS_CT=$2 python bench.py --cpu-iters 0 --rows ${ROWS:-20000} --steps ${STEPS:-3} 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline']['achieved']), round(d['phases_ms_per_iter']['update'],4))")"
done
