cd $GRAFT_REPO_ROOT
for f in 2 4 8; do echo "== fold $f"; TLSQ_GRAM_H3_FOLD=$f python tools/dbg/gram_bench.py 2>&1 | grep "fp16"; done
echo "== C5 exact";  python tools/large_case.py 65536 4096 64 --f32 --no-hist 2>&1 | grep -E " iters=|rel_err"
echo "== C5 randomized";  python tools/large_case.py 65536 4096 64 --f32 --no-hist --randomized 2>&1 | grep -E " iters=|rel_err"
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_gram16.py -x -q -k "c5 or large or gram" 2>&1 | grep -E "passed|failed|Error|assert|error" | head
