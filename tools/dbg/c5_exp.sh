set -u
cd $GRAFT_REPO_ROOT
L="python tools/large_case.py 65536 4096 64 --f32 --no-hist --randomized"
echo "== new hook default (npow 2, warm)"; $L 2>&1 | grep -E " iters=|rel_err|svp_hist"
echo "== HOOK_POWER=1"; TLSQ_HOOK_POWER=1 $L 2>&1 | grep -E " iters=|rel_err"
echo "== HOOK_POWER=3"; TLSQ_HOOK_POWER=3 $L 2>&1 | grep -E " iters=|rel_err"
echo "== HOOK_COLD=1"; TLSQ_HOOK_COLD=1 TLSQ_DEBUG=1 $L 2>&1 | grep -E " iters=|rel_err|hook:" | sort | uniq -c | head
echo "== HOOK_CLASSIC=1"; TLSQ_HOOK_CLASSIC=1 $L 2>&1 | grep -E " iters=|rel_err"
echo "== exact"; python tools/large_case.py 65536 4096 64 --f32 --no-hist 2>&1 | grep -E " iters=|rel_err"
echo "== debug hook lines"; TLSQ_DEBUG=1 $L 2>&1 | grep "hook:" | sort | uniq -c
bash tools/dbg/c5_trace.sh r05c_c5r --randomized
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "c5 or large" 2>&1 | tail -3
