set -u
cd $GRAFT_REPO_ROOT
L="python tools/large_case.py 65536 4096 64 --f32 --no-hist --randomized"
echo "== default (first orth skipped when warm)"; $L 2>&1 | grep -E " iters=|rel_err"
echo "== HOOK_ORTH_ALL=1"; TLSQ_HOOK_ORTH_ALL=1 $L 2>&1 | grep -E " iters=|rel_err"
echo "== debug"; TLSQ_DEBUG=1 $L 2>&1 | grep "hook:" | sort | uniq -c
echo "== C5 exact"; python tools/large_case.py 65536 4096 64 --f32 --no-hist 2>&1 | grep -E " iters=|rel_err"
bash tools/dbg/prof_cmd.sh r05e_c5x $PWD/tools/large_case.py 65536 4096 64 --f32 --no-hist | head -8
