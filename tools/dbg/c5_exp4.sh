cd $GRAFT_REPO_ROOT
for sw in "" "TLSQ_COLD_CGS2=1"; do
echo "== C5 exact $sw"; env $sw python tools/large_case.py 65536 4096 64 --f32 --no-hist 2>&1 | grep -E " iters=|rel_err|svp_hist"
echo "== C5 randomized $sw"; env $sw python tools/large_case.py 65536 4096 64 --f32 --no-hist --randomized 2>&1 | grep -E " iters=|rel_err"
done
echo "== rank 250"; python tools/large_case.py 6016 2304 250 --no-hist 2>&1 | grep -E " iters=|rel_err"
echo "== 16384x8192"; python tools/large_case.py 16384 8192 40 --f32 --no-hist 2>&1 | grep -E " iters=|rel_err"
