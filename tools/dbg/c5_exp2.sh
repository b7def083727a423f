set -u
cd $GRAFT_REPO_ROOT
for sw in "" "TLSQ_NO_WIDE_SWEEP=1"; do
  echo "== 32768x2304 r=40 f32 exact $sw"; env $sw python tools/large_case.py 32768 2304 40 --f32 2>&1 | grep -E " iters=|rel_err|svp_hist"
  echo "== 32768x2304 r=40 f32 randomized $sw"; env $sw python tools/large_case.py 32768 2304 40 --f32 --randomized 2>&1 | grep -E " iters=|rel_err"
done
echo "== C5 randomized"; python tools/large_case.py 65536 4096 64 --f32 --no-hist --randomized 2>&1 | grep -E " iters=|rel_err"
echo "== C5 exact"; python tools/large_case.py 65536 4096 64 --f32 --no-hist 2>&1 | grep -E " iters=|rel_err"
bash tools/dbg/c5_trace.sh r05d_c5r --randomized
