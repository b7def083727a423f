cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; tail -c 600 gpurun_out/r05_bench.json
bash tools/bench_all.sh > gpurun_out/r05_all_configs.txt 2>&1; grep -A1 "C5\|C3\|C4\|large mode" gpurun_out/r05_all_configs.txt | cut -c1-200
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
bash tools/profile_configs.sh r05 > gpurun_out/r05_profile_configs.log 2>&1; tail -20 gpurun_out/r05_profile_configs.log | cut -c1-200
bash tools/dbg/c5_trace.sh r05_c5r_tl --randomized
