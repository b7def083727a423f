export TMPDIR=/tmp
out=$PWD/gpurun_out/c5p
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $PWD/tools/large_case.py 65536 4096 64 --f32 --no-hist > $out/log.txt 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot/1e6)
for r in rows[:26]:
    print(f'{r["Name"].split("(")[0][-44:]:44s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:9.1f} us total {float(r["TotalDurationNs"])/1e6:8.2f} ms {float(r["Percentage"]):5.1f}%')
PY
