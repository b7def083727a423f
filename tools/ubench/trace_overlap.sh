export TMPDIR=/tmp
out=$PWD/gpurun_out/ovl
rm -rf $out; mkdir -p $out
TLSQ_OVERLAP_CHUNKS=${CHUNKS:-8} rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $PWD/${PROG:-tools/large_case.py 200000 512 16 --no-hist} > $out/log.txt 2>&1
f=$(find $out -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find last solve: take the last 400 kernels, print a window around sweep/gram
sel=[r for r in rows if 'rebuild_update_shrink' in r['Kernel_Name'] or 'k_gram_kc' in r['Kernel_Name'] or 'slab_reduce' in r['Kernel_Name']]
sel=sel[-36:]
t0=int(sel[0]['Start_Timestamp'])
for r in sel:
    n=r['Kernel_Name']
    nm='SWEEP' if 'rebuild' in n else ('GRAM' if 'gram' in n else 'reduce')
    print(f"{nm:6s} start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  end {(int(r['End_Timestamp'])-t0)/1e3:9.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  queue {r.get('Queue_Id','?')}")
PY
