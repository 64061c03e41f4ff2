ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/w_trace -o t -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-count-step > $ROOT/gpurun_out/w_trace.json 2> $ROOT/gpurun_out/w_trace.err
cd $ROOT
python3 - <<'P'
import csv,glob
f=glob.glob('gpurun_out/w_trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
out=open('gpurun_out/w_trace_summary.txt','w')
for r in rows:
    n=r['Kernel_Name']
    if 'ftn::' not in n: continue
    short=n.split('(')[0].replace('void ','').replace('ftn::','')
    out.write("%-45s %8.3f ms\n"%(short[:45],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
out.close()
P
rm -rf gpurun_out/w_trace
cat gpurun_out/w_trace_summary.txt | head -60
