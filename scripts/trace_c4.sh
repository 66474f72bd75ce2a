cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/tr_c4
rocprofv3 --kernel-trace -d $R/gpurun_out/tr_c4 -o t --output-format csv -- python $R/scripts/perf_probe.py c4 1 10000000 > $R/gpurun_out/tr_c4.txt 2>&1
f=$(find $R/gpurun_out/tr_c4 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'sb::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last full encode call: find last k_enc_layout; print the 60 kernels before it
idx=[i for i,r in enumerate(rows) if 'k_enc_layout' in r['Kernel_Name']]
# choose a call in the timed loop (not profile pass): the third from last
i1=idx[-3]; i0=idx[-4]
t0=int(rows[i0+1]['Start_Timestamp'])
for r in rows[i0+1:i1+3]:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('sb::','')
    print('%-40s q=%s start %8.1f us dur %7.1f us' % (n[:40], r.get('Queue_Id','?'), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
