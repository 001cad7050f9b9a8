O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=8
for L in more256 more16; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$L -o kt --output-format csv -- python3 tools/step_dev.py build/libzultra_amd_$L.so 100000000 pysrc 6 > $O/tl_$L.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('$O/kt_$L/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if '<true>' in n and 'zh_mf' not in n: print('$L', n.replace('void ','').split('(')[0], r['Calls'], 'avg_ms %.3f max %.3f'%(float(r['AverageNs'])/1e6, float(r['MaxNs'])/1e6))
PY
grep "total min" $O/tl_$L.log | cut -c1-80
rm -rf $O/kt_$L
done
bash tools/r06_ab.sh more pysrc build/libzultra_amd_more256.so build/libzultra_amd_more16.so
