# GPU box: tools/probes/queue_probe.hip under rocprofv3, for the runtime's default and for DEBUG_HIP_DYNAMIC_QUEUES=1: Queue_Id per (round, stream)
O=gpurun_out/r05; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for V in 0 1; do
rm -rf $O/qp
DEBUG_HIP_DYNAMIC_QUEUES=$V GPU_MAX_HW_QUEUES=8 timeout 120 rocprofv3 --kernel-trace -d $O/qp -o kt --output-format csv -- build/queue_probe > $O/qp_$V.log 2>&1
python3 - $V <<'P' > $O/queue_probe_$V.txt
import csv,glob,sys
f=glob.glob('gpurun_out/r05/qp/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'tag_kernel' in r['Kernel_Name']]
tab={}
for r in rows:
    g=int(r.get('Grid_Size_X') or r.get('Grid_Size'))//64   # (threads: workgroups of 64)
    tab[(g-1)//16, (g-1)%16]=r['Queue_Id']
print('DEBUG_HIP_DYNAMIC_QUEUES=%s: Queue_Id of s0 | lane0 lane1 lane2 | side0 side1 side2, per round'%sys.argv[1])
for rd in sorted({k[0] for k in tab}):
    print('round %d: %3s | %3s %3s %3s | %3s %3s %3s'%((rd,)+tuple(tab.get((rd,i),'-') for i in (0,1,2,3,5,6,7))))
P
cat $O/queue_probe_$V.txt
done
rm -rf $O/qp
