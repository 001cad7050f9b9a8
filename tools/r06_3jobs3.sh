O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for sp in "1 0" "1 1" "0 0"; do
  set -- $sp
  ZULTRA_HIP_SPREAD_STREAMS=$1 ZULTRA_HIP_SPREAD_SCOPE=$2 timeout 600 python bench.py --no-other-configs --no-synthetic --steps 10 --warmup 3 > $O/b3.json 2> $O/b3.err
  python -c "
import json;d=json.load(open('$O/b3.json'));print('spread $1 scope $2 value',d['value'],'ms',round(d['ms_per_step'],2),'3jobs',d.get('three_jobs_in_flight',{}).get('MBps'))"
done; done
unset GPU_MAX_HW_QUEUES
for sc in 0 1; do echo "torch first, 4 queues, scope $sc"; ZULTRA_HIP_SPREAD_SCOPE=$sc ZULTRA_HIP_SPREAD_STREAMS=2 timeout 300 python tools/ab_step.py 100000000 pysrc 3 -- zultra_amd/libzultra_amd.so 2>&1 | grep "step ms\|replaced"; done
