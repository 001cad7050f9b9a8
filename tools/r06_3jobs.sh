O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for sp in 1 0; do
  ZULTRA_HIP_SPREAD_STREAMS=$sp timeout 600 python bench.py --no-other-configs --no-synthetic --steps 10 --warmup 3 > $O/b3_$sp.json 2> $O/b3_$sp.err
  python -c "
import json;d=json.load(open('$O/b3_$sp.json'));print('spread $sp value',d['value'],'ms',round(d['ms_per_step'],2),'3jobs',d.get('three_jobs_in_flight',{}).get('MBps'))"
done; done
