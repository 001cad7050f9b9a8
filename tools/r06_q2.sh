O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for q in default 8 12 16 24; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 600 python bench.py --no-other-configs --no-synthetic --steps 20 --warmup 3 > $O/bench_q$q.json 2> $O/bench_q$q.err
  python -c "
import json;d=json.load(open('$O/bench_q$q.json'));print('queues $q value',d['value'],'ms',d['ms_per_step'],'3jobs',d.get('three_jobs_in_flight'))"
done
