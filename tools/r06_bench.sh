# the driver's command (python bench.py, default arguments) on the GPU box
T=${1:-full}
O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
( time timeout 1500 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err ) 2> $O/bench_$T.time; echo "rc=$?" >> $O/bench_$T.time
tail -4 $O/bench_$T.time
python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$T.json') if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step'],'bit_exact',d.get('bit_exact_vs_reference_full'))
print('synthetic', d.get('synthetic_text',{}).get('value'))
for k,v in d.get('other_configs',{}).items(): print(k, v.get('value'), v.get('unit'), v.get('rc'), v.get('bit_exact_vs_reference_full'), v.get('wall_s'))
p=d.get('strong_scaling_projection',{})
print(p.get('note'))
for n,v in p.get('by_ranks',{}).items(): print(n, v)
PY
