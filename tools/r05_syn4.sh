# GPU box: DEBUG_HIP_DYNAMIC_QUEUES=0/1 on the legs of bench.py (first context, three contexts at once, a context behind others) and on a one-block call
O=gpurun_out/r05; mkdir -p $O
for V in 0 1 0 1; do
DEBUG_HIP_DYNAMIC_QUEUES=$V timeout 400 python3 bench.py --config 2 --no-other-configs --no-cpu-baseline --steps 10 --warmup 3 > $O/syn4_$V.json 2> $O/syn4_$V.err
python3 -c "
import json
d=json.loads([l for l in open('$O/syn4_$V.json') if l.startswith('{')][-1])
print('DYN=$V bench', d['value'], d['ms_per_step'], 'synthetic', d['synthetic_text']['ms_per_step'], d['synthetic_text']['device_pipeline_ms'], 'three', d['three_jobs_in_flight']['MBps'])
"
done
for V in 0 1; do DEBUG_HIP_DYNAMIC_QUEUES=$V timeout 300 python3 bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DYN=$V c1', d['ms_per_step'])"; done
