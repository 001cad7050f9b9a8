# Round 6 quick check on the GPU box: the GPU suite (optionally a -k subset), then the bench line of configuration 2 and a kernel timeline of its step
# usage (gpurun): bash tools/r06_quick.sh <tag> ["pytest -k expression" | none]
T=$1; K=$2
O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ "$K" != "none" ]; then
  if [ -n "$K" ]; then timeout 2400 python -m pytest tests -m gpu -x -q -k "$K" > $O/pytest_$T.txt 2>&1; else timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_$T.txt 2>&1; fi
  tail -5 $O/pytest_$T.txt
fi
timeout 600 python bench.py --no-other-configs --no-synthetic --steps 20 --warmup 3 > $O/bench_$T.json 2> $O/bench_$T.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$T -o kt --output-format csv -- python3 bench.py --config 2 --profile-run --no-synthetic --steps 2 --warmup 1 > $O/kt_$T.json 2> $O/kt_$T.err
python tools/timeline.py $(find $O/kt_$T -name "*kernel_trace.csv" | head -1) $O/timeline_$T.txt 2>/dev/null
rm -rf $O/kt_$T
python -c "
import json;d=json.load(open('$O/bench_$T.json'));print('value',d['value'],'ms',d['ms_per_step'],'bit_exact',d.get('bit_exact_vs_reference_full'))"
tail -8 $O/timeline_$T.txt
