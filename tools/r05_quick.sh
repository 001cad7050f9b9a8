# Round-5 quick measurement on the GPU box: A/B timings of one build, bench line of configuration 2, kernel timeline.
# usage (gpurun): bash tools/r05_quick.sh <tag> [lib.so]
T=$1; LIB=${2:-zultra_amd/libzultra_amd.so}
O=gpurun_out/r05; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 python tools/ab_lib.py $LIB 100000000 pysrc > $O/ab_$T.txt 2>&1
ZULTRA_HIP_STREAMS=1 timeout 200 python tools/ab_lib.py $LIB 100000000 pysrc >> $O/ab_$T.txt 2>&1
timeout 300 python bench.py --no-other-configs --no-synthetic --steps 10 --warmup 3 > $O/bench_$T.json 2> $O/bench_$T.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$T -o kt --output-format csv -- python3 bench.py --config 2 --profile-run --no-synthetic --steps 2 --warmup 1 > $O/kt_$T.json 2> $O/kt_$T.err
cp $(find $O/kt_$T -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$T.csv 2>/dev/null
python tools/timeline.py $(find $O/kt_$T -name "*kernel_trace.csv" | head -1) $O/timeline_$T.txt 2>/dev/null
rm -rf $O/kt_$T
cat $O/ab_$T.txt | tail -2; python -c "
import json;d=json.load(open('$O/bench_$T.json'));print(d['value'],d['ms_per_step'])"
