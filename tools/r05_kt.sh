# GPU box: rocprofv3 kernel trace + timeline of one configuration with the current build. usage: bash tools/r05_kt.sh <config> <tag> [bench args]
c=$1; T=$2; shift; shift
O=gpurun_out/r05; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_$T -o kt --output-format csv -- python3 bench.py --config $c --profile-run --no-synthetic --steps 2 --warmup 1 "$@" > $O/kt_$T.json 2> $O/kt_$T.err
cp $(find $O/kt_$T -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$T.csv 2>/dev/null
python tools/timeline.py $(find $O/kt_$T -name "*kernel_trace.csv" | head -1) $O/timeline_$T.txt 2>/dev/null
rm -rf $O/kt_$T
tail -c 600 $O/kt_$T.json
