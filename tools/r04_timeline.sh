# kernel timeline of one step: gpurun -- 'bash tools/r04_timeline.sh <config> [extra bench args]'
C=${1:-2}; shift
O=gpurun_out/r04e; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/kt_c$C -o kt --output-format csv -- python3 bench.py --config $C --profile-run --no-synthetic --steps 2 --warmup 1 "$@" > $O/kt_c$C.json 2> $O/kt_c$C.err
python tools/timeline.py $(find $O/kt_c$C -name "*kernel_trace.csv" | head -1) $O/timeline_c$C.txt
cp $(find $O/kt_c$C -name "*kernel_stats.csv" | head -1) $O/kernel_stats_c$C.csv
rm -rf $O/kt_c$C
tail -3 $O/timeline_c$C.txt
