# kernel timeline of files mode through a build: bash tools/r06_tlf.sh <tag> <lib>
T=$1; L=$2
O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$T -o kt --output-format csv -- python3 tools/ab_files.py $L 262144 > $O/tlf_$T.log 2>&1
python tools/timeline.py $(find $O/kt_$T -name "*kernel_trace.csv" | head -1) $O/timeline_$T.txt 2>/dev/null
cp $(find $O/kt_$T -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$T.csv
rm -rf $O/kt_$T; grep "files/s" $O/tlf_$T.log
