# kernel timeline of one step through a build: bash tools/r06_tl.sh <tag> <lib> <corpus> [bytes]
T=$1; L=$2; K=$3; N=${4:-100000000}
O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
timeout 300 rocprofv3 --kernel-trace -d $O/kt_$T -o kt --output-format csv -- python3 tools/step_dev.py $L $N $K 5 > $O/tl_$T.log 2>&1
python tools/timeline.py $(find $O/kt_$T -name "*kernel_trace.csv" | head -1) $O/timeline_$T.txt 2>/dev/null
rm -rf $O/kt_$T; grep "total min" $O/tl_$T.log
