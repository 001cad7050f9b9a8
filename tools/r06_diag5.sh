O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in 4 2; do
ZULTRA_HIP_LANE_TASKS=$t ZULTRA_HIP_STREAMS=1 timeout 300 python tools/lp_trace.py 33554432 pysrc > $O/lp_trace_alone_t$t.txt 2>&1
done
ZULTRA_HIP_LANE_TASKS=4 ZULTRA_HIP_LANE_WAVES=16 ZULTRA_HIP_STREAMS=1 timeout 300 python tools/lp_trace.py 33554432 pysrc > $O/lp_trace_alone_t4w16.txt 2>&1
head -12 $O/lp_trace_alone_t4.txt;  head -12 $O/lp_trace_alone_t2.txt; head -12 $O/lp_trace_alone_t4w16.txt
