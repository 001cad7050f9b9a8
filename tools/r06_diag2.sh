# Round 6: trace of zh_parse_lanes' groups — one run alone, then the pipelined step
O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ZULTRA_HIP_STREAMS=1 timeout 300 python tools/lp_trace.py 33554432 pysrc > $O/lp_trace_alone.txt 2>&1
timeout 300 python tools/lp_trace.py 100000000 pysrc > $O/lp_trace_step.txt 2>&1
ZULTRA_HIP_STREAMS=1 timeout 300 python tools/lp_profile.py 33554432 pysrc > $O/lp_profile_alone.txt 2>&1
cat $O/lp_trace_alone.txt $O/lp_profile_alone.txt
