O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ZULTRA_HIP_STREAMS=1 timeout 300 python tools/chain_trace.py 33554432 pysrc > $O/chain_trace_alone.txt 2>&1
ZULTRA_HIP_STREAMS=1 ZULTRA_HIP_SEG_WHOLE=4096 timeout 300 python tools/chain_trace.py 33554432 pysrc > $O/chain_trace_alone_sw4096.txt 2>&1
