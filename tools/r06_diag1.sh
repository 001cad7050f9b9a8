# Round 6, first GPU call: where the parse stage's time goes — chain trace, one run alone (kernel timeline), and the chain-length / group-size knobs of a probe build
# usage (gpurun): bash tools/r06_diag1.sh
O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python tools/chain_trace.py 100000000 pysrc > $O/chain_trace_base.txt 2>&1
KNOB_LIB=build/libzultra_amd_knobs.so timeout 900 python tools/knob_sweep.py 100000000 pysrc -- "" ZULTRA_HIP_SEG_WHOLE=8192 ZULTRA_HIP_SEG_WHOLE=4096 \
  ZULTRA_HIP_SEG_WHOLE=4096,ZULTRA_HIP_CUT_LEN=2048 ZULTRA_HIP_SEG_WHOLE=4096,ZULTRA_HIP_CUT_LEN=3072 ZULTRA_HIP_LANE_TASKS=4 ZULTRA_HIP_SEG_WHOLE=4096,ZULTRA_HIP_LANE_TASKS=4 \
  ZULTRA_HIP_SEG_WHOLE=4096,ZULTRA_HIP_LANE_WAVES=16 ZULTRA_HIP_SEG_WHOLE=4096,ZULTRA_HIP_CUT_LEN=2048,ZULTRA_HIP_LANE_TASKS=4,ZULTRA_HIP_LANE_WAVES=16 "" > $O/knobs1.txt 2>&1
ZULTRA_HIP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace -d $O/kt_alone -o kt --output-format csv -- python3 tools/step_dev.py zultra_amd/libzultra_amd.so 33554432 pysrc 4 > $O/alone.txt 2>&1
python tools/timeline.py $(find $O/kt_alone -name "*kernel_trace.csv" | head -1) $O/timeline_alone_33mb.txt 2>/dev/null
rm -rf $O/kt_alone
ZULTRA_HIP_SEG_WHOLE=4096 timeout 300 rocprofv3 --kernel-trace -d $O/kt_sw -o kt --output-format csv -- python3 tools/step_dev.py zultra_amd/libzultra_amd.so 100000000 pysrc 4 > $O/sw4096.txt 2>&1
python tools/timeline.py $(find $O/kt_sw -name "*kernel_trace.csv" | head -1) $O/timeline_segwhole4096.txt 2>/dev/null
rm -rf $O/kt_sw
cat $O/knobs1.txt
