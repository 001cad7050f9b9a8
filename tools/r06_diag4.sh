O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in 8 4 2; do
ZULTRA_HIP_LANE_TASKS=$t ZULTRA_HIP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace -d $O/kt_a -o kt --output-format csv -- python3 tools/step_dev.py build/libzultra_amd_knobs.so 33554432 pysrc 4 > $O/alone_t$t.txt 2>&1
python tools/timeline.py $(find $O/kt_a -name "*kernel_trace.csv" | head -1) $O/timeline_alone_t$t.txt 2>/dev/null
rm -rf $O/kt_a
done
KNOB_LIB=build/libzultra_amd_knobs.so timeout 900 python tools/knob_sweep.py 100000000 pysrc -- "" ZULTRA_HIP_LANE_TASKS_LAST=4 ZULTRA_HIP_LANE_TASKS_LAST=2 ZULTRA_HIP_LANE_TASKS_LAST=4,ZULTRA_HIP_LANE_WAVES=16 ZULTRA_HIP_LANE_TASKS_LAST=2,ZULTRA_HIP_LANE_WAVES=16 \
  ZULTRA_HIP_LANE_TASKS_LAST=4,ZULTRA_HIP_LAST_RUN=60 ZULTRA_HIP_LANE_TASKS_LAST=2,ZULTRA_HIP_LAST_RUN=60 "" > $O/knobs2.txt 2>&1
cat $O/knobs2.txt
grep parse_ $O/timeline_alone_t*.txt
