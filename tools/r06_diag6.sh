O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in 256 512 768 1024; do
ZULTRA_HIP_COOP_TASKS=100000000 ZULTRA_HIP_COOP_SMALL=$s ZULTRA_HIP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace -d $O/kt_a -o kt --output-format csv -- python3 tools/step_dev.py build/libzultra_amd_knobs.so 33554432 pysrc 4 > $O/alone_c$s.txt 2>&1
python tools/timeline.py $(find $O/kt_a -name "*kernel_trace.csv" | head -1) $O/timeline_alone_c$s.txt 2>/dev/null
rm -rf $O/kt_a
grep "total min" $O/alone_c$s.txt
done
for s in 512 1024; do
ZULTRA_HIP_COOP_TASKS=100000000 ZULTRA_HIP_COOP_SMALL=$s ZULTRA_HIP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace -d $O/kt_a -o kt --output-format csv -- python3 tools/step_dev.py build/libzultra_amd_knobs.so 8388608 pysrc 4 > $O/alone8_c$s.txt 2>&1
python tools/timeline.py $(find $O/kt_a -name "*kernel_trace.csv" | head -1) $O/timeline_alone8_c$s.txt 2>/dev/null
rm -rf $O/kt_a
grep "total min" $O/alone8_c$s.txt
done
ZULTRA_HIP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace -d $O/kt_a -o kt --output-format csv -- python3 tools/step_dev.py build/libzultra_amd_knobs.so 8388608 pysrc 4 > $O/alone8_base.txt 2>&1
python tools/timeline.py $(find $O/kt_a -name "*kernel_trace.csv" | head -1) $O/timeline_alone8_base.txt 2>/dev/null
grep "total min" $O/alone8_base.txt
grep parse_ $O/timeline_alone_c*.txt $O/timeline_alone8_*.txt
