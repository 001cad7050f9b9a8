# GPU box: one step of 100 MB of synthetic text, fresh process, round-4 tree against the current one: timelines
O=gpurun_out/r05; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
timeout 300 rocprofv3 --kernel-trace -d $O/kt_a -o kt --output-format csv -- python3 tools/step_dev.py zultra_amd/libzultra_amd.so 100000000 text 5 > $O/syn5_r05.txt 2>&1
python tools/timeline.py $(find $O/kt_a -name "*kernel_trace.csv" | head -1) $O/timeline_text_r05.txt
rm -rf $O/kt_a
cd build/r04tree
timeout 300 rocprofv3 --kernel-trace -d ../../$O/kt_b -o kt --output-format csv -- python3 tools/step_dev.py zultra_amd/libzultra_amd.so 100000000 text 5 > ../../$O/syn5_r04.txt 2>&1
cd ../..
python tools/timeline.py $(find $O/kt_b -name "*kernel_trace.csv" | head -1) $O/timeline_text_r04.txt
rm -rf $O/kt_b
grep total $O/syn5_r05.txt $O/syn5_r04.txt | cut -c1-200
wc -l $O/timeline_text_r0*.txt
