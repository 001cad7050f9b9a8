O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=8
for L in head new; do
  LIB=build/libzultra_amd_head.so; [ $L = new ] && LIB=zultra_amd/libzultra_amd.so
  ZULTRA_HIP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace -d $O/kt_$L -o kt --output-format csv -- python3 tools/step_dev.py $LIB 33554432 pysrc 4 > $O/post_alone_$L.txt 2>&1
  python tools/timeline.py $(find $O/kt_$L -name "*kernel_trace.csv" | head -1) $O/timeline_post_alone_$L.txt 2>/dev/null; rm -rf $O/kt_$L
  echo "== $L: one run of 32 MiB alone"; grep "total min" $O/post_alone_$L.txt | cut -c1-70; grep "post_tasks<false>\|emit_tasks<false>" $O/timeline_post_alone_$L.txt | awk '{print $3, $4}' | tr '\n' ' '; echo
done
bash tools/r06_ab3.sh post build/libzultra_amd_head.so zultra_amd/libzultra_amd.so
: > $O/abf_post.txt
for round in 1 2; do for L in build/libzultra_amd_head.so zultra_amd/libzultra_amd.so; do timeout 300 python tools/ab_files.py $L 262144 2>&1 | grep "files/s" >> $O/abf_post.txt; done; done
sort $O/abf_post.txt
