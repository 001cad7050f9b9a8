# GPU box: why is the synthetic-text leg of bench.py slower than the same step in a fresh process? timelines of both
O=gpurun_out/r05; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 python tools/step_dev.py zultra_amd/libzultra_amd.so 100000000 text 10 > $O/syn_fresh.txt 2>&1
(cd build/r04tree && timeout 200 python tools/step_dev.py zultra_amd/libzultra_amd.so 100000000 text 10) > $O/syn_fresh_r04.txt 2>&1
timeout 300 rocprofv3 --kernel-trace -d $O/kt_syn -o kt --output-format csv -- python3 tools/step_dev.py zultra_amd/libzultra_amd.so 100000000 text 4 > $O/syn_kt.txt 2>&1
python tools/timeline.py $(find $O/kt_syn -name "*kernel_trace.csv" | head -1) $O/timeline_syn_fresh.txt 2>/dev/null
rm -rf $O/kt_syn
timeout 400 rocprofv3 --kernel-trace -d $O/kt_synb -o kt --output-format csv -- python3 bench.py --config 2 --no-other-configs --no-cpu-baseline --steps 2 --warmup 1 > $O/syn_bench.json 2> $O/syn_bench.err
python tools/timeline.py $(find $O/kt_synb -name "*kernel_trace.csv" | head -1) $O/timeline_syn_bench.txt 2>/dev/null
rm -rf $O/kt_synb
cat $O/syn_fresh.txt $O/syn_fresh_r04.txt; wc -l $O/timeline_syn_*.txt
