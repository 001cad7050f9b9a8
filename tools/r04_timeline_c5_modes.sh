O=gpurun_out/r04e; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for mode in graphs direct; do
  if [ $mode = direct ]; then export ZULTRA_HIP_STREAMS=2; fi
  rocprofv3 --kernel-trace -d $O/kt5_$mode -o kt --output-format csv -- python3 bench.py --config 5 --files 262144 --profile-run --steps 2 --warmup 1 > $O/kt5_$mode.json 2> $O/kt5_$mode.err
  python tools/timeline.py $(find $O/kt5_$mode -name "*kernel_trace.csv" | head -1) $O/timeline_c5_$mode.txt all
  rm -rf $O/kt5_$mode
  echo "== $mode"; awk '$3>0.4' $O/timeline_c5_$mode.txt | head -50
done
