# Round-3 measurement run on the GPU box (gpurun -- 'bash tools/final_run.sh [quick]'): tests, bench lines, rocprofv3 kernel trace, PMC traffic per
# configuration, SQ counters. Everything lands in gpurun_out/r03/; what is kept goes to profiles/ by hand.
O=gpurun_out/r03; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ "$1" != "quick" ]; then
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "rc=$?" >> $O/bench_driver.err
python bench.py --config 1 > $O/bench_c1.json 2> $O/bench_c1.err
fi
prof() {  # config, extra bench args
  c=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/kt_c$c -o kt --output-format csv -- python3 bench.py --config $c --profile-run --no-synthetic --steps 2 --warmup 1 "$@" > $O/kt_c$c.json 2> $O/kt_c$c.err
  cp $(find $O/kt_c$c -name "*kernel_stats.csv" | head -1) $O/r03_kernel_stats_c$c.csv 2>/dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_c$c -o f -- python3 bench.py --config $c --profile-run --no-synthetic --steps 1 --warmup 1 "$@" > $O/fetch_c$c.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_c$c -o w -- python3 bench.py --config $c --profile-run --no-synthetic --steps 1 --warmup 1 "$@" > $O/write_c$c.log 2>&1
  python tools/pmc_traffic.py $(find $O/fetch_c$c -name "*results.db" | head -1) $(find $O/write_c$c -name "*results.db" | head -1) 268435456 $O/r03_traffic_c$c.json $c "python3 bench.py --config $c --profile-run --no-synthetic --steps 1 --warmup 1 $*" > $O/traffic_c$c.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/sq_c$c -o s -- python3 bench.py --config $c --profile-run --no-synthetic --steps 1 --warmup 1 "$@" > $O/sq_c$c.log 2>&1
  python tools/rocpd_summary.py $(find $O/sq_c$c -name "*results.db" | head -1) $O/r03_sq_c$c > $O/sq_summary_c$c.log 2>&1
  rm -rf $O/fetch_c$c $O/write_c$c $O/sq_c$c $O/kt_c$c
}
prof 2
prof 3
prof 5 --files 262144
prof 4
for k in json pysrc mixed; do python tools/profile_encode.py 50000000 $k > $O/pe_$k.log 2>&1; done
ls -la $O | head -60
