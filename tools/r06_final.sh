# Round-6 measurement run on the GPU box (gpurun -- 'bash tools/r06_final.sh [quick]'): tests, smoke, bench lines, rocprofv3 kernel trace, PMC traffic per
# configuration, SQ counters, the fuzz soak against the compiled reference. Everything lands in gpurun_out/r06f/; what is kept goes to profiles/ (tools/r06_keep.sh).
O=gpurun_out/r06f; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ "$1" != "quick" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
fi
prof() {  # config, extra bench args
  c=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_c$c -o kt --output-format csv -- python3 bench.py --config $c --profile-run --no-synthetic --steps 2 --warmup 1 "$@" > $O/kt_c$c.json 2> $O/kt_c$c.err
  cp $(find $O/kt_c$c -name "*kernel_stats.csv" | head -1) $O/r06_kernel_stats_c$c.csv 2>/dev/null
  python tools/timeline.py $(find $O/kt_c$c -name "*kernel_trace.csv" | head -1) $O/r06_timeline_c$c.txt 2>/dev/null
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_c$c -o f -- python3 bench.py --config $c --profile-run --no-synthetic --steps 1 --warmup 1 "$@" > $O/fetch_c$c.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_c$c -o w -- python3 bench.py --config $c --profile-run --no-synthetic --steps 1 --warmup 1 "$@" > $O/write_c$c.log 2>&1
  python tools/pmc_traffic.py $(find $O/fetch_c$c -name "*results.db" | head -1) $(find $O/write_c$c -name "*results.db" | head -1) 268435456 $O/r06_traffic_c$c.json $c "python3 bench.py --config $c --profile-run --no-synthetic --steps 1 --warmup 1 $*" > $O/traffic_c$c.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/sq_c$c -o s -- python3 bench.py --config $c --profile-run --no-synthetic --steps 1 --warmup 1 "$@" > $O/sq_c$c.json 2> $O/sq_c$c.err
  python tools/rocpd_summary.py $(find $O/sq_c$c -name "*results.db" | head -1) $O/r06_sq_c$c > $O/sq_summary_c$c.log 2>&1
  # (the step of the UNPROFILED kernel-trace run of the same command: the counter pass itself slows the kernels)
  MS=$(python -c "import json;print(json.loads([l for l in open('$O/kt_c$c.json') if l.startswith('{')][-1])['ms_per_step'])" 2>/dev/null || echo 0)
  python tools/sq_profile.py $(find $O/sq_c$c -name "*results.db" | head -1) $O/r06_sq_c$c.json $c 2 "python3 bench.py --config $c --profile-run --no-synthetic --steps 1 --warmup 1 $*" $MS > $O/sq_profile_c$c.log 2>&1
  rm -rf $O/fetch_c$c $O/write_c$c $O/sq_c$c $O/kt_c$c
}
prof 2
prof 3
prof 5 --files 262144
prof 4
# the traffic / issue files have to be in profiles/ for the bench line to carry them
mkdir -p profiles; cp $O/r06_traffic_c*.json $O/r06_sq_c*.json profiles/ 2>/dev/null
if [ "$1" != "quick" ]; then
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "rc=$?" >> $O/bench_driver.err
timeout 300 python bench.py --gpus 1 --scaling strong --steps 10 --warmup 3 --no-other-configs --no-synthetic > $O/bench_strong.json 2> $O/bench_strong.err
timeout 200 python bench.py --config 1 --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err
{
echo "# python tools/fuzz_gpu.py on the MI355X box (gpurun), round-6 build $(python -c 'import zultra_amd; print(zultra_amd.csrc_digest())')"
timeout 900 python tools/fuzz_gpu.py 1500 91 3000000 2>&1 | tail -1
timeout 900 python tools/fuzz_gpu.py 300 92 20000000 2>&1 | tail -1
timeout 600 python tools/fuzz_gpu.py --files 200000 93 65536 2>&1 | tail -1
timeout 600 python tools/fuzz_gpu.py --stream 400 94 3000000 2>&1 | tail -1
timeout 1500 python tools/fuzz_gpu.py 40 95 300000000 2>&1 | tail -1
} > $O/fuzz_gpu.txt 2>&1
fi
ls -la $O | head -60
