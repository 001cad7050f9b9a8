mkdir -p gpurun_out/r4p; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r4p/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r4p/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4p/smoke.log 2>&1
for c in 2 3 4 5 1; do python bench.py --config $c > gpurun_out/r4p/bench_c$c.json 2> gpurun_out/r4p/bench_c$c.err; echo "rc=$?" >> gpurun_out/r4p/bench_c$c.err; done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4p/bench_driver.json 2> gpurun_out/r4p/bench_driver.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r4p/kt -o kt --output-format csv -- python3 bench.py --config 2 --profile-run --no-synthetic --steps 2 --warmup 1 > gpurun_out/r4p/kt_bench.json 2> gpurun_out/r4p/kt_bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r4p/fetch -o f -- python3 bench.py --config 2 --profile-run --no-synthetic --steps 1 --warmup 1 > gpurun_out/r4p/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r4p/write -o w -- python3 bench.py --config 2 --profile-run --no-synthetic --steps 1 --warmup 1 > gpurun_out/r4p/write.log 2>&1
for k in json pysrc mixed; do python tools/profile_encode.py 50000000 $k > gpurun_out/r4p/pe_$k.log 2>&1; done
ls -R gpurun_out/r4p | head -50
