# What of a tools/r06_final.sh run is kept: gpurun_out/r06f -> profiles/ (tracked).
O=gpurun_out/r06f
for c in 2 3 4 5; do cp $O/r06_kernel_stats_c$c.csv $O/r06_timeline_c$c.txt $O/r06_traffic_c$c.json $O/r06_sq_c$c.json profiles/ 2>/dev/null; cp $O/r06_sq_c${c}_counters.csv profiles/r06_sq_counters_c$c.csv 2>/dev/null; done
cp $O/bench_driver.json profiles/r06_bench_line.json; cp $O/bench_strong.json profiles/r06_bench_line_strong.json; cp $O/bench_c1.json profiles/r06_bench_line_config1.json
cp $O/pytest.log profiles/r06_pytest_gpu.txt; cp $O/fuzz_gpu.txt profiles/r06_fuzz_gpu.txt
ls profiles | grep r06
