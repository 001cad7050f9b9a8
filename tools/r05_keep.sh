# What of a tools/r05_final.sh run is kept: gpurun_out/r05f -> profiles/ (tracked).
O=gpurun_out/r05f
for c in 2 3 4 5; do cp $O/r05_kernel_stats_c$c.csv $O/r05_timeline_c$c.txt $O/r05_traffic_c$c.json $O/r05_sq_c$c.json profiles/ 2>/dev/null; cp $O/r05_sq_c${c}_counters.csv profiles/r05_sq_counters_c$c.csv 2>/dev/null; done
cp $O/bench_driver.json profiles/r05_bench_line.json; cp $O/bench_strong.json profiles/r05_bench_line_strong.json; cp $O/bench_c1.json profiles/r05_bench_line_config1.json
cp $O/pytest.log profiles/r05_pytest_gpu.txt
ls profiles | grep r05
