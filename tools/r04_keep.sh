# Copy what is kept of a tools/r04_final.sh run (gpurun_out/r04/, scratch) into profiles/ (tracked).
O=gpurun_out/r04; P=profiles
for c in 2 3 4 5; do
  cp $O/r04_kernel_stats_c$c.csv $O/r04_timeline_c$c.txt $O/r04_traffic_c$c.json $O/r04_sq_c$c.json $P/
  cp $O/r04_sq_c${c}_counters.csv $P/r04_sq_counters_c$c.csv
done
cp $O/bench_driver.json $P/r04_bench_line.json
cp $O/bench_strong.json $P/r04_bench_line_strong.json
cp $O/bench_c1.json $P/r04_bench_line_config1.json
tail -2 $O/pytest.log > $P/r04_pytest_gpu.txt
