# Round 4, run A (gpurun -- 'bash tools/r04_run_a.sh'): the in-LDS zh_mf_group against round 3's (build/libzultra_amd_hbm.so, ZULTRA_HIP_MF_CAP=0)
O=gpurun_out/r04a; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
NEW=zultra_amd/libzultra_amd.so; OLD=build/libzultra_amd_hbm.so
for s in 5 2; do
ZH_MF_STOP=$s timeout 300 python tools/probes/mf_stop_probe.py 100000000 $NEW > $O/stop${s}_new.log 2>&1
ZH_MF_STOP=$s ZULTRA_HIP_MF_CAP=0 timeout 300 python tools/probes/mf_stop_probe.py 100000000 $OLD > $O/stop${s}_old.log 2>&1
done
for k in pysrc text json mixed; do
timeout 300 python tools/ab_lib.py $NEW 50000000 $k > $O/ab_${k}_new.log 2>&1
ZULTRA_HIP_MF_CAP=0 timeout 300 python tools/ab_lib.py $OLD 50000000 $k > $O/ab_${k}_old.log 2>&1
ZULTRA_HIP_STREAMS=1 timeout 300 python tools/ab_lib.py $NEW 50000000 $k > $O/ab1_${k}_new.log 2>&1
ZULTRA_HIP_STREAMS=1 ZULTRA_HIP_MF_CAP=0 timeout 300 python tools/ab_lib.py $OLD 50000000 $k > $O/ab1_${k}_old.log 2>&1
done
timeout 600 python bench.py --config 2 --steps 10 --warmup 3 --no-other-configs --no-synthetic > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --config 5 --files 262144 --steps 3 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
tail -3 $O/pytest.log; cat $O/stop*.log $O/ab*.log | grep -v "^$"; cat $O/bench_c2.json | cut -c1-600
