O=gpurun_out/r03t; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
c=2
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_c$c -o f -- python3 bench.py --config $c --profile-run --no-synthetic --steps 1 --warmup 1 > $O/fetch_c$c.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_c$c -o w -- python3 bench.py --config $c --profile-run --no-synthetic --steps 1 --warmup 1 > $O/write_c$c.log 2>&1
python tools/pmc_traffic.py $(find $O/fetch_c$c -name "*results.db" | head -1) $(find $O/write_c$c -name "*results.db" | head -1) 268435456 $O/t.json $c | head -8
rm -rf $O/fetch_c$c $O/write_c$c
python tools/ab_lib.py zultra_amd/libzultra_amd.so 100000000 pysrc | grep -o "total=[0-9.]*"
ZULTRA_HIP_STREAMS=1 python tools/ab_lib.py zultra_amd/libzultra_amd.so 50000000 text | grep -o "parse=[0-9.]*"
