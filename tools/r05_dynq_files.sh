# GPU box: DEBUG_HIP_DYNAMIC_QUEUES=1 against files mode (configuration 5): which batch counts crash the runtime
O=gpurun_out/r05; mkdir -p $O
export DEBUG_HIP_DYNAMIC_QUEUES=1
run() { timeout 300 python3 bench.py --config 5 --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/c5d.json 2> $O/c5d.err; echo "$* rc=$?"; }
run --files 65536
run --files 131072
run --files 150000
run --files 262144
run --files 1000000 --batch 32768
ZULTRA_HIP_FILES_RUN_GRAPHS=0 run --files 1000000
which gdb; 
timeout 300 gdb -batch -ex run -ex bt --args python3 bench.py --config 5 --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --files 1000000 2>&1 | tail -40 > $O/c5d_gdb.txt; tail -40 $O/c5d_gdb.txt
