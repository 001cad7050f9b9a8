# GPU box: configuration 1 (one 39 KB max-block through zultra_memory_compress) a few times, the GPU tests, and a fuzz run (most of its cases are batches of a few max-blocks)
O=gpurun_out/r05; mkdir -p $O
for i in 1 2 3; do timeout 200 python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c1', d['ms_per_step'], d['known_answer_ok'], d['device_ms'])"; done
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_chunk.log 2>&1; tail -2 $O/pytest_chunk.log
timeout 600 python tools/fuzz_gpu.py 800 81 1500000 2>&1 | tail -1
timeout 300 python tools/fuzz_gpu.py --stream 200 82 2000000 2>&1 | tail -1
