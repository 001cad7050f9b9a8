# GPU box: N batches in flight on one GPU, for several builds of the library. usage: bash tools/r05_inflight.sh lib1.so ...
O=gpurun_out/r05; mkdir -p $O
for L in "$@"; do
  echo "== $L" >> $O/inflight.txt
  for n in 2 3; do timeout 300 python tools/two_in_flight.py 100000000 $n 4 $L >> $O/inflight.txt 2>&1; done
done
grep -v amdgpu.ids $O/inflight.txt
