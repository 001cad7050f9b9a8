O=gpurun_out/r05; mkdir -p $O
for Q in 16 24; do
for L in "$@"; do
  echo "== $L queues $Q" >> $O/inflightq.txt
  for n in 1 3; do GPU_MAX_HW_QUEUES=$Q timeout 300 python tools/two_in_flight.py 100000000 $n 4 $L >> $O/inflightq.txt 2>&1; done
done
done
grep -v amdgpu.ids $O/inflightq.txt
