# GPU box: A/B of library builds, every kernel alone on the chip (one run at a time) and the pipelined step. usage: bash tools/r05_ab.sh <tag> lib1.so lib2.so ...
T=$1; shift
O=gpurun_out/r05; mkdir -p $O
for L in "$@"; do
  for rep in 1 2; do
    ZULTRA_HIP_STREAMS=1 timeout 200 python tools/ab_lib.py $L 100000000 pysrc >> $O/ab_$T.txt 2>&1
    timeout 200 python tools/ab_lib.py $L 100000000 pysrc >> $O/ab_$T.txt 2>&1
  done
done
grep -v "^/opt\|amdgpu.ids" $O/ab_$T.txt | cut -c1-330
