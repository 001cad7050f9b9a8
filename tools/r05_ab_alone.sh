# GPU box: A/B of library builds with every kernel alone on the chip (one run at a time), on several corpora. usage: bash tools/r05_ab_alone.sh <tag> "<corpus:bytes ...>" lib1.so lib2.so ...
T=$1; shift; CS=$1; shift
O=gpurun_out/r05; mkdir -p $O
for C in $CS; do
  K=${C%%:*}; N=${C##*:}
  for L in "$@"; do
    ZULTRA_HIP_STREAMS=1 timeout 300 python tools/ab_lib.py $L $N $K >> $O/abalone_$T.txt 2>&1
  done
done
grep " group=" $O/abalone_$T.txt | awk '{print $1, $2, $3, $11, $12, $9}'
