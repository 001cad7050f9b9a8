# GPU box: pipelined step on three corpora, A/B of builds. usage: bash tools/r05_ab3.sh <tag> lib1.so lib2.so ...
T=$1; shift
O=gpurun_out/r05; mkdir -p $O
for C in "text:100000000" "pysrc:100000000" "mixed:268435456"; do
  K=${C%%:*}; N=${C##*:}
  for rep in 1 2 3; do
    for L in "$@"; do timeout 300 python tools/ab_lib.py $L $N $K >> $O/ab3_$T.txt 2>&1; done
  done
done
grep " group=" $O/ab3_$T.txt | awk '{print $1, $2, $9, $5, $13}'
