# GPU box: pipelined step on the synthetic word stream (100 MB), A/B of builds. usage: bash tools/r05_ab_text.sh <tag> lib1.so lib2.so ...
T=$1; shift
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2 3; do
  for L in "$@"; do
    timeout 300 python tools/ab_lib.py $L 100000000 text >> $O/abtext_$T.txt 2>&1
  done
done
grep " group=" $O/abtext_$T.txt | awk '{print $1, $2, $9, $5, $13}'
