# GPU box: pipelined step on the mixed stream of configuration 4 (256 MiB), A/B of builds. usage: bash tools/r05_ab_mixed.sh <tag> lib1.so lib2.so ...
T=$1; shift
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do
  for L in "$@"; do
    timeout 300 python tools/ab_lib.py $L 268435456 mixed >> $O/abmixed_$T.txt 2>&1
  done
done
grep " group=" $O/abmixed_$T.txt | awk '{print $1, $2, $9, $5, $13}'
