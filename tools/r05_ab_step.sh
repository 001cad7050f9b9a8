# GPU box: pipelined-step A/B of library builds (the default three runs), several repetitions interleaved. usage: bash tools/r05_ab_step.sh <tag> lib1.so lib2.so ...
T=$1; shift
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2 3; do
  for L in "$@"; do
    timeout 200 python tools/ab_lib.py $L 100000000 pysrc >> $O/abstep_$T.txt 2>&1
  done
done
grep "pysrc" $O/abstep_$T.txt | awk '{print $1, $9, $5, $6}' 
