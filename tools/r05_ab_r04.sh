# GPU box: the round-4 tree (build/r04tree: git archive of the round-4 commit + its library built here) against the current one: pipelined step, three corpora.
O=gpurun_out/r05; mkdir -p $O
for C in "text:100000000" "pysrc:100000000" "mixed:268435456"; do
  K=${C%%:*}; N=${C##*:}
  for rep in 1 2; do
    (cd build/r04tree && timeout 300 python tools/ab_lib.py zultra_amd/libzultra_amd.so $N $K 2>&1 | sed 's/^/r04 /') >> $O/abr04.txt
    timeout 300 python tools/ab_lib.py zultra_amd/libzultra_amd.so $N $K 2>&1 | sed 's/^/r05 /' >> $O/abr04.txt
  done
done
grep " group=" $O/abr04.txt | awk '{print $1, $3, $10, $6, $14}'
