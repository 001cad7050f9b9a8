# A/B on one box over three corpora: bash tools/r06_ab3.sh <tag> <libA> <libB>
T=$1; shift
O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=8
: > $O/ab3_$T.txt
for K in pysrc mixed text; do for round in 1 2; do for L in "$@"; do
  N=100000000; [ $K = mixed ] && N=268435456
  timeout 600 python tools/ab_step.py $N $K 3 -- $L 2>&1 | grep "step ms" | sed "s/^/$K /" >> $O/ab3_$T.txt
done; done; done
sort $O/ab3_$T.txt
