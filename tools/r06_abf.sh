# A/B of builds on one box: files mode (tools/ab_files.py) and the synthetic text without chains (tools/ab_step.py ... text), every build in a process of its own, alternating
# usage (gpurun): bash tools/r06_abf.sh <tag> <lib> [<lib> ...]
T=$1; shift
O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
: > $O/abf_$T.txt; : > $O/abt_$T.txt
for round in 1 2 3; do for L in "$@"; do
  timeout 300 python tools/ab_files.py $L 262144 2>&1 | grep "files/s" >> $O/abf_$T.txt
  timeout 300 python tools/ab_step.py 100000000 text 4 -- $L 2>&1 | grep "step ms" >> $O/abt_$T.txt
done; done
sort $O/abf_$T.txt; sort $O/abt_$T.txt
