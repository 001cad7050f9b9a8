# A/B of builds of the library on ONE box, every build in a process of its own (a process's later contexts may be placed worse on the hardware queues), in alternating rounds
# usage (gpurun): bash tools/r06_ab.sh <tag> <corpus> <lib> [<lib> ...]
T=$1; shift; K=$1; shift
O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}   # (enough hardware queues for every stream of a context: the placement no longer depends on what the process created before)
: > $O/ab_$T.txt
for round in 1 2 3; do for L in "$@"; do timeout 300 python tools/ab_step.py ${AB_BYTES:-100000000} $K 4 -- $L 2>&1 | grep "step ms" >> $O/ab_$T.txt; done; done
sort $O/ab_$T.txt
