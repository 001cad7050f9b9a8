T=$1
O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=8
timeout 1200 python -m pytest tests -m gpu -x -q -k "files or edge or config1 or bench" > $O/pytest_$T.txt 2>&1; tail -3 $O/pytest_$T.txt
: > $O/abf_$T.txt
for round in 1 2 3; do for L in build/libzultra_amd_r05.so zultra_amd/libzultra_amd.so; do timeout 300 python tools/ab_files.py $L 262144 2>&1 | grep "files/s" >> $O/abf_$T.txt; done; done
sort $O/abf_$T.txt
