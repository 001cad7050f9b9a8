O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "outgrows or chains_turn_up or overflow or no_host or scan_equals or queues" > $O/pytest_nomore.txt 2>&1; tail -3 $O/pytest_nomore.txt
bash tools/r06_ab3.sh nomore build/libzultra_amd_head.so zultra_amd/libzultra_amd.so
