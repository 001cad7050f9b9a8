O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=8
ZULTRA_HIP_STREAMS=1 timeout 300 python tools/mf_profile.py 33554432 pysrc > $O/mf_profile_pysrc.txt 2>&1; cat $O/mf_profile_pysrc.txt
ZULTRA_HIP_STREAMS=1 timeout 300 python tools/mf_profile.py 33554432 text > $O/mf_profile_text.txt 2>&1; cat $O/mf_profile_text.txt
