O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=8
for L in build/libzultra_amd_ext0.so zultra_amd/libzultra_amd.so build/libzultra_amd_ext0.so zultra_amd/libzultra_amd.so; do
ZULTRA_HIP_STREAMS=1 timeout 300 python tools/step_dev.py $L 33554432 pysrc 8 | tail -1 | sed 's/tokenize.*frontier/ frontier/' | cut -c1-140
done
ZULTRA_HIP_STREAMS=1 timeout 300 python tools/mf_profile.py 33554432 pysrc 2>&1 | tail -9
