O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo default; timeout 300 python tools/ab_step.py 100000000 pysrc 3 -- zultra_amd/libzultra_amd.so 2>&1 | grep "step ms"
for q in 2 4 6 8 12; do echo GPU_MAX_HW_QUEUES=$q; GPU_MAX_HW_QUEUES=$q timeout 300 python tools/ab_step.py 100000000 pysrc 3 -- zultra_amd/libzultra_amd.so 2>&1 | grep "step ms"; done
echo step_dev default; timeout 300 python tools/step_dev.py zultra_amd/libzultra_amd.so 100000000 pysrc 6 | tail -1 | cut -c1-100
for q in 4 8; do echo step_dev GPU_MAX_HW_QUEUES=$q;  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/step_dev.py zultra_amd/libzultra_amd.so 100000000 pysrc 6 | tail -1 | cut -c1-100; done
