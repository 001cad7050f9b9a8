O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
unset GPU_MAX_HW_QUEUES
for q in unset 4 8; do for sp in 0 2; do
  if [ $q = unset ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo "queues=$q spread=$sp"; ZULTRA_HIP_SPREAD_STREAMS=$sp timeout 300 python tools/ab_step.py 100000000 pysrc 3 -- zultra_amd/libzultra_amd.so 2>&1 | grep "step ms\|replaced"
done; done
