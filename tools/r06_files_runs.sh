# files mode: runs per batch of 65 536 inputs (graph per run), a process per setting
O=gpurun_out/r06; mkdir -p $O; export GPU_MAX_HW_QUEUES=8
: > $O/files_runs.txt
for round in 1 2; do for S in 0 2 3 4; do
  echo -n "ZULTRA_HIP_STREAMS=$S  " >> $O/files_runs.txt
  ZULTRA_HIP_STREAMS=$S timeout 300 python tools/ab_files.py zultra_amd/libzultra_amd.so 262144 2>&1 | grep "files/s" >> $O/files_runs.txt
done; done
sort $O/files_runs.txt
