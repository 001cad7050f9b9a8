# A/B of the hand-tracked walk read-ahead: parity first (the emulator cannot see a wrong wait count), then times
O=gpurun_out/r06; mkdir -p $O; export GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not hardware_queues" 2>&1 | tail -3 > $O/async_pytest.txt; cat $O/async_pytest.txt
bash tools/r06_post.sh
