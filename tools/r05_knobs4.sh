# GPU box: configuration 4's knobs (segment length, who parses the segments, demotion, runs) on 256 MiB of its stream, with the round's final kernels
O=gpurun_out/r05; mkdir -p $O
KNOB_LIB=build/libzultra_amd_knobs.so timeout 1200 python tools/knob_sweep.py 268435456 mixed -- "" ZULTRA_HIP_CUT_LEN=2048 ZULTRA_HIP_CUT_LEN=1024 ZULTRA_HIP_SEG_WIDE=256 ZULTRA_HIP_SEG_WIDE=8192 ZULTRA_HIP_DEMOTE=1 ZULTRA_HIP_DEMOTE=4 ZULTRA_HIP_STREAMS=3 ZULTRA_HIP_STREAMS=5 ZULTRA_HIP_STREAMS=6 ZULTRA_HIP_CUT_MIN=8192 ZULTRA_HIP_LANE_WAVES=8 ZULTRA_HIP_LANE_WAVES=16 ZULTRA_HIP_CUT_LEN=2048,ZULTRA_HIP_DEMOTE=1 "" > $O/knobs_c4.txt 2>&1
cat $O/knobs_c4.txt
