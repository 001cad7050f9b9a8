# GPU box: the matchfinder's CU share and the stagger point on configuration 3's binaries (32 KiB max-blocks) and on the Python sources
O=gpurun_out/r05; mkdir -p $O
S='"" ZULTRA_HIP_MF_CUS=94 ZULTRA_HIP_MF_CUS=88 ZULTRA_HIP_MF_CUS=75 ZULTRA_HIP_STAGGER=3 ZULTRA_HIP_STAGGER=4 ZULTRA_HIP_STREAMS=2 ZULTRA_HIP_STREAMS=4 ZULTRA_HIP_FIRST_RUN=60 ZULTRA_HIP_FIRST_RUN=40 ZULTRA_HIP_MF_CUS=88,ZULTRA_HIP_FIRST_RUN=60 ""'
KNOB_BLOCK=32768 KNOB_LIB=build/libzultra_amd_knobs.so timeout 600 python tools/knob_sweep.py 51220480 binary -- "" ZULTRA_HIP_MF_CUS=94 ZULTRA_HIP_MF_CUS=88 ZULTRA_HIP_MF_CUS=75 ZULTRA_HIP_STAGGER=3 ZULTRA_HIP_STAGGER=4 ZULTRA_HIP_STREAMS=2 ZULTRA_HIP_STREAMS=4 ZULTRA_HIP_FIRST_RUN=60 ZULTRA_HIP_FIRST_RUN=40 ZULTRA_HIP_MF_CUS=88,ZULTRA_HIP_FIRST_RUN=60 "" > $O/knobs_c3.txt 2>&1
KNOB_LIB=build/libzultra_amd_knobs.so timeout 600 python tools/knob_sweep.py 100000000 pysrc -- "" ZULTRA_HIP_MF_CUS=94 ZULTRA_HIP_MF_CUS=88 ZULTRA_HIP_MF_CUS=75 "" > $O/knobs_c2b.txt 2>&1
cat $O/knobs_c3.txt $O/knobs_c2b.txt
