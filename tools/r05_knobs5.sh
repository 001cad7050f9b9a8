# GPU box: waves per splitter workgroup (2: 25.7 KB of LDS, fits next to a frontier workgroup; 8: the default) on the sources and on configuration 3's binaries
O=gpurun_out/r05; mkdir -p $O
KNOB_LIB=build/libzultra_amd_knobs.so timeout 600 python tools/knob_sweep.py 100000000 pysrc -- "" ZULTRA_HIP_SPLIT_WAVES=2 ZULTRA_HIP_SPLIT_WAVES=4 ZULTRA_HIP_SPLIT_WAVES=16 "" > $O/knobs_split.txt 2>&1
KNOB_BLOCK=32768 KNOB_LIB=build/libzultra_amd_knobs.so timeout 600 python tools/knob_sweep.py 51220480 binary -- "" ZULTRA_HIP_SPLIT_WAVES=2 ZULTRA_HIP_SPLIT_WAVES=4 "" >> $O/knobs_split.txt 2>&1
cat $O/knobs_split.txt
