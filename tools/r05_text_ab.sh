# GPU box: 100 MB of synthetic text and of the sources, input in HBM, through several builds. usage: bash tools/r05_text_ab.sh <tag> lib.so ...
T=$1; shift; O=gpurun_out/r05; mkdir -p $O
for rep in 1 2 3; do for L in "$@"; do for K in text pysrc; do timeout 200 python tools/step_dev.py $L 100000000 $K 8 2>/dev/null | grep total | cut -c1-75 >> $O/textab_$T.txt; done; done; done
cat $O/textab_$T.txt
