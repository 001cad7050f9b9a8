# GPU box: configuration 5 (bench.py, with each build copied over the product library in turn: the box's copy of the tree is scratch) and the three stream corpora through two builds.
# usage: bash tools/r05_c5ab.sh <tag> old.so new.so
T=$1; A=$2; B=$3; O=gpurun_out/r05; mkdir -p $O build
cp $A build/_a.so; cp $B build/_b.so
for rep in 1 2; do
for L in build/_a.so build/_b.so; do
cp $L zultra_amd/libzultra_amd.so
timeout 300 python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L c5', d['value'], d['ms_per_step'], d['graph_ms_per_batch'], d['gzip_roundtrip_ok_first_files'])" >> $O/c5ab_$T.txt
done; done
cat $O/c5ab_$T.txt
bash tools/r05_ab3.sh $T build/_a.so build/_b.so
