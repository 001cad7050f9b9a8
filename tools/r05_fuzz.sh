# GPU box: the fuzz soak on the round's final build (tools/fuzz_gpu.py: product library against the compiled reference, byte for byte)
O=gpurun_out/r05; mkdir -p $O
{
echo "# python tools/fuzz_gpu.py on the MI355X box (gpurun), round-5 build $(python -c 'import zultra_amd; print(zultra_amd.csrc_digest())')"
timeout 900 python tools/fuzz_gpu.py 1500 91 3000000 2>&1 | tail -1
timeout 900 python tools/fuzz_gpu.py 300 92 20000000 2>&1 | tail -1
timeout 600 python tools/fuzz_gpu.py --files 200000 93 65536 2>&1 | tail -1
timeout 600 python tools/fuzz_gpu.py --stream 400 94 3000000 2>&1 | tail -1
timeout 1500 python tools/fuzz_gpu.py 40 95 300000000 2>&1 | tail -1
} >> $O/fuzz_gpu.txt 2>&1
cat $O/fuzz_gpu.txt
