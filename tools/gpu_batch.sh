O=gpurun_out/s36; mkdir -p $O
for pos in 516 900; do FLM_GPU_LIB=fast-llama_amd/lib/var/libflm_abl.so timeout 300 python tools/trace_back.py 4 $pos "" 103 > $O/trace_$pos.txt 2>&1; grep "attention (thread\|    part" $O/trace_$pos.txt | cut -c1-400; done
