mkdir -p gpurun_out/s16
for lib in fast-llama_amd/lib/var/libflm_prev.so fast-llama_amd/lib/libflm_gpu.so; do echo "== $lib"; for pos in 300 516 900; do FLM_GPU_LIB=$lib python tools/back_bench.py 32 $pos int8 "tuning=1"; done; done > gpurun_out/s16/scx.txt 2>&1; cat gpurun_out/s16/scx.txt
(timeout 900 python -m pytest tests -m gpu -q -x -k "long_context or split or config3 or rank_spanning or fuzz or one_launch" > gpurun_out/s16/sel.log 2>&1; echo rc=$? >> gpurun_out/s16/sel.log); tail -3 gpurun_out/s16/sel.log
