mkdir -p gpurun_out/s5
(timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/s5/gputests.log 2>&1; echo rc=$? >> gpurun_out/s5/gputests.log); grep -n "passed\|failed\|^FAILED\|^E  " gpurun_out/s5/gputests.log | head -30
for w in 2 8; do GPU_MAX_HW_QUEUES=16 timeout 300 python tools/tp_onegpu.py $w 4 64 > gpurun_out/s5/tp$w.txt 2>&1; head -3 gpurun_out/s5/tp$w.txt | cut -c1-200; done
python tools/back_bench.py 32 14 int8 "tuning=1,gr_edges=0;gr_edges=1" > gpurun_out/s5/bb32.txt 2>&1; cat gpurun_out/s5/bb32.txt
python tools/back_bench.py 32 14 int16 "tuning=1,gr_edges=0;gr_edges=1" > gpurun_out/s5/bb32_i16.txt 2>&1; cat gpurun_out/s5/bb32_i16.txt
python tools/back_bench.py 32 516 int8 "tuning=1,gr_edges=0;gr_edges=1" > gpurun_out/s5/bb32_516.txt 2>&1; cat gpurun_out/s5/bb32_516.txt
