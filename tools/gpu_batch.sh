mkdir -p gpurun_out/s20
python tools/back_bench.py 32 14 int8 "tuning=1,back_res2=0;back_res2=1;back_res2=0;back_res2=1" > gpurun_out/s20/res2.txt 2>&1; cat gpurun_out/s20/res2.txt
python tools/back_bench.py 32 516 int8 "tuning=1,back_res2=0;back_res2=1" >> gpurun_out/s20/res2.txt 2>&1; tail -2 gpurun_out/s20/res2.txt
(timeout 600 python -m pytest tests -m gpu -q -x -k "one_launch or config3 or fuzz or back_half or every_code_path or long_context" > gpurun_out/s20/sel.log 2>&1; echo rc=$? >> gpurun_out/s20/sel.log); tail -3 gpurun_out/s20/sel.log
