mkdir -p gpurun_out/s18
python tools/back_bench.py 32 14 int8 "tuning=1,back_nwo=16;back_nwo=0;back_nwo=12;back_nwo=14;back_nwo=10,back_nst13=20;back_nst13=-1,back_nwo=16;back_nwo=0" > gpurun_out/s18/nwo.txt 2>&1; cat gpurun_out/s18/nwo.txt
python tools/back_bench.py 8 14 int16 "tuning=1,back_nwo=16;back_nwo=0" >> gpurun_out/s18/nwo.txt 2>&1; tail -2 gpurun_out/s18/nwo.txt
(timeout 600 python -m pytest tests -m gpu -q -x -k "one_launch or config3 or fuzz or back_half or every_code_path" > gpurun_out/s18/sel.log 2>&1; echo rc=$? >> gpurun_out/s18/sel.log); tail -3 gpurun_out/s18/sel.log
