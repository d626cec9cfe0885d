mkdir -p gpurun_out/s17
python tools/back_bench.py 32 14 int8 "tuning=1;back_nst13=20;back_nst13=16;back_nst13=12;back_nst13=8;back_nst13=-1,back_nst13_head=16;back_nst13_head=8;back_nst13_head=-1,back_pre13=12;back_pre13=16;back_pre13=8" > gpurun_out/s17/nst.txt 2>&1; cat gpurun_out/s17/nst.txt
