O=gpurun_out/s42; mkdir -p $O
(GPU_MAX_HW_QUEUES=16 timeout 900 python tools/tp_soak.py 4 900 1 2>&1 | grep -v Warning | tail -14) > $O/r06_tp_soak.txt; cat $O/r06_tp_soak.txt | cut -c1-200
(GPU_MAX_HW_QUEUES=16 timeout 600 python tools/fuzz_tp.py 40 11 2>&1 | grep -v Warning | tail -6) > $O/r06_fuzz_tp.txt; cat $O/r06_fuzz_tp.txt | cut -c1-200
