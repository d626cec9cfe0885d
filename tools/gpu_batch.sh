mkdir -p gpurun_out/s4
python tools/back_bench.py 8 14 int8 "tuning=1,gr_edges=0;gr_edges=1;gr_edges=0;gr_edges=1" > gpurun_out/s4/bb8.txt 2>&1; cat gpurun_out/s4/bb8.txt
python tools/back_bench.py 32 14 int8 "tuning=1,gr_edges=0;gr_edges=1;gr_edges=0;gr_edges=1" > gpurun_out/s4/bb32.txt 2>&1; cat gpurun_out/s4/bb32.txt
python tools/alloc_diag.py > gpurun_out/s4/alloc_diag.txt 2>&1; grep -v " 0$" gpurun_out/s4/alloc_diag.txt | tail -8
(timeout 900 python -m pytest tests -m gpu -q -x -k "one_launch or nothing_is_allocated or tensor_parallel_p2p_is_bit or rank_spanning or batched_prompt" > gpurun_out/s4/gputests_sel.log 2>&1; echo rc=$? >> gpurun_out/s4/gputests_sel.log); tail -5 gpurun_out/s4/gputests_sel.log
