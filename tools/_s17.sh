cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
( time timeout 2400 python -m pytest tests/ -q -m gpu --durations=15 2>&1 | tail -40 ) > gpurun_out/r02/gputests.log 2>&1
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r02/smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_n1.json 2> gpurun_out/r02/bench_n1.err
bash tools/prof_bench.sh --steps 20 --warmup 5 > gpurun_out/r02/prof_bench.txt 2>&1
cp gpurun_out/prof_bench/run_kernel_stats.csv gpurun_out/r02/bench_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_bench/bench.json gpurun_out/r02/bench_under_rocprof.json 2>/dev/null
bash tools/pmc.sh > gpurun_out/r02/pmc.txt 2>&1
cp gpurun_out/pmc/pmc_fetch_write_raw.json gpurun_out/r02/pmc_fetch_write_raw.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 --pos 512 --no-cpu-baseline > gpurun_out/r02/bench_pos512.json 2> /dev/null
timeout 900 python bench.py --steps 20 --warmup 5 --quant int16 --no-cpu-baseline > gpurun_out/r02/bench_int16.json 2> /dev/null
timeout 900 python bench.py --steps 20 --warmup 5 --shape 1.3B --no-cpu-baseline > gpurun_out/r02/bench_1p3B.json 2> /dev/null
timeout 900 python bench.py --steps 20 --warmup 5 --shape 110M --no-cpu-baseline > gpurun_out/r02/bench_110M.json 2> /dev/null
timeout 900 python bench.py --config prefill512-int16 --steps 5 --warmup 2 > gpurun_out/r02/bench_prefill512_int16.json 2> /dev/null
timeout 900 python bench.py --config prefill512-int8 --steps 5 --warmup 2 > gpurun_out/r02/bench_prefill512_int8.json 2> /dev/null
FLM_PF_ONLY=1 bash tools/prof_prefill.sh 4 512 int16 > gpurun_out/r02/prof_prefill512_int16.txt 2>&1
FLM_PF_ONLY=1 bash tools/prof_prefill.sh 4 512 > gpurun_out/r02/prof_prefill512_int8.txt 2>&1
tail -25 gpurun_out/r02/gputests.log; cat gpurun_out/r02/smoke.log | tail -4; cut -c1-400 gpurun_out/r02/bench_n1.json; tail -12 gpurun_out/r02/prof_bench.txt; tail -12 gpurun_out/r02/pmc.txt
