O=gpurun_out/s37; mkdir -p $O
bash tools/prof_bench.sh --steps 20 --warmup 5 > $O/prof_bench.txt 2>&1; cp gpurun_out/prof_bench/run_kernel_stats.csv $O/r06_bench_kernel_stats.csv; cp gpurun_out/prof_bench/bench.json $O/r06_bench_under_rocprof.json; cp gpurun_out/prof_bench/run_kernel_groups.txt $O/r06_bench_kernel_groups.txt; cat $O/r06_bench_kernel_groups.txt
bash tools/pmc_bench.sh > $O/pmc_bench.txt 2>&1; cp gpurun_out/pmc/pmc_fetch_write_raw.json $O/r06_pmc_fetch_write_raw.json; grep k_layers $O/pmc_bench.txt
