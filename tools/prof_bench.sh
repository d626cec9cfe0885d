#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench.py run (7B int8, 1 GPU); summaries land in gpurun_out/prof_bench/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench; mkdir -p gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o run -- python bench.py --no-cpu-baseline --no-decode128 "$@" > gpurun_out/prof_bench/bench.json 2> gpurun_out/prof_bench/bench.err
python3 tools/ktrace_groups.py gpurun_out/prof_bench/run_kernel_trace.csv > gpurun_out/prof_bench/run_kernel_groups.txt 2>&1    # the token's launches as back-to-back groups: the timed region apart from the load-time warm-up's replays
rm -f gpurun_out/prof_bench/run_kernel_trace.csv          # tens of MB; the stats are what is kept
f=gpurun_out/prof_bench/run_kernel_stats.csv
if [ -f "$f" ]; then python3 tools/kstats.py "$f"; else echo "no stats file"; tail -5 gpurun_out/prof_bench/bench.err; fi
cat gpurun_out/prof_bench/run_kernel_groups.txt
cut -c1-200 gpurun_out/prof_bench/bench.json
