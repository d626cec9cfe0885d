#!/bin/bash
# usage: tools/prof.sh <layers> <pos> [extra kbench args]  -> rocprofv3 per-kernel stats of tools/kbench.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_k; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k -o run -- python tools/kbench.py "$@" > gpurun_out/prof_k.log 2>&1
f=gpurun_out/prof_k/run_kernel_stats.csv
if [ -f "$f" ]; then python3 tools/kstats.py "$f"; else echo "no stats file"; tail -5 gpurun_out/prof_k.log; fi
