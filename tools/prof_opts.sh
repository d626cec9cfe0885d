#!/bin/bash
# usage: tools/prof_opts.sh <tag> <layers> <pos> <tokens> "<k=v,...>"  -> rocprofv3 per-kernel stats of tools/decode_opts.py under that option set
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf gpurun_out/prof_$tag; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o run -- python tools/decode_opts.py "$@" > gpurun_out/prof_$tag.log 2>&1
tail -1 gpurun_out/prof_$tag.log
f=gpurun_out/prof_$tag/run_kernel_stats.csv
if [ -f "$f" ]; then python3 tools/kstats.py "$f" | head -12; else echo "no stats file"; tail -5 gpurun_out/prof_$tag.log; fi
