#!/bin/bash
# usage: tools/abl.sh "<ablate values>"   -> rocprof kernel durations per ablation
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ab in $1; do
  rm -rf gpurun_out/prof_abl; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_abl -o run -- python tools/kbench.py 2 64 0 $ab > /dev/null 2>&1
  echo "== ablate=$ab"; [ -f gpurun_out/prof_abl/run_kernel_stats.csv ] && python3 tools/kstats.py gpurun_out/prof_abl/run_kernel_stats.csv | grep gemv
done
