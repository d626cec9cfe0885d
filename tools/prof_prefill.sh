#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_pf; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pf -o run -- python tools/prefill_bench.py "$@" > gpurun_out/prof_pf.log 2>&1
rm -f gpurun_out/prof_pf/run_kernel_trace.csv
f=gpurun_out/prof_pf/run_kernel_stats.csv
if [ -f "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e6:9.2f} ms total {float(r['AverageNs'])/1000:9.2f} us avg")
PY
else echo "no stats"; tail -3 gpurun_out/prof_pf.log; fi
