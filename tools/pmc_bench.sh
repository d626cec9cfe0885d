#!/bin/bash
# HBM traffic per launch of the TOKEN's kernels on the full 32-layer model: two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# --kernel-trace only beside them) over bench.py with the driver's --steps 20 --warmup 5; writes gpurun_out/pmc/pmc_fetch_write_raw.json  (k_layers = all 32 layers in one launch: its counters are a whole
# token's layers)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc/$ctr -o run -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-decode128 > gpurun_out/pmc/$ctr.log 2>&1
done
python3 tools/pmc_agg.py --last 25 gpurun_out/pmc/FETCH_SIZE gpurun_out/pmc/WRITE_SIZE > gpurun_out/pmc/pmc_fetch_write_raw.json      # (the last 25 dispatches of a kernel: a leg's 5 warm-up + 20 timed steps, not the load-time warm-up's replays)
rm -rf gpurun_out/pmc/FETCH_SIZE gpurun_out/pmc/WRITE_SIZE
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/pmc/pmc_fetch_write_raw.json"))
for k, v in d.items():
    if "k_gemv" in k or "attn" in k or "k_layers" in k:
        f = v.get("FETCH_SIZE", {}).get("mean", 0); w = v.get("WRITE_SIZE", {}).get("mean", 0)
        print(f"{k[:60]:60s} fetch x2 {2*f*1024/1e6:9.2f} MB  write {w*1024/1e6:8.3f} MB  (n={v.get('FETCH_SIZE', {}).get('n', 0)})")
PY
