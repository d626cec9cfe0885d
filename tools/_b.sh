#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b
timeout 600 python bench.py --config prefill512-int8 --steps 5 --warmup 2 > gpurun_out/b/pf8.json 2> gpurun_out/b/pf8.err; tail -c 600 gpurun_out/b/pf8.json
timeout 600 python bench.py --config prefill512-int16 --steps 5 --warmup 2 > gpurun_out/b/pf16.json 2> gpurun_out/b/pf16.err; tail -c 600 gpurun_out/b/pf16.json
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/b/n1.json 2> gpurun_out/b/n1.err; head -c 700 gpurun_out/b/n1.json
