#!/bin/bash
# build experiment variants of the HIP library side by side:  tools/variants.sh tag1="-DFOO=1" tag2="-DFOO=2 -DBAR" ...
# -> fast-llama_amd/lib/var/libflm_<tag>.so ; run one with FLM_GPU_LIB=fast-llama_amd/lib/var/libflm_<tag>.so python tools/kbench.py
set -e
cd "$(dirname "$0")/.."
mkdir -p fast-llama_amd/lib/var
for spec in "$@"; do
  tag="${spec%%=*}"; defs="${spec#*=}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-result -Wno-unused-value $defs -shared \
      -Iinclude -Ifast-llama_amd/csrc -o fast-llama_amd/lib/var/libflm_$tag.so fast-llama_amd/csrc/flm_gpu.hip -L/opt/rocm/lib -lrccl && echo "built $tag ($defs)" ) &
done
wait
