#!/bin/bash
# build experiment variants of the HIP library side by side:  tools/variants.sh tag1="-DFOO=1" tag2="-DFOO=2 -DBAR" ...
# -> fast-llama_amd/lib/var/libflm_<tag>.so ; run one with FLM_GPU_LIB=fast-llama_amd/lib/var/libflm_<tag>.so python tools/kbench.py
# (the translation units of a variant compile side by side: __graft_entry__.build with FLM_EXTRA_DEFS / FLM_BUILD_LIB; objects are cached per set of defines)
set -e
cd "$(dirname "$0")/.."
mkdir -p fast-llama_amd/lib/var
for spec in "$@"; do
  tag="${spec%%=*}"; defs="${spec#*=}"
  ( FLM_ALLOW_SPILLS=1 FLM_EXTRA_DEFS="$defs" FLM_BUILD_LIB="$PWD/fast-llama_amd/lib/var/libflm_$tag.so" FLM_LIB_ONLY=1 python -c "import __graft_entry__ as g; g.build()" > /tmp/variant_$tag.log 2>&1 && echo "built $tag ($defs)" || { echo "FAILED $tag"; tail -5 /tmp/variant_$tag.log; } ) &
done
wait
