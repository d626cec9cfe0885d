cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
V=fast-llama_amd/lib/var
echo "== product"; timeout 120 python tools/kbench.py 4 64 2>&1 | tail -9 | head -7
for v in t4 t8 t16; do echo "== $v"; FLM_GPU_LIB=$V/libflm_$v.so timeout 120 python tools/kbench.py 4 64 2>&1 | tail -9 | head -7; done
echo "== product again"; timeout 120 python tools/kbench.py 4 64 2>&1 | head -1
for m in 1 2; do echo "== int8 n=1000 use_mfma=$m"; FLM_PF_ONLY=1 FLM_MFMA=$m timeout 200 python tools/prefill_bench.py 4 1000 2>&1 | tail -1; done
