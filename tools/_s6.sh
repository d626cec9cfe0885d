cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
V=fast-llama_amd/lib/var
for ab in 0 2; do FLM_TRACE_PRO=1 FLM_GPU_LIB=$V/libflm_tr2.so timeout 120 python tools/trace.py ffn13 2 64 $ab 2>&1 | tail -10; done
