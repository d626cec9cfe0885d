cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
V=fast-llama_amd/lib/var
for pos in 60 507 900; do for sp in 1 0; do echo "== pos $pos split $sp"; FLM_SPLIT=$sp FLM_GPU_LIB=$V/libflm_ab.so timeout 120 python tools/trace_attn.py 2 $pos 2>&1 | tail -1; done; done
