cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
V=fast-llama_amd/lib/var
for v in tr2 tr2l; do echo "== $v"; FLM_TRACE_PRO=1 FLM_GPU_LIB=$V/libflm_$v.so timeout 120 python tools/trace.py ffn13 2 64 2>&1 | tail -9; done
echo "== product"; timeout 120 python tools/kbench.py 4 64 2>&1 | tail -9
for v in late early earlylate; do echo "== $v"; FLM_GPU_LIB=$V/libflm_$v.so timeout 120 python tools/kbench.py 4 64 2>&1 | tail -9; done
echo "== product fuse=0"; FLM_FUSE=0 timeout 120 python tools/kbench.py 4 64 2>&1 | head -1
