cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
V=fast-llama_amd/lib/var
echo "== product int16 512"; FLM_PF_ONLY=1 timeout 200 python tools/prefill_bench.py 4 512 int16 2>&1 | tail -1
for v in gnc gnl; do echo "== $v"; FLM_PF_ONLY=1 FLM_GPU_LIB=$V/libflm_$v.so timeout 200 python tools/prefill_bench.py 4 512 int16 2>&1 | tail -1; done
