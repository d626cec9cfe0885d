cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
V=fast-llama_amd/lib/var
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "attention" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "long_context or fused_attention" 2>&1 | tail -5
for pos in 200 507 900; do echo "== pos $pos split 1"; FLM_SPLIT=1 FLM_GPU_LIB=$V/libflm_ab.so timeout 120 python tools/trace_attn.py 2 $pos 2>&1 | tail -1; done
for pos in 200 512 900; do echo "== pos $pos attn_split=1"; FLM_SPLIT=1 timeout 200 python tools/kbench.py 4 $pos 2>&1 | head -4 | grep -v embed; done
