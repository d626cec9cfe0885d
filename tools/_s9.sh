cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "attention" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "long_context or fused_attention or 7b_width" --durations=5 2>&1 | tail -12
for pos in 64 200 512 900; do for sp in 1 0; do echo "== pos $pos attn_split=$sp"; FLM_SPLIT=$sp timeout 200 python tools/kbench.py 4 $pos 2>&1 | head -5 | grep -v embed; done; done
