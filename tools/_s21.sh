cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "op_matmul or config5" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "prefill or logits_and_greedy" 2>&1 | tail -3
echo "== int16 n=512"; FLM_PF_ONLY=1 timeout 200 python tools/prefill_bench.py 4 512 int16 2>&1 | tail -1
echo "== int16 n=128"; FLM_PF_ONLY=1 timeout 200 python tools/prefill_bench.py 4 128 int16 2>&1 | tail -1
echo "== int16 n=1000"; FLM_PF_ONLY=1 timeout 200 python tools/prefill_bench.py 4 1000 int16 2>&1 | tail -1
