cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "prefill or long_context_positions" --durations=5 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "config5" 2>&1 | tail -3
for q in 1 0; do echo "== int8 n=512 qk_mfma=$q"; FLM_PF_ONLY=1 FLM_QKMFMA=$q timeout 200 python tools/prefill_bench.py 4 512 2>&1 | tail -1; done
for q in 1 0; do echo "== int16 n=512 qk_mfma=$q"; FLM_PF_ONLY=1 FLM_QKMFMA=$q timeout 200 python tools/prefill_bench.py 4 512 int16 2>&1 | tail -1; done
echo "== profile int16 512"; FLM_PF_ONLY=1 bash tools/prof_prefill.sh 4 512 int16 2>&1 | tail -14
