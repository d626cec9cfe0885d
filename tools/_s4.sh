cd $GRAFT_REPO_ROOT
nproc; cat /proc/loadavg; echo OMP=$OMP_NUM_THREADS
(time timeout 280 python -m pytest tests/test_gpu_model.py -q -m gpu -k "test_logits_and_greedy_vs_oracle" --durations=8 2>&1 | tail -20)
cat /proc/loadavg
(time OMP_NUM_THREADS=8 timeout 280 python -m pytest tests/test_gpu_model.py -q -m gpu -k "test_logits_and_greedy_vs_oracle" --durations=8 2>&1 | tail -20)
