cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_gpu_tp.py -q -m gpu --durations=8 2>&1 | tail -30
