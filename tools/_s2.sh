cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
(timeout 500 python -m pytest tests/test_gpu_model.py -q -m gpu --timeout 100 --durations=12 -x 2>&1 | tail -40) > gpurun_out/s2/model.log 2>&1
cat gpurun_out/s2/model.log
