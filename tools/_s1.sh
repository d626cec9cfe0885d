cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
(timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -15) > gpurun_out/s1/ops.log 2>&1
(timeout 600 python tools/kbench.py 4 64 2>&1 | tail -12) > gpurun_out/s1/kbench.log 2>&1
(timeout 1500 python -m pytest tests/test_gpu_model.py -q -m gpu 2>&1 | tail -25) > gpurun_out/s1/model.log 2>&1
(timeout 1800 python -m pytest tests/test_gpu_configs.py -q -m gpu 2>&1 | tail -40) > gpurun_out/s1/configs.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_cli.py -q -m gpu 2>&1 | tail -8) > gpurun_out/s1/cli.log 2>&1
cat gpurun_out/s1/ops.log gpurun_out/s1/kbench.log; tail -5 gpurun_out/s1/model.log; tail -12 gpurun_out/s1/configs.log; tail -3 gpurun_out/s1/cli.log
