O=gpurun_out/s28; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "deep_in_the_context or one_launch" > $O/t.log 2>&1); tail -15 $O/t.log
