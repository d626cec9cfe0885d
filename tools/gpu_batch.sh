O=gpurun_out/s38; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "deep_in_the_context or one_launch or option_and_query" > $O/t.log 2>&1); tail -12 $O/t.log
