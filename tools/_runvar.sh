timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "prefill or long_context" 2>&1 | grep -v DEBUG | tail -3
for n in 128 512 1000; do for mq in 1; do echo "== n=$n mq=$mq"; FLM_MQ=$mq timeout 300 python tools/prefill_bench.py 4 $n 2>&1 | grep -v DEBUG | grep "use_prefill=1" | tail -1; done; done
