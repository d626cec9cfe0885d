cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
echo "== stress (1000-token decodes, 4 layers 7B width, fused / split / graph variants)"; timeout 900 python tools/stress.py 4 1000 3 2>&1 | tail -12
echo "== stress2"; timeout 600 python tools/stress2.py 200 2>&1 | tail -8
echo "== TP x10"; for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python -m pytest tests/test_gpu_tp.py -q -m gpu 2>&1 | tail -1; done
echo "== suite x2"; for i in 1 2; do timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -1; done
echo "== bench 2 procs on one GPU"; FLM_BENCH_FORCE_DEVICE=0 GPU_MAX_HW_QUEUES=16 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 16 --warmup 4 --shape 1.3B --no-cpu-baseline 2>/dev/null | cut -c1-700
