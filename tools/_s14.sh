cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
echo "== 2 ranks on one GPU (IPC peers, gloo bootstrap), 1.3B shape"
FLM_BENCH_FORCE_DEVICE=0 GPU_MAX_HW_QUEUES=16 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 16 --warmup 4 --shape 1.3B --no-cpu-baseline 2>&1 | tail -12
echo "== N=1 7B"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -8
echo "== N=1 7B pos 512"
timeout 900 python bench.py --steps 20 --warmup 5 --pos 512 --no-cpu-baseline 2>&1 | tail -3
