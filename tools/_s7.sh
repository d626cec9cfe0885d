cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
V=fast-llama_amd/lib/var
FLM_SQ_ITERS=1 timeout 100 python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi
rng = np.random.default_rng(1)
for n in (768, 4096, 4096, 8192):
    x = rng.standard_normal(n).astype(np.float32)
    print(n, capi.op_square_sum(x)[:2])
PY
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "square_sum or rmsnorm" 2>&1 | tail -3
for v in tr2 tr2l; do echo "== $v"; FLM_TRACE_PRO=1 FLM_GPU_LIB=$V/libflm_$v.so timeout 120 python tools/trace.py ffn13 2 64 2>&1 | tail -9; done
echo "== product"; timeout 120 python tools/kbench.py 4 64 2>&1 | tail -9
echo "== late"; FLM_GPU_LIB=$V/libflm_late.so timeout 120 python tools/kbench.py 4 64 2>&1 | tail -9
echo "== product fuse=0"; FLM_FUSE=0 timeout 120 python tools/kbench.py 4 64 2>&1 | head -1
