cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=8
V=fast-llama_amd/lib/var
for k in ffn13 qkv; do FLM_TRACE_PRO=1 FLM_GPU_LIB=$V/libflm_tr.so timeout 120 python tools/trace.py $k 2 64 2>&1 | tail -12; done
for ab in 0 2; do echo "== ablate $ab"; FLM_GPU_LIB=$V/libflm_ab.so timeout 120 python tools/kbench.py 2 64 0 $ab 2>&1 | tail -9; done
FLM_SQ_ITERS=1 timeout 100 python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi
rng = np.random.default_rng(1)
for n in (768, 4096, 8192):
    x = rng.standard_normal(n).astype(np.float32)
    print(n, capi.op_square_sum(x)[:2])
PY
