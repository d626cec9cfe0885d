mkdir -p gpurun_out/s22
for pos in 300 516 900; do python tools/back_bench.py 32 $pos int8 "tuning=1,attn_kpre=0;attn_kpre=1;attn_kpre=0;attn_kpre=1"; done > gpurun_out/s22/kpre.txt 2>&1; cat gpurun_out/s22/kpre.txt
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = 2
ctx = capi.Ctx(capi.desc_from_config(cfg)); ctx.upload_all(synth.make_tensors(cfg, seed=1))
p = (np.arange(1, 301, dtype=np.int64) * 7919 % cfg.vocab_size).astype(np.int32)
f = ctx.forward_argmax(p, 0); ctx.decode_greedy(f, len(p), 4)
print("kpre_active", ctx.query("kpre_active"))
PY
(timeout 900 python -m pytest tests -m gpu -q -x -k "long_context or split or config3 or rank_spanning or fuzz or one_launch or config5" > gpurun_out/s22/sel.log 2>&1; echo rc=$? >> gpurun_out/s22/sel.log); tail -3 gpurun_out/s22/sel.log
