"""prompt-processing time: python tools/prefill_bench.py [layers] [prompt_len] [quant]   (7B width)"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
qt = ff.QT_INT16 if len(sys.argv) > 3 and sys.argv[3] == "int16" else ff.QT_INT8
cfg = synth.make_config("7B", qt); cfg.n_layers = L
ctx = capi.Ctx(capi.desc_from_config(cfg))
ctx.upload_all(synth.make_tensors(cfg, seed=1, share_layers=True))
if os.environ.get("FLM_MFMA") is not None: ctx.set_option("use_mfma", int(os.environ["FLM_MFMA"]))   # 2 / 3: 64 x 64 / 128 x 128 tiles
if os.environ.get("FLM_QKMFMA") is not None: ctx.set_option("use_qk_mfma", int(os.environ["FLM_QKMFMA"]))   # 0: scores on VALU chains
if os.environ.get("FLM_MQ") is not None: ctx.set_option("use_prefill_mq", int(os.environ["FLM_MQ"]))   # 0: one query per workgroup
prompt = (np.arange(1, n + 1, dtype=np.int64) * 7919 % cfg.vocab_size).astype(np.int32)
for mode in ((1, 1) if os.environ.get("FLM_PF_ONLY") else (1, 0, 1)):
    ctx.set_option("use_prefill", mode); ctx.reset_kv()
    ctx.sync(); t0 = time.perf_counter()
    tok = ctx.forward_argmax(prompt, 0)
    dt = time.perf_counter() - t0
    print(f"L={L} n={n} use_prefill={mode}: {dt*1e3:8.2f} ms  ({n/dt:8.0f} prompt tokens/s, {dt/n/L*1e6:6.2f} us per token-layer)  next={tok}")
