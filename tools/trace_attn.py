"""timeline of the attention kernel (FLM_ABLATE build): python tools/trace_attn.py [layers] [pos]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pos = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = L
ctx = capi.Ctx(capi.desc_from_config(cfg))
ctx.upload_all(synth.make_tensors(cfg, seed=1))
if os.environ.get("FLM_SPLIT") is not None: ctx.set_option("attn_split", int(os.environ["FLM_SPLIT"]))
prompt = np.arange(1, pos + 1, dtype=np.int32) % cfg.vocab_size
first = ctx.forward_argmax(prompt, 0)
ctx.decode_greedy(first, pos, 4)
ctx.set_option("trace", capi.KCLASSES.index("attn"))
ctx.set_option("use_graph", 0)
ctx.decode_greedy(first, pos + 4, 1)
raw = ctx.debug_read("trace", 1, 128 * 8).reshape(128, 8) / 2200.0
raw = raw[raw[:, 4] > 0]
t = raw[:, :5]
d = t[:, 1:] - t[:, :-1]
print(f"T={pos+5} ({len(t)} workgroups): median stage us: scores {np.median(d[:,0]):.2f} | max+exp {np.median(d[:,1]):.2f} | sum chain {np.median(d[:,2]):.2f} | divide+PV {np.median(d[:,3]):.2f} | total {np.median(t[:,4]):.2f}")
