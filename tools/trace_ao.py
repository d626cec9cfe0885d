"""timeline of the fused attention + Wo kernel (FLM_ABLATE build): python tools/trace_ao.py [layers] [pos]"""
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
prompt = np.arange(1, pos + 1, dtype=np.int32) % cfg.vocab_size
first = ctx.forward_argmax(prompt, 0)
ctx.decode_greedy(first, pos, 8)
ctx.set_option("trace", 101)
ctx.set_option("use_graph", 0)
ctx.decode_greedy(first, pos + 8, 1)
t = ctx.debug_read("trace", 1, 256 * 8).reshape(256, 8) / 2300.0      # ticks relative to each workgroup's own start (XCD clocks differ)
nh = cfg.n_heads
h, w = t[:nh], t[nh:]; w = w[w[:, 4] > 0]
print(f"pos {pos + 8}: heads  : attention done {np.median(h[:, 1]):6.2f} us (max {h[:, 1].max():6.2f}), flag bumped {np.median(h[:, 4]):6.2f} (max {h[:, 4].max():6.2f})")
for k, nm in ((1, "weights requested"), (2, "heads' flag seen"), (3, "activation quantized"), (4, "end")):
    print(f"          gemv wg: {nm:20s} median {np.median(w[:, k]):6.2f}  min {w[:, k].min():6.2f}  max {w[:, k].max():6.2f} us after the workgroup's start")
