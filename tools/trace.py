"""in-kernel timeline of one GEMV class (FLM_ABLATE build): python tools/trace.py [kclass name] [layers] [pos]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
kname = sys.argv[1] if len(sys.argv) > 1 else "ffn13"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pos = int(sys.argv[3]) if len(sys.argv) > 3 else 64
abl = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = L
ctx = capi.Ctx(capi.desc_from_config(cfg))
ctx.upload_all(synth.make_tensors(cfg, seed=1))
if abl: ctx.set_option("ablate", abl)
prompt = np.arange(1, pos + 1, dtype=np.int32) % cfg.vocab_size
first = ctx.forward_argmax(prompt, 0)
ctx.decode_greedy(first, pos, 8)
ctx.set_option("trace", capi.KCLASSES.index(kname))
ctx.set_option("use_graph", 0)
ctx.decode_greedy(first, pos + 8, 1)
t = ctx.debug_read("trace", 1, 256 * 8).reshape(256, 8)
live = t[:, 0] >= 0
t = t[live]
rt = t[:, 7]; dur = t[:, 6] - t[:, 0]
mhz = np.median(dur[rt > 0] / rt[rt > 0]) * 100.0
print(f"{kname}: {live.sum()} workgroups, s_memtime runs at ~{mhz:.0f} MHz (vs 100 MHz realtime)")
us = t[:, :7] / mhz
names = ["start", "loads issued", "prologue done", "first step reduced", "pass-1 steps done", "pass-1 chain done", "end"]
if os.environ.get("FLM_TRACE_PRO"):
    us = us[:, [0, 1, 3, 4, 5, 2, 6]]
    names = ["start", "loads issued", "staged+sync", "chain done", "scale+sync", "prologue done", "end"]
for k, nm in enumerate(names):
    col = us[:, k]
    print(f"  {nm:22s} min {col.min():7.2f}  median {np.median(col):7.2f}  max {col.max():7.2f} us")
d = us[:, 1:] - us[:, :-1]
print("  stage durations (median):", " | ".join(f"{nm} {np.median(d[:, k]):.2f}" for k, nm in enumerate((["issue", "x wait+stage", "chain", "scale", "quantize", "main loop"] if os.environ.get("FLM_TRACE_PRO") else ["issue", "prologue", "first reduce", "rest of pass 1", "barrier+chain", "rest"]))))
