"""when each wave of a GEMV workgroup finishes its steps (FLM_ABLATE + FLM_TRACE_WAVES build): python tools/trace_waves.py [kclass] [layers] [pos] [ablate]"""
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
t = t[t[:, 1] > 0]
mhz = np.median(t[:, 1:7].max(axis=1) / np.maximum(t[:, 7], 1)) * 100.0      # rough: last wave done ~ kernel end
print(f"{kname} ablate={abl}: {len(t)} workgroups; steps-done time per wave (us after workgroup start, clock ~{mhz:.0f} MHz assumed 2300)")
us = t[:, 1:7] / 2300.0
for k, w in enumerate((0, 3, 6, 9, 12, 15)):
    c = us[:, k]
    print(f"  wave {w:2d}: min {c.min():6.2f} median {np.median(c):6.2f} max {c.max():6.2f}")
print(f"  spread inside a workgroup (max-min over the 6 sampled waves): median {np.median(us.max(1)-us.min(1)):.2f} us")
