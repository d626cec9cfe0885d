"""in-kernel timeline of one engine launch (7B width): python tools/trace_eng.py [layers] [engine mode] [first phase of the launch] [pos]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ph0 = int(sys.argv[3]) if len(sys.argv) > 3 else 6
pos = int(sys.argv[4]) if len(sys.argv) > 4 else 32
NPH = 4 if mode == 2 else 2
cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = L
ctx = capi.Ctx(capi.desc_from_config(cfg)); ctx.upload_all(synth.make_tensors(cfg, seed=1))
ctx.set_option("engine", mode)
if os.environ.get("FLM_ABL"): ctx.set_option("ablate", int(os.environ["FLM_ABL"]))
prompt = np.arange(1, pos + 1, dtype=np.int32) % cfg.vocab_size
first = ctx.forward_argmax(prompt, 0)
ctx.decode_greedy(first, pos, 8)
ctx.set_option("eng_trace", ph0)
ctx.decode_greedy(first, pos, 1)
t = ctx.debug_read("eng_trace", 0, 256 * 512).reshape(256, 512)
def col(i, short=False):
    v = t[:, i]; v = v[v >= 0]
    if not v.size: return "  -"
    return f"{np.median(v):5.1f}" if short else f"{np.median(v):6.2f} [{v.min():6.2f} {v.max():6.2f}]"
print("us after the launch's earliest stamp, median over the workgroups (first line: median [min max]); per phase: prologue done / stream done (/ consumer 0: own part staged, 2: all staged, 1: chains done, 11: group owner's duty done)")
for w in range(12):
    print(f"consumer {w:2d}: start {col(16*w, True)} | " + " | ".join(f"ph{k} {col(16*w+1+3*k, True)} / {col(16*w+2+3*k, True)}" + (f" / {col(16*w+3+3*k, True)}" if w in (0, 1, 2, 11) else "") for k in range(NPH))
          + f" | slow waits {col(16*w+15, True)} | ns/piece {col(16*w+14, True)} | waiting for fills {col(320+4*w, True)} dots {col(320+4*w+1, True)} epilogues {col(320+4*w+2, True)} slots {col(320+4*w+3, True)}")
for l in range(4):
    print(f"loader {l}: start {col(256+8*l, True)} | issued: " + " ".join(f"ph{k} {col(256+8*l+1+k)}" for k in range(NPH)) + f" | waiting for slots {col(256+8*l+7)}")
