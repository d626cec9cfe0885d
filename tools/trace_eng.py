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
cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = L
ctx = capi.Ctx(capi.desc_from_config(cfg)); ctx.upload_all(synth.make_tensors(cfg, seed=1))
ctx.set_option("engine", mode)
if os.environ.get("FLM_ABL"): ctx.set_option("ablate", int(os.environ["FLM_ABL"]))
prompt = np.arange(1, pos + 1, dtype=np.int32) % cfg.vocab_size
first = ctx.forward_argmax(prompt, 0)
ctx.decode_greedy(first, pos, 8)
ctx.set_option("eng_trace", ph0)
ctx.decode_greedy(first, pos, 1)
t = ctx.debug_read("eng_trace", 0, 256 * 256).reshape(256, 256)
def col(i):
    v = t[:, i]; v = v[v >= 0]
    return f"{np.median(v):6.2f} [{v.min():6.2f} {v.max():6.2f}]" if v.size else "   -"
print("us after the launch's earliest stamp: median [min max] over the workgroups; wait = total time in slow waits on LDS sequence words")
for w in range(12):
    print(f"consumer {w}: start {col(8*w)} | " + " | ".join(f"ph{k} pro {col(8*w+1+3*k)} stream {col(8*w+2+3*k)}" + (f" hop1 {col(8*w+3+3*k)}" if w == 11 else "") for k in range(2)) + f" | wait {col(8*w+7)} | ns/piece {col(8*w+6)}")
for l in range(4):
    print(f"loader {l}: start {col(96+8*l)} | " + " | ".join(f"ph{k} issued {col(96+8*l+1+k)}" for k in range(2)) + f" | wait for slots {col(96+8*l+7)}")
for w in range(12):
    print(f"consumer {w}: us waiting for fills {col(128+4*w)} | in the dot loops {col(128+4*w+1)} | in epilogues {col(128+4*w+2)} | slots {col(128+4*w+3)}")
