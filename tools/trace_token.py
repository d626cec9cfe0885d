"""timeline of the persistent token kernel (FLM_ABLATE build): python tools/trace_token.py [layers] [pos]"""
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
ctx.set_option("use_mega", 1)
prompt = np.arange(1, pos + 1, dtype=np.int32) % cfg.vocab_size
first = ctx.forward_argmax(prompt, 0)
ctx.decode_greedy(first, pos, 8)
ctx.set_option("trace", 100)
ctx.set_option("use_graph", 0)
ctx.decode_greedy(first, pos + 8, 1)
NW = 256
t = ctx.debug_read("trace", 16, NW * 16 * 8).reshape(NW, 16, 8)
MHZ = 2370.0
names = ["qkv", "attn+o", "ffn13", "ffn2"] * L + ["cls"]
print("per phase, median over workgroups (us): start | prefetch  barrier-wait  prologue  gemv | total")
for p, nm in enumerate(names[:16]):
    d = t[:, p, :5]
    ok = (d >= 0).all(axis=1)
    if not ok.any(): continue
    d = d[ok] / MHZ
    dd = d[:, 1:] - d[:, :-1]
    st = np.median(dd, axis=0); mx = np.max(dd, axis=0)
    print(f"  {p:2d} {nm:7s} @{np.median(d[:,0]):8.2f} | {st[0]:7.2f} {st[1]:7.2f} {st[2]:7.2f} {st[3]:7.2f} | {np.median(d[:,4]-d[:,0]):7.2f}   max: {mx[0]:6.2f} {mx[1]:6.2f} {mx[2]:6.2f} {mx[3]:6.2f}")
