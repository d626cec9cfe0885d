"""which workgroups of a GEMV launch start / finish late?  (FLM_ABLATE build)  python tools/trace_skew.py [kclass] [layers] [pos]
per workgroup: s_memtime at start and end; printed by blockIdx octile, by blockIdx % 8 (the XCD a workgroup lands on) and as a correlation"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
kname = sys.argv[1] if len(sys.argv) > 1 else "ffn13"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pos = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = L
ctx = capi.Ctx(capi.desc_from_config(cfg))
ctx.upload_all(synth.make_tensors(cfg, seed=1))
prompt = np.arange(1, pos + 1, dtype=np.int32) % cfg.vocab_size
first = ctx.forward_argmax(prompt, 0)
ctx.decode_greedy(first, pos, 8)
ctx.set_option("trace", capi.KCLASSES.index(kname))
ctx.set_option("use_graph", 0)
ctx.set_option("ablate", 64)
acc_s, acc_e = [], []
for rep in range(6):
    ctx.decode_greedy(first, pos + 8 + rep, 1)
    t = ctx.debug_read("trace_abs", 1, 256 * 8).reshape(256, 8).astype(np.float64)
    s = t[:, 1] / 100.0; e = t[:, 2] / 100.0          # 100 MHz ticks -> us
    acc_s.append(s); acc_e.append(e)
s = np.mean(acc_s[1:], 0); e = np.mean(acc_e[1:], 0)
idx = np.arange(256)
print(f"{kname}: start  min {s.min():.2f} median {np.median(s):.2f} max {s.max():.2f} | end min {e.min():.2f} median {np.median(e):.2f} max {e.max():.2f} us (mean of 5 launches)")
print("corr(blockIdx, start) %.2f  corr(blockIdx, end) %.2f  corr(start, end) %.2f" % (np.corrcoef(idx, s)[0, 1], np.corrcoef(idx, e)[0, 1], np.corrcoef(s, e)[0, 1]))
print("by blockIdx octile : start " + " ".join(f"{s[i*32:(i+1)*32].mean():5.2f}" for i in range(8)))
print("                     end   " + " ".join(f"{e[i*32:(i+1)*32].mean():5.2f}" for i in range(8)))
print("by blockIdx % 8    : start " + " ".join(f"{s[i::8].mean():5.2f}" for i in range(8)))
print("                     end   " + " ".join(f"{e[i::8].mean():5.2f}" for i in range(8)))
one = acc_e[-1]; print("single launch: end sorted, the last 12 workgroups:", " ".join(f"{int(i)}:{one[int(i)]:.2f}" for i in np.argsort(one)[-12:]))
rep_corr = np.corrcoef(acc_e[1], acc_e[2])[0, 1]; print("corr(end of launch 1, end of launch 2) %.2f" % rep_corr)
