"""engine vs per-phase kernels on a 7B-width model: first differences in hd / x1 / logits.  python tools/eng_check.py [layers] [engine mode]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
shape = sys.argv[3] if len(sys.argv) > 3 else "7B"
cfg = synth.make_config(shape, ff.QT_INT8); cfg.n_layers = L
tensors = synth.make_tensors(cfg, seed=1)
res = {}
for eng in (0, mode):
    ctx = capi.Ctx(capi.desc_from_config(cfg)); ctx.upload_all(tensors)
    ctx.set_option("engine", eng)
    ctx.set_option("use_graph", 0)
    if os.environ.get("FLM_ABL") and eng: ctx.set_option("ablate", int(os.environ["FLM_ABL"]))
    prompt = np.array([5, 9, 100], dtype=np.int32)
    lg = ctx.forward(prompt, 0)
    res[eng] = dict(logits=lg.copy(), hd=ctx.debug_read("hd", 0, cfg.hidden_dim), x1=ctx.debug_read("x1", 0, cfg.dim), q=ctx.debug_read("q", 0, cfg.dim), att=ctx.debug_read("att_out", 0, cfg.dim))
    ctx.close()
for k in ("q", "att", "hd", "x1", "logits"):
    a, b = res[0][k], res[mode][k]
    bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
    print(f"{k:7s} n={a.size} mismatches={bad.size}", "first:", bad[:24].tolist(), "mod4 hist:", np.bincount(bad % 4, minlength=4).tolist() if bad.size else "")
    if bad.size: print("   ref", a[bad[:6]], "\n   eng", b[bad[:6]])
