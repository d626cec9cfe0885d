import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
def T(msg, t0): print(f"{msg}: {time.time()-t0:.3f}s", flush=True)
for shape in ("tiny", "small"):
    cfg = synth.make_config(shape, ff.QT_INT8)
    tensors = synth.make_tensors(cfg, seed=1234)
    t0=time.time(); ctx = capi.Ctx(capi.desc_from_config(cfg)); T(shape+" create", t0)
    t0=time.time(); ctx.upload_all(tensors); T("upload", t0)
    prompt = np.array([1,5,9,11,200,33,7,8], np.int32)
    t0=time.time(); lg = ctx.forward(prompt, 0); T("forward prompt 8", t0)
    t0=time.time(); lg = ctx.forward(prompt[:3], 0); T("forward prompt 3 (token by token)", t0)
    for i in range(3):
        t0=time.time(); lg = ctx.forward(np.array([5], np.int32), 8+i); T("forward 1", t0)
    t0=time.time(); ids = ctx.decode_greedy(5, 11, 16); T("decode_greedy 16", t0)
    for opt in (("fuse_attn_o",0),("use_graph",0)):
        ctx.set_option(*opt)
        t0=time.time(); lg = ctx.forward(np.array([5], np.int32), 30); T(f"forward 1 {opt}", t0)
        t0=time.time(); lg = ctx.forward(np.array([5], np.int32), 31); T(f"forward 1 {opt} again", t0)
    t0=time.time(); ctx.close(); T("close", t0)
