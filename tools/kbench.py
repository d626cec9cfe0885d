"""per-kernel timing on a 7B-WIDTH model with few layers (fast to upload): python tools/kbench.py [layers] [pos]"""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pos = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = L
ctx = capi.Ctx(capi.desc_from_config(cfg))
ctx.upload_all(synth.make_tensors(cfg, seed=1))
if wg: ctx.set_option("wg_per_cu", wg)
abl = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if abl: ctx.set_option("ablate", abl)
if os.environ.get("FLM_SPLIT") is not None: ctx.set_option("attn_split", int(os.environ["FLM_SPLIT"]))
if os.environ.get("FLM_FUSE") is not None: ctx.set_option("fuse_attn_o", int(os.environ["FLM_FUSE"]))
prompt = np.arange(1, pos + 1, dtype=np.int32) % cfg.vocab_size
first = ctx.forward_argmax(prompt, 0)
ms = ctx.decode_timed(first, pos, 64)
print(f"L={L} pos={pos} wg={wg} ablate={abl}: graph decode {ms/64*1000:.1f} us/token")
kt = ctx.kernel_times(pos + 64, iters=5)
for k, (us, cnt) in kt.items():
    if cnt: print(f"  {k:8s} {us:8.2f} us x{cnt}  {ctx.kernel_bytes(k, pos+64)/us/1e3 if us else 0:8.1f} GB/s")
