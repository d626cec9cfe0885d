"""graph decode on a 7B-width model with few layers under an option set (for rocprofv3): python tools/decode_opts.py [layers] [pos] [tokens] ["k=v,k=v"]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pos = int(sys.argv[2]) if len(sys.argv) > 2 else 14
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
spec = sys.argv[4] if len(sys.argv) > 4 else ""
cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = L
ctx = capi.Ctx(capi.desc_from_config(cfg))
ctx.upload_all(synth.make_tensors(cfg, seed=1))
for kv in spec.split(","):
    if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
prompt = (np.arange(1, pos + 1, dtype=np.int64) * 7919 % cfg.vocab_size).astype(np.int32)
first = ctx.forward_argmax(prompt, 0)
ms = ctx.decode_timed(first, len(prompt), n)
print(f"{spec or 'defaults'}: {ms / n * 1000:.1f} us/token, token_path {ctx.query('token_path')}, fallback {ctx.query('fallback')}")
