"""k_attn_ffn (attention + Wo + FFN13 + FFN2 in one launch, [W1; W3] / W2 stashed in LDS) against the two launches it replaces, on a 7B-WIDTH model with few
layers: graph decode time per token for a list of option sets, ids checked against the first set.  python tools/back_bench.py [layers] [pos] [quant]"""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pos = int(sys.argv[2]) if len(sys.argv) > 2 else 14
qt = ff.QT_INT16 if len(sys.argv) > 3 and sys.argv[3] == "int16" else ff.QT_INT8
cfg = synth.make_config("7B", qt); cfg.n_layers = L
ctx = capi.Ctx(capi.desc_from_config(cfg, max_seq_len=int(os.environ.get("FLM_MAXLEN", "1024"))))   # (FLM_MAXLEN: the KV cache rows per head = the stride between two heads)
ctx.upload_all(synth.make_tensors(cfg, seed=1))
prompt = (np.arange(1, pos + 1, dtype=np.int64) * 7919 % cfg.vocab_size).astype(np.int32)
specs = sys.argv[4].split(";") if len(sys.argv) > 4 else ["fuse_back=0", "fuse_back=1,back_nst13=0", "back_nst13=-1", "back_pre13=1", "back_nst13_head=-1", "back_pre13=0,back_nst13_head=0,back_nst2=8", "back_nst2=0,fuse_back=0"]
sets = [(sp, dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in sp.split(",") if kv)) for sp in specs]      # (options persist from one set to the next)
ref = None
N = 48
for name, opts in sets:
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.reset_kv()
    first = ctx.forward_argmax(prompt, 0)
    ids = [first] + list(ctx.decode_greedy(first, len(prompt), N))
    if ref is None: ref = ids
    best = 1e9
    for rep in range(4):
        ms = ctx.decode_timed(first, len(prompt), N)
        best = min(best, ms / N * 1000)
    print(f"{name:42s} {best:8.1f} us/token  {best / L:6.2f} us/layer-ish  ids {'ok' if ids == ref else 'MISMATCH'}  fallback {ctx.query('fallback')}", flush=True)
