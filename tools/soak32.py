"""the full 32-layer LLaMA2-7B int8 synthetic model: a 1000-token greedy decode with the default token path (one launch per token, graphs of eight tokens, arrival-order hand-offs)
against the same decode with one launch per phase -- any stale cross-workgroup read, flag epoch slip or graph-chunk mistake changes the ids.  python tools/soak32.py [tokens] [reps]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = synth.make_config("7B", ff.QT_INT8)
prompt = np.array([1] + [int(x) for x in (np.arange(1, 9) * 7919) % cfg.vocab_size], dtype=np.int32)
ref = None
for opts in ({"fuse_attn_o": 0, "fuse_ffn": 0, "attn_split": 0, "fuse_back": 0}, {}, {"gr_edges": 0}, {"fuse_tail": 0}, {"graph_chunks": 0}, {"back_ao": 0}):     # ({}: the default -- the one-launch token on granules, round 6)
    ctx = capi.Ctx(capi.desc_from_config(cfg)); bench.upload_synthetic(ctx, cfg)
    for k, v in opts.items(): ctx.set_option(k, v)
    for r in range(reps if opts else 1):
        ctx.reset_kv()
        first = ctx.forward_argmax(prompt, 0)
        ids = [first] + list(ctx.decode_greedy(first, len(prompt), n))
        if ref is None: ref = ids
        print(opts, "rep", r, "identical" if ids == ref else f"MISMATCH at {next(i for i, (a, b) in enumerate(zip(ids, ref)) if a != b)}", "fallback", ctx.query("fallback"), flush=True)
    ctx.close()
gold = bench.golden_ids(cfg, ff.QT_INT8, 9)
if gold is not None: print("first", min(len(gold), len(ref)), "ids against the reference's:", ref[:len(gold)] == list(gold)[:len(ref)])
