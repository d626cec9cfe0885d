"""hand-off stress: long greedy decodes on a 7B-width model with the fused launches (k_layers with the arrival-order and the round-4 hand-offs, k_attn_ffn, k_attn_o, k_qkv_attn_o, k_ffn, split heads) against one launch per phase;
any stale cross-workgroup read changes the token ids.  python tools/stress.py [layers] [tokens] [reps]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
bad = 0
for qt, name in ((ff.QT_INT8, "int8"), (ff.QT_INT16, "int16")):
    cfg = synth.make_config("7B", qt); cfg.n_layers = L
    tensors = synth.make_tensors(cfg, seed=3, share_layers=True)
    prompt = (np.arange(1, 9, dtype=np.int64) * 7919 % cfg.vocab_size).astype(np.int32)
    ref = None
    base = {"fuse_attn_o": 0, "fuse_ffn": 0, "attn_split": 0, "fuse_back": 0}
    for opts in (base, {}, {"use_graph": 0}, {"fuse_token": 0}, {"tok_preq": 7, "tok_nstq": 11}, {"fuse_layer": 0}, {"fuse_back": 0}, {"back_nst13_head": 0, "back_pre13": 5}, {"back_nst2": 9, "back_pre2": 3}, {"back_ao": 0}, {"back_ao2": 1}, {"back_ao2": 3, "tok_preq": 3}, {"fuse_attn_o": 0}, {"attn_split": 0}, {"fuse_back": 0, "fuse_qkv": 2}):
        ctx = capi.Ctx(capi.desc_from_config(cfg)); ctx.upload_all(tensors)
        for k, v in opts.items(): ctx.set_option(k, v)
        for r in range(reps if opts is not base else 1):
            ctx.reset_kv()
            first = ctx.forward_argmax(prompt, 0)
            ids = [first] + list(ctx.decode_greedy(first, len(prompt), n))
            if ref is None: ref = ids
            ok = ids == ref
            if not ok:
                bad += 1
                k = next(i for i, (a, b) in enumerate(zip(ids, ref)) if a != b)
                print(f"{name} {opts} rep {r}: MISMATCH at generated token {k}")
        print(f"{name} {opts}: done", flush=True)
        ctx.close()
print("stress:", "FAILED" if bad else "ok")
