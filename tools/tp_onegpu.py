"""tensor-parallel ranks as threads on ONE GPU under CU masks: us per token by launch structure.   python tools/tp_onegpu.py [world] [layers] [tokens]
(a data-flow / launch-count measurement: the ranks share one HBM, so this is no scaling number -- it shows what a launch costs a sharded layer)"""
import sys, os, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ntok = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = L
tensors = synth.make_tensors(cfg, seed=1, share_layers=True)
ctxs = [capi.Ctx(capi.desc_from_config(cfg), device=0, rank=r, world=world, comm_id=None) for r in range(world)]
for c in ctxs: c.upload_all(tensors)
fence = os.environ.get("FLM_TP_FENCE")    # the rank-spanning launch's fences (tuning dial "tp_fence": bit 0 release, bit 1 acquire; default 3)
for c in ctxs:
    c.set_option("cu_parts", world)
    if fence is not None:
        c.set_option("tuning", 1); c.set_option("tp_fence", int(fence))
for kv in filter(None, os.environ.get("FLM_TP_OPTS", "").split(",")):      # e.g. FLM_TP_OPTS=tok_preq=12,back_pre13=8 (tuning dials, every rank)
    for c in ctxs:
        c.set_option("tuning", 1); c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
capi.Ctx.regroup(ctxs)
prompt = (np.arange(1, 9, dtype=np.int64) * 7919 % cfg.vocab_size).astype(np.int32)

def run(fn):
    out = [None] * world
    th = [threading.Thread(target=lambda r=r: out.__setitem__(r, fn(ctxs[r]))) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    return out

only = os.environ.get("FLM_TP_ONLY")      # e.g. "1,2,1": that structure alone (for a profile)
for fold, fa, fn, name in ((1, 2, 0, "ALL layers in one rank-spanning launch (k_layers<TP>), data-tagged granules"), (1, 2, 0, "ALL layers in one rank-spanning launch (k_layers<TP>), flag rounds"), (0, 0, 0, "k_xchg launches (9 per layer)"), (1, 0, 0, "folded exchanges (5 per layer)"), (1, 1, 0, "folded + attention and Wo in one launch (4 per layer)"),
                           (1, 2, 0, "folded + QKV, attention and Wo in one launch (3 per layer)"), (1, 1, 1, "folded + attention and Wo, FFN13 and FFN2 fused (3 per layer)"),
                           (1, 2, 1, "folded + QKV, attention, Wo | FFN13, FFN2 (2 per layer)")):
    tpl = 1 if name.startswith("ALL") else 0
    gr = 1 if "granules" in name else 0
    if only and only != (f"{fold},{fa},{fn}" if not tpl else ("tplg" if gr else "tpl")): continue
    for c in ctxs:
        c.set_option("gr_edges", gr); c.set_option("tp_fuse_layers", tpl); c.set_option("fold_xchg", fold); c.set_option("tp_fuse_attn", fa); c.set_option("tp_fuse_ffn", fn); c.reset_kv()
    capi.Ctx.regroup(ctxs)              # the group's launch structure is agreed when the blobs are exchanged
    first = run(lambda c: c.forward_argmax(prompt, 0))[0]
    run(lambda c: c.decode_greedy(first, len(prompt), 8))
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        ids = run(lambda c: c.decode_greedy(first, len(prompt), ntok))
        best = min(best, time.perf_counter() - t0)
    assert all(list(i) == list(ids[0]) for i in ids)
    dev_ms = run(lambda c: c.decode_timed(first, len(prompt), ntok))          # HIP events on each rank's own stream
    name += f" [device: {max(dev_ms) / ntok * 1e3:.1f} us]"
    print(f"tp{world} on one GPU ({256 // world} CUs per rank), {L} layers of 7B width: {name:90s} {best / ntok * 1e6:8.1f} us per token  ids {list(ids[0][:4])}")
