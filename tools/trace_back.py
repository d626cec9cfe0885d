"""timeline of k_attn_ffn (attention + Wo + FFN13 + FFN2 in one launch; FLM_ABLATE build), all workgroups on the 100 MHz clock:
FLM_GPU_LIB=fast-llama_amd/lib/var/libflm_abl.so python tools/trace_back.py [layers] [pos] ["k=v,k=v;k=v..." option sets] [103 | 102]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pos = int(sys.argv[2]) if len(sys.argv) > 2 else 14
sets = sys.argv[3].split(";") if len(sys.argv) > 3 else [""]
tclass = int(sys.argv[4]) if len(sys.argv) > 4 else 103          # 103: k_layers (its second layer), 102: k_attn_ffn (layer 0)
cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = L
ctx = capi.Ctx(capi.desc_from_config(cfg))
ctx.upload_all(synth.make_tensors(cfg, seed=1))
prompt = (np.arange(1, pos + 1, dtype=np.int64) * 7919 % cfg.vocab_size).astype(np.int32)
names = {0: "start", 1: "heads: attention done | Wo: weights + stash requested", 2: "heads: flag raised | Wo: heads' flags seen", 3: "Wo: rows done", 4: "Wo: x1 flag raised",
         5: "x1 flags seen", 6: "FFN13 prologue done", 7: "FFN13 rows done", 8: "hd flag raised", 9: "hd flags seen", 10: "FFN2 prologue done", 11: "end", 12: "QKV prologue done", 13: "QKV rows done", 14: "QKV flag raised", 15: "Wo sets + [W1; W3] stash landed (wave 15)"}
for spec in sets:
    opts = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in spec.split(",") if kv)
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.set_option("trace", -1); ctx.set_option("use_graph", 1)
    ctx.reset_kv()
    first = ctx.forward_argmax(prompt, 0)
    ms = ctx.decode_timed(first, len(prompt), 32)
    ctx.set_option("trace", tclass); ctx.set_option("use_graph", 0)
    rows = []
    for rep in range(5):
        ctx.decode_greedy(first, len(prompt) + 32 + rep, 1)
        rows.append(ctx.debug_read("back_trace", 0, 6 * 256 * 16).reshape(6, 256, 16).copy())
    pro = rows[-1][1:3]; rn = rows[-1][4]
    at = rows[-1][5].reshape(-1, 16)[:cfg.n_heads]
    ch = rows[-1][3]                  # the FFN13 chain's stages (wave 0): shader-clock ticks after the chain's start; [15] = rounds
    ok = ch[:, 14] > 0
    if ok.any():
        c = ch[ok]
        labels = ["scan", "head", "increments", "fp64 prefix", "r1", "r2", "r3", "r4", "r5", "r6", "r7", "r8", "r9"]
        seg = []
        prev = np.zeros(len(c))
        for k in range(1, 15):
            cur = c[:, k]; m = cur > 0
            if m.any(): seg.append(f"{labels[k - 1] if k < 14 else 'end'} +{np.median((cur - prev)[m]):.0f}")
            prev = np.where(m, cur, prev)
        print("  FFN13 chain (wave 0), shader ticks per stage (median): " + " ".join(seg) + f" | total {np.median(c[:, 14]):.0f} ticks, rounds {np.median(c[:, 15]):.0f}")
    rows = [r[0] for r in rows]
    t = rows[-1]
    nh = cfg.n_heads
    print(f"--- {spec or 'defaults'}: graph decode {ms / 32 * 1000:.1f} us/token ({L} layers); stamps of {'layer 1 inside k_layers' if tclass == 103 else 'layer 0 (k_attn_ffn)'}, us after the first workgroup's start (median / min / max)")
    for cls, sel in (("heads", slice(0, nh)), ("others", slice(nh, 256))):
        for k in (0, 12, 13, 14, 1, 15, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11):
            v = t[sel, k]; v = v[v >= 0]
            if len(v): print(f"  {cls:6s} {k:2d} {names[k]:58s} {np.median(v):6.2f} {v.min():6.2f} {v.max():6.2f}")
    pn = {0: "entry", 1: "staged", 3: "stage barrier passed", 2: "chain / issue done", 4: "(same)", 5: "r known / quantized", 6: "quantize round done", 7: "final barrier passed"}
    for which, nm in ((0, "FFN13 prologue"), (1, "FFN2 prologue")):
        for w, wn in ((0, "wave 0"), (8, "wave 15")):
            line = []
            for k in (0, 1, 3, 2, 5, 6, 7):
                v = pro[which][:, w + k]; v = v[v >= 0]
                if len(v): line.append(f"{pn[k]} {np.median(v):.2f}")
            if line: print(f"  {nm}, {wn}: " + " | ".join(line))
    if (rn[:, 4] > 0).any():
        print("  FFN13 run(), chain wave: " + " | ".join(f"{nm} {np.median(rn[:, k][rn[:, k] > 0]):.2f}" for k, nm in ((3, "first refill requested"), (4, "last step reduced (at the barrier)"), (5, "chains + epilogue + stores issued"))))
    if (at[:, 0] > 0).any():
        an = {0: "entry (earlier rows requested)", 7: "q flags seen", 5: "first K tile parked", 1: "scores done", 2: "exp done", 3: "sum done", 12: "V tile 0 parked", 8: "barrier passed", 9: "tile 0 walked", 4: "weighted sum done", 6: "output quantized + stored"}
        print("  attention (thread 0 of a head): " + " | ".join(f"{an[k]} {np.median(at[:, k][at[:, k] > 0]):.2f}" for k in (0, 7, 5, 1, 2, 3, 12, 8, 9, 4, 6) if (at[:, k] > 0).any()))
        if pos >= 128 and (at[:, 10] > 0).any():          # split heads: the steps of the parts' scores, part by part (workgroup % 4)
            sn = {0: "entry", 7: "swept", 10: "step 1 at its barrier", 5: "through", 11: "scored", 13: "step 2 parked", 14: "through", 15: "scored", 9: "last barrier", 1: "V requested", 2: "exp done", 3: "sum done", 4: "weighted sum done", 6: "stored"}
            for gpart in range(4):
                ap = at[gpart::4]
                print(f"    part {gpart}: " + " | ".join(f"{sn[k]} {np.median(ap[:, k][ap[:, k] > 0]):.2f}" for k in (0, 7, 10, 5, 11, 13, 14, 15, 9, 1, 2, 3, 4, 6) if (ap[:, k] > 0).any()))
    ends = np.array([r[:, 11].max() for r in rows])
    print("  launch span over 5 tokens:", " ".join(f"{e:.2f}" for e in ends))
