"""long greedy decodes of a 7B-width model sharded over CU-masked ranks on ONE GPU, the rank-spanning launch on granules and on flag rounds against a single-GPU context:
every rank's ids must be the single GPU's (a stale cross-rank read, a torn granule or an epoch slip changes them).  python tools/tp_soak.py [layers] [tokens] [reps]"""
import sys, os, threading
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ntok = int(sys.argv[2]) if len(sys.argv) > 2 else 900
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
bad = 0
for qt, qn in ((ff.QT_INT8, "int8"), (ff.QT_INT16, "int16")):
    cfg = synth.make_config("7B", qt); cfg.n_layers = L
    tensors = synth.make_tensors(cfg, seed=1, share_layers=True)
    prompt = (np.arange(1, 9, dtype=np.int64) * 7919 % cfg.vocab_size).astype(np.int32)
    one = capi.Ctx(capi.desc_from_config(cfg)); one.upload_all(tensors)
    first = one.forward_argmax(prompt, 0)
    ref = [int(first)] + [int(x) for x in one.decode_greedy(first, len(prompt), ntok)]
    one.close()
    for world in (2, 4, 8):
        ctxs = [capi.Ctx(capi.desc_from_config(cfg), device=0, rank=r, world=world, comm_id=None) for r in range(world)]
        for c in ctxs: c.upload_all(tensors); c.set_option("cu_parts", world)
        for gr in (1, 0):
            for c in ctxs: c.set_option("gr_edges", gr); c.set_option("tp_fuse_layers", 1)
            capi.Ctx.regroup(ctxs)
            for rep in range(reps):
                out = [None] * world
                def work(r):
                    c = ctxs[r]; c.reset_kv(); f = c.forward_argmax(prompt, 0); out[r] = [int(f)] + [int(x) for x in c.decode_greedy(f, len(prompt), ntok)]
                th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
                [t.start() for t in th]; [t.join(120) for t in th]
                if any(t.is_alive() for t in th): print(f"A RANK HUNG: {qn} world {world} {'granules' if gr else 'flag rounds'} rep {rep}", flush=True); os._exit(3)
                ok = all(o == ref for o in out)
                bad += 0 if ok else 1
                print(f"{qn} world {world} {'granules' if gr else 'flag rounds'} rep {rep}: {ntok} tokens on every rank {'identical to one GPU' if ok else 'MISMATCH'} (rank-spanning launch {ctxs[0].query('tp_layers_active')}, granules {ctxs[0].query('gr_active')}, fallback {ctxs[0].query('fallback')})", flush=True)
        for c in ctxs: c.close()
print("tp_soak:", "ok" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
