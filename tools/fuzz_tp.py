"""random small model shapes as tensor-parallel ranks (threads, CU masks) on one GPU against the CPU oracle, logits and greedy ids bit for bit:
python tools/fuzz_tp.py [n] [seed]      (shapes a rank split cannot take -- rows not 64-aligned per rank, heads not divisible -- are skipped)"""
import sys, os, threading, faulthandler
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
import oracle_py as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)

def run(ctxs, fn):
    out = [None] * len(ctxs); err = [None] * len(ctxs)
    def work(r):
        try: out[r] = fn(ctxs[r])
        except Exception as e: err[r] = e          # noqa: BLE001
    th = [threading.Thread(target=work, args=(r,)) for r in range(len(ctxs))]
    [t.start() for t in th]; [t.join(90) for t in th]
    if any(t.is_alive() for t in th):          # (a hung rank's thread never returns: say which shape and leave hard, or the interpreter waits for it for ever)
        print("A RANK HUNG:", current, flush=True); os._exit(3)
    for e in err:
        if e is not None: raise e
    return out

done = skipped = bad = 0
while done < n:
    world = int(rng.choice([2, 4]))
    hs = int(rng.choice([32, 64, 128])); heads = world * int(rng.integers(1, 5))
    dim = hs * heads
    if (dim // world) % 64: skipped += 1; continue
    hidden = world * 64 * int(rng.integers(1, 7))
    vocab = world * 64 * int(rng.integers(2, 9))
    qt = ff.QT_INT8 if rng.random() < 0.6 else ff.QT_INT16
    cfg = synth.make_config("tiny", qt, dim=dim, hidden_dim=hidden, n_heads=heads, n_kv_heads=heads, n_layers=2, vocab_size=vocab)
    tensors = synth.make_tensors(cfg, seed=300 + done)
    om = O.OracleModel(cfg, tensors)
    npr = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(130, 300))]))
    prompt = np.array([1] + [int(x) for x in rng.integers(2, vocab, npr - 1)], dtype=np.int32) if npr > 1 else np.array([1], np.int32)
    want = [om.forward(prompt, 0)]
    cur, pos = int(np.argmax(want[0])), len(prompt)
    for _ in range(4):
        want.append(om.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(want[-1])); pos += 1
    ids_want = [int(np.argmax(w)) for w in want]
    current = f"world {world} dim {dim} hidden {hidden} heads {heads} hs {hs} vocab {vocab} {'int8' if qt == ff.QT_INT8 else 'int16'} prompt {npr}"
    print("...", current, flush=True)
    faulthandler.dump_traceback_later(150, exit=True)          # (a shape takes seconds: a hang anywhere -- create, upload, regroup, a rank -- leaves every thread's Python stack on stderr)
    try:
        ctxs = [capi.Ctx(capi.desc_from_config(cfg), device=0, rank=r, world=world, comm_id=None) for r in range(world)]
    except capi.FlmError as e:
        skipped += 1; continue
    for c in ctxs: c.upload_all(tensors)
    fa, fn_ = int(rng.choice([0, 1, 2])), int(rng.choice([0, 1]))
    tpl, gr = int(rng.choice([0, 1, 1])), int(rng.choice([0, 1, 1]))          # (round 6: the rank-spanning launch, on granules or on flag rounds)
    current += f" tp_fuse_attn {fa} tp_fuse_ffn {fn_} tp_fuse_layers {tpl} gr_edges {gr}"; print("   ", current, flush=True)
    for c in ctxs:
        c.set_option("cu_parts", world); c.set_option("tp_fuse_attn", fa); c.set_option("tp_fuse_ffn", fn_); c.set_option("tp_fuse_layers", tpl); c.set_option("gr_edges", gr)
    capi.Ctx.regroup(ctxs)
    def rank_main(c):
        lg = [c.forward(prompt, 0)]
        cur, pos = int(np.argmax(lg[0])), len(prompt)
        lg.append(c.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(lg[-1])); pos += 1
        return lg, [int(x) for x in c.decode_greedy(cur, pos, 3)]
    ok = True
    res = run(ctxs, rank_main)
    act, gra = ctxs[0].query("tp_layers_active"), ctxs[0].query("gr_active")
    for r, (lg, ids) in enumerate(res):
        ok &= np.array_equal(lg[0].view(np.uint32), want[0].view(np.uint32)) and np.array_equal(lg[1].view(np.uint32), want[1].view(np.uint32)) and ids == ids_want[2:5]
    for c in ctxs: c.close()
    print(f"world {world} dim {dim} hidden {hidden} heads {heads} hs {hs} vocab {vocab} {'int8' if qt == ff.QT_INT8 else 'int16'} prompt {npr} tp_fuse_attn {fa} tp_fuse_ffn {fn_} tp_fuse_layers {tpl} gr_edges {gr} (rank-spanning launch ran: {act}, on granules: {gra}): {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1; done += 1
print(f"fuzz_tp: {'ok' if bad == 0 else f'{bad} MISMATCHES'} ({done} shapes, {skipped} skipped)")
sys.exit(1 if bad else 0)
