"""random small model shapes (odd group counts, hs 32/64/128, int8/int16) against the CPU oracle, logits bit for bit: python tools/fuzz_shapes.py [n] [seed] [long]
(long = 1: prompts of 100..700 tokens -- the batched prompt kernels, then decode steps with the heads split over workgroups and QKV in the
attention's launch where the shape allows it)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
import oracle_py as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
long_ctx = len(sys.argv) > 3 and int(sys.argv[3]) != 0
bad = 0
for it in range(n):
    hs = int(rng.choice([32, 64, 128])); heads = int(rng.integers(1, 9))
    dim = hs * heads
    if dim % 64: dim = (dim + 63) // 64 * 64; heads = dim // hs
    hidden = int(rng.integers(1, 24)) * 64
    vocab = int(rng.integers(300, 1500))
    qt = ff.QT_INT8 if rng.random() < 0.6 else ff.QT_INT16
    cfg = synth.make_config("tiny", qt, dim=dim, hidden_dim=hidden, n_heads=heads, n_kv_heads=heads, n_layers=2, vocab_size=vocab)
    tensors = synth.make_tensors(cfg, seed=100 + it)
    om = O.OracleModel(cfg, tensors)
    ctx = capi.Ctx(capi.desc_from_config(cfg)); ctx.upload_all(tensors)
    npr = int(rng.integers(100, 700)) if long_ctx else int(rng.integers(1, 40))
    prompt = np.array([1] + [int(x) for x in rng.integers(2, vocab, npr - 1)], dtype=np.int32) if npr > 1 else np.array([1], np.int32)
    lg = ctx.forward(prompt, 0); lo = om.forward(prompt, 0)
    ok = np.array_equal(lg.view(np.uint32), lo.view(np.uint32))
    cur, pos = int(np.argmax(lo)), len(prompt)
    for _ in range(3):
        t = np.array([cur], np.int32); lg = ctx.forward(t, pos); lo = om.forward(t, pos)
        ok = ok and np.array_equal(lg.view(np.uint32), lo.view(np.uint32)); cur = int(np.argmax(lo)); pos += 1
    # ... and the device-resident greedy loop from there (the one-launch token where the shape takes it: embedding row, classifier and argmax inside the launch, odd vocabulary sizes)
    ng = min(11, cfg.max_seq_len - pos) if hasattr(cfg, "max_seq_len") else 11
    want, oc = [], cur
    for k in range(ng):
        oc = int(np.argmax(om.forward(np.array([oc], np.int32), pos + k))); want.append(oc)
    got = list(ctx.decode_greedy(cur, pos, ng)) if ng > 0 else []
    ok = ok and got == want and ctx.query("fallback") == 0
    print(f"dim {dim} hidden {hidden} heads {heads} hs {hs} vocab {vocab} {'int8' if qt == ff.QT_INT8 else 'int16'} prompt {npr}: {'ok' if ok else 'MISMATCH'} (one-launch token: {bool(ctx.query('token_path') & 1024)})", flush=True)
    bad += (not ok)
    ctx.close()
print("fuzz:", "FAILED" if bad else "ok")
