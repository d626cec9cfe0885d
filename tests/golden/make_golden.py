#!/usr/bin/env python3
"""Generate the committed golden vectors FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and oracle/_ref/libflref.so, built by
`make -C oracle ref`).  Outputs small .npz / .flm fixtures next to this script.  The fixtures hold
inputs and expected outputs only -- no reference source.  Recorded with them: the reference build
flags (oracle/_ref/BUILD_FLAGS.txt), because FMA contraction makes fp32 results build-dependent.

  ops.npz        G1 quantize (int8/int16 incl. zero group, negative-max group, +-max values)
                 G2 quant::matmul int8/int16 at (96,256,1) (96,256,3) (64,11008,1)
                 G3 rmsnorm, swiglu, softmax(columns<len), rope_v2 @ pos {0,1,37,1023} x hs {64,128},
                    batch weighted_sum with a sub-threshold weight
  attention.npz  G4 ATTN task for one head: prefill (pos 0, bs 5) then decode (pos 5, bs 1), composed
                    from the reference primitives in execute_attn's order (transformer.cpp:397-455)
  model_*.npz    G5 per-step logits + greedy ids of synthetic models through ParallelTransformer::forward
  ref_writer.flm a tiny .flm written by the reference's own tools/convert_flm.py FLFWriter
  sampler/tokenizer.npz  G6/G7 tokenizer encode/decode pairs and argmax tie behaviour
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
import oracle_py as O  # noqa: E402
from fast_llama_amd import flmfile as ff, synth  # noqa: E402

R = O.ref()
fp = O._p


def g_ops():
    out = {}
    rng = np.random.default_rng(20260928)
    for qt, name, lim, dt in ((O.QT_INT8, "i8", 127, np.int8), (O.QT_INT16, "i16", 5792, np.int16)):
        x = (rng.standard_normal(64 * 40) * 2.5).astype(np.float32)
        x[64:128] = 0.0
        x[128:192] = -np.abs(x[128:192])
        x[200] = np.abs(x[192:256]).max(); x[201] = -x[200]
        x[256:320] *= 1e-20; x[320:384] *= 1e20
        q, s = O.quantize(x, qt, lib=R)
        out[f"quant_{name}_x"] = x; out[f"quant_{name}_q"] = q; out[f"quant_{name}_s"] = s
        for (m, n, w) in ((96, 256, 1), (96, 256, 3), (64, 11008, 1)):
            mr = np.random.default_rng([qt, m, n, w])
            W = mr.integers(-lim, lim + 1, (m, n)).astype(dt); X = mr.integers(-lim, lim + 1, (w, n)).astype(dt)
            sW = mr.uniform(1e-4, 1e-3, (m, n // 64)).astype(np.float32); sX = mr.uniform(1e-3, 1e-2, (w, n // 64)).astype(np.float32)
            out[f"matmul_{name}_{m}_{n}_{w}"] = O.matmul_q(qt, W, sW, X, sX, lib=R)      # inputs are regenerated from the seed
    for n in (64, 768, 4096):
        x = (rng.standard_normal(n) * 3).astype(np.float32); w = rng.uniform(0.5, 1.5, n).astype(np.float32)
        out[f"rms_{n}_x"] = x; out[f"rms_{n}_w"] = w; out[f"rms_{n}_o"] = O.rmsnorm(x, w, lib=R)
    a = (rng.standard_normal(2048) * 4).astype(np.float32); b = rng.standard_normal(2048).astype(np.float32)
    a[:4] = [0.0, -30.0, 30.0, 1e-8]
    out["swiglu_a"] = a; out["swiglu_b"] = b; out["swiglu_o"] = O.swiglu(a, b, lib=R)
    x = (rng.standard_normal(300) * 4).astype(np.float32)
    out["softmax_x"] = x; out["softmax_cols"] = np.int32(257); out["softmax_o"] = O.softmax(x, 257, lib=R)
    for hs in (64, 128):
        for pos in (0, 1, 37, 1023):
            x = rng.standard_normal(hs).astype(np.float32)
            out[f"rope_{hs}_{pos}_x"] = x; out[f"rope_{hs}_{pos}_o"] = O.rope(x, pos, lib=R)
    V = rng.standard_normal((37, 128)).astype(np.float32); att = rng.uniform(0, 0.1, (3, 37)).astype(np.float32)
    att[1, 5] = 1e-16; att[0, 0] = 0.0
    out["wsum_V"] = V; out["wsum_att"] = att; out["wsum_o"] = O.weighted_sum(V, att, 1e-15, lib=R)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)


def ref_attention_head(kc, vc, q, k, v, pos):
    """execute_attn for one head with reference primitives (transformer.cpp:431-449)."""
    bs, hs = q.shape
    seqlen = pos + bs
    q = q.copy()
    for i in range(bs):
        kc[pos + i] = k[i]; vc[pos + i] = v[i]
        q[i] = O.rope(q[i], pos + i, lib=R)
        kc[pos + i] = O.rope(kc[pos + i], pos + i, lib=R)
    att = np.zeros((bs, seqlen), np.float32)
    kl = np.ascontiguousarray(kc[:seqlen])
    R.ref_matmul(0, fp(att), fp(kl), None, fp(q), None, seqlen, hs, bs, 64)          # att[b][t] = K[t].q[b]
    R.ref_mul(fp(att), C.c_float(1.0 / np.sqrt(np.float32(hs))), C.c_size_t(att.size))
    for i in range(bs):
        row = np.ascontiguousarray(att[i])
        R.ref_softmax(fp(row), pos + i + 1)
        row[pos + i + 1:] = 0
        att[i] = row
    return O.weighted_sum(np.ascontiguousarray(vc[:seqlen]), att, 1e-15, lib=R)


def g_attention():
    out = {}
    rng = np.random.default_rng(404)
    for hs in (64, 128):
        kc = np.zeros((1024, hs), np.float32); vc = np.zeros_like(kc)
        q = rng.standard_normal((5, hs)).astype(np.float32); k = rng.standard_normal((5, hs)).astype(np.float32); v = rng.standard_normal((5, hs)).astype(np.float32)
        o1 = ref_attention_head(kc, vc, q, k, v, 0)
        q2 = rng.standard_normal((1, hs)).astype(np.float32); k2 = rng.standard_normal((1, hs)).astype(np.float32); v2 = rng.standard_normal((1, hs)).astype(np.float32)
        o2 = ref_attention_head(kc, vc, q2, k2, v2, 5)
        for nm, a in (("q", q), ("k", k), ("v", v), ("o", o1), ("q2", q2), ("k2", k2), ("v2", v2), ("o2", o2), ("kc", kc[:6].copy()), ("vc", vc[:6].copy())):
            out[f"hs{hs}_{nm}"] = a
    np.savez_compressed(os.path.join(HERE, "attention.npz"), **out)


def g_model(shape, qt, seed, fp32_master=False, nprompt=8, ndec=16, threads=2):
    cfg = synth.make_config(shape, qt)
    path = f"/tmp/golden-{shape}-{qt}-{int(fp32_master)}.flm"
    tensors = synth.write_synthetic_flm(path, cfg, seed=seed, fp32_master=fp32_master)
    m = O.RefModel(path, qt, threads=threads)
    V = cfg.vocab_size
    prompt = np.array([1] + [int(x) for x in (np.arange(1, nprompt) * 7919) % V], dtype=np.int32)
    logits, ids, margin = [], [], []
    pos, cur = 0, prompt
    for _ in range(ndec + 1):
        l = m.forward(cur, pos)
        srt = np.sort(l)
        logits.append(l); ids.append(int(np.argmax(l))); margin.append(float((srt[-1] - srt[-2]) / abs(srt[-1])))
        pos += len(cur); cur = np.array([ids[-1]], np.int32)
    # checksum of the regenerated weights guards against PRNG drift between numpy versions
    chk = 0
    for key in sorted(tensors):
        v = tensors[key]
        arrs = v if isinstance(v, tuple) else (v,)
        for a in arrs:
            chk = (chk * 1000003 + int(np.frombuffer(np.ascontiguousarray(a).tobytes(), dtype=np.uint8).astype(np.uint64).sum())) % (1 << 61)
    name = f"model_{shape}_{'int8' if qt == O.QT_INT8 else 'int16'}{'_f32master' if fp32_master else ''}.npz"
    np.savez_compressed(os.path.join(HERE, name), seed=np.int64(seed), prompt=prompt, logits=np.stack(logits), ids=np.array(ids, np.int32),
                        margin=np.array(margin, np.float32), weights_checksum=np.uint64(chk))
    os.remove(path)


def g_ref_writer():
    """a tiny .flm produced by the reference's own Python writer: pins byte-compatibility of flmfile.py."""
    sys.path.insert(0, "/root/reference/tools")
    import io
    import convert_flm as cf
    cfg = synth.make_config((64, 128, 1, 1, 264), ff.QT_INT8)
    tok = synth.make_tokenizer(cfg.vocab_size)
    tensors = synth.make_tensors(cfg, seed=3)
    path = os.path.join(HERE, "ref_writer.flm")
    w = cf.FLFWriter(path, True)
    cf.ModelConverter._dump_file_header(None, w)
    mc = cf.ModelConfig(name=cfg.name, quant_type=cf.QuantType.INT8, vocab_size=cfg.vocab_size, dim=cfg.dim, hidden_dim=cfg.hidden_dim,
                        n_heads=cfg.n_heads, n_kv_heads=cfg.n_kv_heads, n_layers=cfg.n_layers, max_length=cfg.max_length,
                        bos_token_id=1, eos_token_id=2, pad_token_id=0, rms_norm_eps=1e-5, rope_theta=10000.0, quant_group_size=64)
    w.dump_block("model_config", mc.serialize_as_flf(True), cf.BlockType.DICT)
    t = cf.Tokenizer()
    t.vocab = cf.Vocab(cf.VocabType.SPM, tok.texts, tok.scores, tok.types) if hasattr(cf, "Vocab") else None
    t.special_tokens = {"bos_token_id": 1, "eos_token_id": 2, "pad_token_id": 0}
    w.dump_block("tokenizer", t.serialize_as_flf(True), cf.BlockType.DICT)
    order = [(ff.T_TOKEN_EMBD, 0)] + [(k, 0) for k in ff.LAYER_KINDS] + [(ff.T_OUTPUT_NORM, 0), (ff.T_CLASSIFIER, 0)]
    for kind, layer in order:
        v = tensors[(kind, layer)]
        nm = ff.KIND_NAMES[kind] if kind < 16 else f"model.layers.{layer}.{ff.KIND_NAMES[kind]}"
        if isinstance(v, tuple):
            w.dump_named_tensor(nm, v[0], v[1], cf.TensorType(kind), layer)
        else:
            w.dump_named_tensor(nm, np.asarray(v, np.float32), None, cf.TensorType(kind), layer)
    w.ofile.close()


def g_tok_sampler():
    cfg = synth.make_config("tiny", ff.QT_INT8)
    path = "/tmp/golden-tok.flm"
    synth.write_synthetic_flm(path, cfg, seed=1)
    m = O.RefModel(path, O.QT_INT8, threads=1)
    texts = ["the shape of it", "That was a long long story.", "hello", " a", "tea time!", "résumé ünï", "I'm on it, OK?"]
    enc = {}
    buf = np.zeros(4096, np.int32); cbuf = C.create_string_buffer(65536)
    for i, t in enumerate(texts):
        n = R.ref_model_encode(m.h, t.encode(), fp(buf), 4096)
        ids = buf[:n].copy()
        R.ref_model_decode(m.h, fp(ids), n, cbuf, 65536)
        enc[f"text_{i}"] = np.frombuffer(t.encode(), dtype=np.uint8); enc[f"ids_{i}"] = ids
        enc[f"dec_{i}"] = np.frombuffer(cbuf.value, dtype=np.uint8)
    # sampler: argmax tie -> lowest index; seed-0 top-p == most probable token
    lg = np.zeros(cfg.vocab_size, np.float32); lg[[7, 3, 200]] = 5.0
    enc["tie_logits_idx"] = np.array([7, 3, 200], np.int32)
    enc["tie_argmax"] = np.int32(R.ref_model_sample(m.h, fp(lg.copy()), C.c_float(0.0), C.c_float(0.9)))
    lg2 = np.random.default_rng(9).standard_normal(cfg.vocab_size).astype(np.float32)
    enc["rand_logits"] = lg2
    enc["rand_topp"] = np.int32(R.ref_model_sample(m.h, fp(lg2.copy()), C.c_float(1.0), C.c_float(0.9)))
    enc["rand_t0"] = np.int32(R.ref_model_sample(m.h, fp(lg2.copy()), C.c_float(0.0), C.c_float(0.9)))
    np.savez_compressed(os.path.join(HERE, "tokenizer_sampler.npz"), **enc)
    os.remove(path)


CLI_CASES = [   # (name, shape, qt, seed, extra CLI args)
    ("greedy_int8", "tiny", ff.QT_INT8, 21, ["-q", "int8", "-t", "0", "-n", "24", "-i", "the shape of it"]),
    ("sample_int16", "tiny", ff.QT_INT16, 22, ["-q", "int16", "-n", "12", "-i", "tea time!"]),
    ("tiny128_int8", "tiny128", ff.QT_INT8, 23, ["-q", "int8", "-t", "0", "-n", "8", "-i", "Oliver lived in a small village."]),
    # (the reference CLI overruns its 128-token batch buffers on longer prompts, e.g. its own default prompt with this vocabulary)
    ("encode", "tiny", ff.QT_INT8, 21, ["-e", "I'm on it, OK?"]),
    ("decode", "tiny", ff.QT_INT8, 21, ["-d", "[1, 279, 264, 260, 305]"]),
]

GGUF_CASES = [  # (name, shape, seed, f16, extra CLI args): fp32 masters written as gguf by fast_llama_amd/gguffile.py
    ("gguf_f32_int8", "tiny", 33, False, ["-f", "gguf", "-q", "int8", "-t", "0", "-n", "16", "-i", "the shape of it"]),
    ("gguf_f32_int16", "tiny128", 34, False, ["-f", "gguf", "-q", "int16", "-t", "0", "-n", "8", "-i", "tea time!"]),
]


def g_cli():
    """transcripts of the reference CLI (oracle/_ref/main) on synthetic .flm files; the summary line's timing
    fields are stripped by the test, everything else must match byte for byte"""
    import subprocess
    out = {}
    for name, shape, qt, seed, extra in CLI_CASES:
        cfg = synth.make_config(shape, qt)
        path = f"/tmp/golden-cli-{name}.flm"
        synth.write_synthetic_flm(path, cfg, seed=seed)
        r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "main"), "-c", path, "-j", "1", *extra], capture_output=True, check=True)
        lines = [l for l in r.stdout.split(b"\n") if not l.startswith(b"DEBUG:")]
        out[name] = np.frombuffer(b"\n".join(lines), dtype=np.uint8)
        os.remove(path)
    from fast_llama_amd import gguffile
    for name, shape, seed, f16, extra in GGUF_CASES:
        cfg = synth.make_config(shape, ff.QT_NONE)
        path = f"/tmp/golden-cli-{name}.gguf"
        gguffile.write_gguf(path, cfg, synth.make_tokenizer(cfg.vocab_size), synth.make_tensors(cfg, seed=seed, fp32_master=True), f16=f16)
        r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "main"), "-c", path, "-j", "1", *extra], capture_output=True, check=True)
        lines = [l for l in r.stdout.split(b"\n") if not l.startswith(b"DEBUG:")]
        out[name] = np.frombuffer(b"\n".join(lines), dtype=np.uint8)
        os.remove(path)
    np.savez_compressed(os.path.join(HERE, "cli_transcripts.npz"), **out)


if __name__ == "__main__":
    assert O.have_ref(), "build oracle/_ref first: make -C oracle ref"
    g_ops(); g_attention()
    g_model("tiny", O.QT_INT8, 1234); g_model("tiny", O.QT_INT16, 1234); g_model("tiny128", O.QT_INT8, 4321)
    g_model("tiny", O.QT_INT8, 1234, fp32_master=True); g_model("small", O.QT_INT8, 99, threads=4)
    g_tok_sampler()
    g_cli()
    try:
        g_ref_writer()
    except Exception as e:   # the reference converter is Python-version sensitive; report, do not hide
        print("ref_writer.flm NOT generated:", repr(e))
    with open(os.path.join(HERE, "BUILD_FLAGS.txt"), "w") as f:
        f.write(open(os.path.join(ROOT, "oracle", "_ref", "BUILD_FLAGS.txt")).read())
        f.write(f"numpy {np.__version__}\n")
    print("golden vectors written to", HERE)
