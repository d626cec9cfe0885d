"""Seeded synthetic LLaMA-shaped checkpoints (there are no real model files and no network).

The same arrays feed every consumer -- the reference (through a `.flm` file written by flmfile.py),
the CPU oracle and the GPU library (through host pointers) -- so parity is checked on identical
inputs.  Shapes follow SURVEY.md section 8: `7B` = LLaMA2-7B, `1.3B` = 4 x 7B-width layers, etc.
"""
from __future__ import annotations

import numpy as np

from . import flmfile as ff

SHAPES = {
    # name: dim, hidden, layers, heads, vocab
    "7B": (4096, 11008, 32, 32, 32000),
    "1.3B": (4096, 11008, 4, 32, 55296),
    "110M": (768, 2048, 12, 12, 32000),
    "tiny": (256, 512, 2, 4, 320),      # smallest width the reference's thread-group sizing
                                         # handles (dim*dim/24576 >= n_threads, transformer.cpp:227-242)
    "tiny128": (256, 512, 2, 2, 320),   # head_size 128
    "small": (512, 1536, 2, 8, 1024),
}


def make_config(shape="tiny", qt=ff.QT_INT8, **over) -> ff.FlmConfig:
    if isinstance(shape, str):
        dim, hidden, layers, heads, vocab = SHAPES[shape]
    else:
        dim, hidden, layers, heads, vocab = shape
    c = ff.FlmConfig(name=f"synthetic-{shape}", quant_type=qt, vocab_size=vocab, dim=dim, hidden_dim=hidden,
                     n_heads=heads, n_kv_heads=heads, n_layers=layers)
    for k, v in over.items():
        setattr(c, k, v)
    return c


def _qweights(rng, rows, cols, qt, gs):
    lim = 127 if qt == ff.QT_INT8 else 5792
    dt = np.int8 if qt == ff.QT_INT8 else np.int16
    q = rng.integers(-lim, lim + 1, size=(rows, cols), dtype=dt)
    # W ~ U(-1,1) * sqrt(3)/sqrt(cols) * (0.5 + u): unit-gain rows, O(1) activations through the stack
    base = np.float32(np.sqrt(3.0) / (lim * np.sqrt(cols)))
    s = (base * (0.5 + rng.random((rows, cols // gs), dtype=np.float32))).astype(np.float32)
    return q, s


def linear_shapes(c: ff.FlmConfig):
    return {
        ff.T_ATTN_Q: (c.dim, c.dim), ff.T_ATTN_K: (c.kv_dim, c.dim), ff.T_ATTN_V: (c.kv_dim, c.dim),
        ff.T_ATTN_O: (c.dim, c.dim), ff.T_MLP_GATE: (c.hidden_dim, c.dim), ff.T_MLP_UP: (c.hidden_dim, c.dim),
        ff.T_MLP_DOWN: (c.dim, c.hidden_dim),
    }


def make_tensors(c: ff.FlmConfig, seed=1234, fp32_master=False, layers=None, share_layers=False):
    """-> {(kind, layer): fp32 ndarray | (q, scales)}.  `fp32_master=True` produces unquantized linear
    weights (the file then carries quant_type NONE and consumers quantize at load).
    `share_layers=True` reuses layer 0's arrays for every layer (big benchmark shapes: same bytes
    streamed, 1/L of the host RAM and generation time)."""
    gs = c.quant_group_size
    qt = c.quant_type if c.quant_type != ff.QT_NONE else ff.QT_INT8
    t = {}
    rng = np.random.default_rng(seed)
    t[(ff.T_TOKEN_EMBD, 0)] = rng.standard_normal((c.vocab_size, c.dim), dtype=np.float32)
    L = c.n_layers if layers is None else layers
    for l in range(L):
        if share_layers and l > 0:
            for k in ff.LAYER_KINDS:
                t[(k, l)] = t[(k, 0)]
            continue
        lr = np.random.default_rng([seed, 1000 + l])
        t[(ff.T_INPUT_NORM, l)] = (0.8 + 0.4 * lr.random(c.dim, dtype=np.float32)).astype(np.float32)
        t[(ff.T_POST_NORM, l)] = (0.8 + 0.4 * lr.random(c.dim, dtype=np.float32)).astype(np.float32)
        for kind, (r, k) in linear_shapes(c).items():
            q, s = _qweights(lr, r, k, qt, gs)
            if fp32_master:
                t[(kind, l)] = (q.astype(np.float32).reshape(r, k // gs, gs) * s[:, :, None]).reshape(r, k).astype(np.float32)
            else:
                t[(kind, l)] = (q, s)
    fr = np.random.default_rng([seed, 999])
    t[(ff.T_OUTPUT_NORM, 0)] = (0.8 + 0.4 * fr.random(c.dim, dtype=np.float32)).astype(np.float32)
    q, s = _qweights(fr, c.vocab_size, c.dim, qt, gs)
    if fp32_master:
        t[(ff.T_CLASSIFIER, 0)] = (q.astype(np.float32).reshape(c.vocab_size, c.dim // gs, gs) * s[:, :, None]).reshape(c.vocab_size, c.dim).astype(np.float32)
    else:
        t[(ff.T_CLASSIFIER, 0)] = (q, s)
    return t


def make_tokenizer(vocab_size: int, seed=7) -> ff.FlmTokenizer:
    """llama-style synthetic vocab: 0 <unk>, 1 <s>, 2 </s>, 3..258 byte tokens, then '▁', letters,
    '▁'+letter, and random lowercase pieces with descending scores."""
    texts, scores, types = ["<unk>", "<s>", "</s>"], [0.0, 0.0, 0.0], [0, 2, 2]
    for b in range(256):
        texts.append(f"<0x{b:02X}>"); scores.append(0.0); types.append(3)
    rng = np.random.default_rng(seed)
    pool = ["▁"] + [chr(c) for c in range(ord("a"), ord("z") + 1)] + ["▁" + chr(c) for c in range(ord("a"), ord("z") + 1)]
    pool += [".", ",", "!", "?", "'", "T", "I", "O", "E", "▁T", "▁I", "▁O"]
    seen = set(texts)
    for p in pool:
        if len(texts) >= vocab_size:
            break
        if p not in seen:
            seen.add(p); texts.append(p); scores.append(-float(len(texts))); types.append(1)
    letters = "etaoinshrdlucmfwyp"
    while len(texts) < vocab_size:
        n = int(rng.integers(2, 5))
        w = "".join(letters[int(i)] for i in rng.integers(0, len(letters), n))
        if rng.random() < 0.4:
            w = "▁" + w
        if w in seen:
            continue
        seen.add(w); texts.append(w); scores.append(-float(len(texts))); types.append(1)
    return ff.FlmTokenizer(texts=texts[:vocab_size], scores=scores[:vocab_size], types=types[:vocab_size])


def write_synthetic_flm(path, c: ff.FlmConfig, tensors=None, seed=1234, fp32_master=False):
    tensors = tensors if tensors is not None else make_tensors(c, seed, fp32_master)
    file_cfg = c
    if fp32_master:
        import copy
        file_cfg = copy.copy(c); file_cfg.quant_type = ff.QT_NONE
    ff.write_flm(path, file_cfg, make_tokenizer(c.vocab_size), tensors)
    return tensors


# ---------------------------------------------------------------------------------------------
# Portable synthetic checkpoint: every value is a pure function of (tensor kind, layer, element index)
# through splitmix64, so that any host language regenerates the same checkpoint from nothing (C++: 20 lines, below in the
# docstrings).  bench.py and the full-size golden test (tests/golden/make_golden_r2.py) use it; the small model fixtures of
# round 1 keep numpy's Generator.  Deviation from 8d's wording: counter-based splitmix64 instead of a sequential
# xorshift64* (numpy can evaluate 7e9 values of a counter-based generator vectorised; both are a handful of integer
# operations), and the embedding is a sum of four uniforms (Irwin-Hall, std 0.3) instead of Box-Muller (no libm call).
#   r(stream, j)  = splitmix64((stream << 32) + j),  stream = 0x5EED0001 + kind * 256 + layer (+ 0x8000 for scales)
#   int8  weight i = byte (i % 8) of r(stream, i / 8), as signed, -128 -> -127
#   int16 weight i = ((16-bit field (i % 4) of r(stream, i / 4)) % 11585) - 5792
#   scale j        = float32(0.02 / F) * (0.5f + (r(stream + 0x8000, j) >> 40) * 2^-24),  F = 127 | 5792
#   embedding i    = float32(sum of the four 16-bit fields of r(stream, i) - 131072) * float32(0.3 * sqrt(3) / 65536)
#   norm weights   = 1.0
# ---------------------------------------------------------------------------------------------
PORTABLE_SEED = 0x5EED0001
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(z):
    """z: uint64 ndarray, modified IN PLACE (wraps mod 2^64) and returned"""
    with np.errstate(over="ignore"):
        z += np.uint64(0x9E3779B97F4A7C15)
        t = z >> np.uint64(30); z ^= t; z *= np.uint64(0xBF58476D1CE4E5B9)
        np.right_shift(z, np.uint64(27), out=t); z ^= t; z *= np.uint64(0x94D049BB133111EB)
        np.right_shift(z, np.uint64(31), out=t); z ^= t
        return z


def _stream(kind, layer, scales=False):
    return np.uint64(((PORTABLE_SEED + kind * 256 + layer + (0x8000 if scales else 0)) << 32) & 0xFFFFFFFFFFFFFFFF)


def _r(kind, layer, n, scales=False, chunk=1 << 16):
    """r(stream, 0 .. n-1); small chunks keep the temporaries in cache (and out of mmap/page-fault territory)"""
    out = np.empty(n, dtype=np.uint64)
    base = _stream(kind, layer, scales)
    idx = np.arange(chunk, dtype=np.uint64)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        z = idx[: b - a] + (base + np.uint64(a))
        out[a:b] = splitmix64(z)
    return out


def portable_qweights(kind, layer, rows, cols, qt):
    n = rows * cols
    if qt == ff.QT_INT8:
        q = _r(kind, layer, (n + 7) // 8).view(np.int8)[:n]
        np.maximum(q, -127, out=q)
        F = 127.0
    else:
        h = _r(kind, layer, (n + 3) // 4).view(np.uint16)[:n]
        q = ((h.astype(np.int32) % 11585) - 5792).astype(np.int16)
        F = 5792.0
    u = (_r(kind, layer, rows * (cols // 64), scales=True) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    s = np.float32(0.02 / F) * (np.float32(0.5) + u)
    return q.reshape(rows, cols), s.astype(np.float32).reshape(rows, cols // 64)


def portable_embedding(rows, cols):
    r = _r(ff.T_TOKEN_EMBD, 0, rows * cols)
    f = r.view(np.uint16).reshape(-1, 4).astype(np.int32).sum(axis=1) - 131072
    return (f.astype(np.float32) * np.float32(0.3 * np.sqrt(3.0) / 65536.0)).reshape(rows, cols)


def iter_portable(c: ff.FlmConfig):
    """yield ((kind, layer), value) of the portable checkpoint, one tensor at a time (peak RAM = one tensor)"""
    yield (ff.T_TOKEN_EMBD, 0), portable_embedding(c.vocab_size, c.dim)
    ones = np.ones(c.dim, np.float32)
    for l in range(c.n_layers):
        yield (ff.T_INPUT_NORM, l), ones
        yield (ff.T_POST_NORM, l), ones
        for kind, (r, k) in linear_shapes(c).items():
            yield (kind, l), portable_qweights(kind, l, r, k, c.quant_type)
    yield (ff.T_OUTPUT_NORM, 0), ones
    yield (ff.T_CLASSIFIER, 0), portable_qweights(ff.T_CLASSIFIER, 0, c.vocab_size, c.dim, c.quant_type)


def make_tensors_portable(c: ff.FlmConfig):
    return dict(iter_portable(c))


def write_llama2c(path, tokpath, c: ff.FlmConfig, rng, scale=None):
    """a llama2.c checkpoint (legacy v0 header, positive vocab: classifier shared with the embedding,
    llama2c_loader.cpp:21-29,126-194) + tokenizer.bin, random fp32 weights; returns the arrays"""
    import struct
    d, hd, L, nh, V = c.dim, c.hidden_dim, c.n_layers, c.n_heads, c.vocab_size
    hs = d // nh
    g = lambda *shape, sd=1.0: (rng.standard_normal(shape) * sd).astype(np.float32)   # noqa: E731
    sd = scale if scale is not None else 1.0 / np.sqrt(d)
    w = {"emb": g(V, d, sd=0.02), "rms_att": (0.8 + 0.4 * rng.random((L, d))).astype(np.float32),
         "wq": g(L, d, d, sd=sd), "wk": g(L, d, d, sd=sd), "wv": g(L, d, d, sd=sd), "wo": g(L, d, d, sd=sd),
         "rms_ffn": (0.8 + 0.4 * rng.random((L, d))).astype(np.float32),
         "w1": g(L, hd, d, sd=sd), "w2": g(L, d, hd, sd=1.0 / np.sqrt(hd) if scale is None else scale), "w3": g(L, hd, d, sd=sd),
         "rms_final": (0.8 + 0.4 * rng.random(d)).astype(np.float32)}
    with open(path, "wb") as f:
        f.write(struct.pack("7i", d, hd, L, nh, nh, V, 256))
        for k in ("emb", "rms_att", "wq", "wk", "wv", "wo", "rms_ffn", "w1", "w2", "w3", "rms_final"):
            f.write(w[k].tobytes())
        f.write(np.zeros((256, hs // 2), np.float32).tobytes() * 2)   # legacy freq_cis_real / imag
    tok = make_tokenizer(V)
    with open(tokpath, "wb") as f:
        f.write(struct.pack("i", max(len(t.encode()) for t in tok.texts)))
        for t, s in zip(tok.texts, tok.scores):
            b = t.encode(); f.write(struct.pack("fi", float(s), len(b))); f.write(b)
    return w
