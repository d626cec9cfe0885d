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
