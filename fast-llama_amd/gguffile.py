"""Minimal gguf (v3) writer for LLaMA-architecture fp32 / fp16 checkpoints -- test and tooling support for the gguf reader of the
drop-in CLI (fast-llama_amd/host/model_file.cpp).  Emits exactly the key set the reference's loader understands
(src/model_loaders/gguf_loader.cpp:235-318: it rejects any other key), tensors named the llama.cpp way
(token_embd / blk.N.attn_q ... / output_norm / output), dims innermost first, data aligned to 32 bytes."""
from __future__ import annotations

import struct

import numpy as np

from . import flmfile as ff

GGUF_MAGIC = 0x46554747
T_UINT32, T_INT32, T_FLOAT32, T_STRING, T_ARRAY = 4, 5, 6, 8, 9
GGML_F32, GGML_F16 = 0, 1

_NAMES = {ff.T_TOKEN_EMBD: "token_embd.weight", ff.T_OUTPUT_NORM: "output_norm.weight", ff.T_CLASSIFIER: "output.weight",
          ff.T_INPUT_NORM: "attn_norm.weight", ff.T_ATTN_Q: "attn_q.weight", ff.T_ATTN_K: "attn_k.weight", ff.T_ATTN_V: "attn_v.weight",
          ff.T_ATTN_O: "attn_output.weight", ff.T_POST_NORM: "ffn_norm.weight", ff.T_MLP_GATE: "ffn_gate.weight",
          ff.T_MLP_UP: "ffn_up.weight", ff.T_MLP_DOWN: "ffn_down.weight"}


def _s(s: str) -> bytes:
    b = s.encode()
    return struct.pack("<q", len(b)) + b


def _kv(key: str, vtype: int, payload: bytes) -> bytes:
    return _s(key) + struct.pack("<i", vtype) + payload


def write_gguf(path, cfg: ff.FlmConfig, tok: ff.FlmTokenizer, tensors: dict, f16: bool = False, context_length: int = 1024):
    """tensors: {(kind, layer): fp32 ndarray} (the fp32-master form of synth.make_tensors)"""
    kvs = [
        _kv("general.architecture", T_STRING, _s("llama")),
        _kv("general.name", T_STRING, _s(cfg.name)),
        _kv("general.file_type", T_UINT32, struct.pack("<I", 1 if f16 else 0)),
        _kv("llama.context_length", T_UINT32, struct.pack("<I", context_length)),
        _kv("llama.embedding_length", T_UINT32, struct.pack("<I", cfg.dim)),
        _kv("llama.block_count", T_UINT32, struct.pack("<I", cfg.n_layers)),
        _kv("llama.feed_forward_length", T_UINT32, struct.pack("<I", cfg.hidden_dim)),
        _kv("llama.attention.head_count", T_UINT32, struct.pack("<I", cfg.n_heads)),
        _kv("llama.attention.head_count_kv", T_UINT32, struct.pack("<I", cfg.n_kv_heads)),
        _kv("llama.rope.dimension_count", T_UINT32, struct.pack("<I", cfg.dim // cfg.n_heads)),
        _kv("llama.attention.layer_norm_rms_epsilon", T_FLOAT32, struct.pack("<f", cfg.rms_norm_eps)),
        _kv("tokenizer.ggml.model", T_STRING, _s("llama")),
        _kv("tokenizer.ggml.tokens", T_ARRAY, struct.pack("<iq", T_STRING, len(tok.texts)) + b"".join(_s(t) for t in tok.texts)),
        _kv("tokenizer.ggml.scores", T_ARRAY, struct.pack("<iq", T_FLOAT32, len(tok.scores)) + np.asarray(tok.scores, "<f4").tobytes()),
        _kv("tokenizer.ggml.token_type", T_ARRAY, struct.pack("<iq", T_INT32, len(tok.types)) + np.asarray(tok.types, "<i4").tobytes()),
        _kv("tokenizer.ggml.bos_token_id", T_UINT32, struct.pack("<I", tok.bos)),
        _kv("tokenizer.ggml.eos_token_id", T_UINT32, struct.pack("<I", tok.eos)),
    ]
    order = [(ff.T_TOKEN_EMBD, 0)]
    for l in range(cfg.n_layers):
        order += [(k, l) for k in ff.LAYER_KINDS]
    order += [(ff.T_OUTPUT_NORM, 0), (ff.T_CLASSIFIER, 0)]
    infos, blobs, off = [], [], 0
    for kind, layer in order:
        a = np.asarray(tensors[(kind, layer)], dtype=np.float32)
        name = _NAMES[kind] if kind < 16 else f"blk.{layer}.{_NAMES[kind]}"
        half = f16 and a.ndim > 1
        data = a.astype("<f2").tobytes() if half else a.astype("<f4").tobytes()
        dims = list(a.shape[::-1])                                  # gguf: innermost dimension first
        infos.append(_s(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<q", d) for d in dims)
                     + struct.pack("<iQ", GGML_F16 if half else GGML_F32, off))
        blobs.append(data)
        off += (len(data) + 31) & ~31
    head = struct.pack("<IIqq", GGUF_MAGIC, 3, len(infos), len(kvs)) + b"".join(kvs) + b"".join(infos)
    with open(path, "wb") as f:
        f.write(head)
        f.write(b"\0" * ((-len(head)) % 32))
        for b in blobs:
            f.write(b); f.write(b"\0" * ((-len(b)) % 32))
