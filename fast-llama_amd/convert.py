"""HF LLaMA checkpoint directory -> .flm  (the reference's tools/convert_flm.py, re-done from scratch)

    python tools/convert_hf_to_flm.py -m <hf_dir> -t int8|int16|f32 [-o out.flm]

What it reproduces of the reference tool (so that its output is byte-identical on the same input, tests/test_convert.py):
  * config.json keys -> model_config block (convert_flm.py:355-384): hidden_size -> dim, intermediate_size -> hidden_dim,
    num_attention_heads / num_key_value_heads / num_hidden_layers, max_position_embeddings -> max_length, _name_or_path -> name,
    plus bos/eos/pad ids, rms_norm_eps, rope_theta when config.json carries them under the same names;
  * SentencePiece tokenizer.model -> tokenizer block (pieces, scores, NORMAL/UNKNOWN/CONTROL/UNUSED/BYTE types,
    convert_flm.py:792-833); special token ids from config.json (:930-945).  (The reference's BPE/vocab.json path does not
    run -- undefined names at :747-753 -- and is not offered here either.)
  * tensors in checkpoint order; q_proj / k_proj rows permuted from the HF rotary layout (two half-blocks per head) to the
    interleaved pairs the engine rotates (:1010-1015); every 2-D tensor except the embedding table quantized per 64 columns with
    scale = max|x| / F and C truncation (:216-243); norms and the embedding stay fp32.
Checkpoints: *.safetensors or pytorch_model*.bin / consolidated.00.pth / *.pt (torch zip pickles, via torch.load).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys

import numpy as np

from . import flmfile as ff

_LAYER_KINDS = {
    "input_layernorm": ff.T_INPUT_NORM, "self_attn.q_proj": ff.T_ATTN_Q, "self_attn.k_proj": ff.T_ATTN_K,
    "self_attn.v_proj": ff.T_ATTN_V, "self_attn.o_proj": ff.T_ATTN_O, "post_attention_layernorm": ff.T_POST_NORM,
    "mlp.gate_proj": ff.T_MLP_GATE, "mlp.up_proj": ff.T_MLP_UP, "mlp.down_proj": ff.T_MLP_DOWN,
}
_TOP_KINDS = {"model.embed_tokens.weight": ff.T_TOKEN_EMBD, "model.norm.weight": ff.T_OUTPUT_NORM, "lm_head.weight": ff.T_CLASSIFIER}
# TokenType (convert_flm.py:42-48)
_TT_UNKNOWN, _TT_NORMAL, _TT_CONTROL, _TT_BYTE, _TT_USER, _TT_UNUSED = 0, 1, 2, 3, 4, 5


def load_config(hf_dir: str, qt: int, group_size: int = 64) -> ff.FlmConfig:
    with open(os.path.join(hf_dir, "config.json")) as f:
        conf = json.load(f)
    c = ff.FlmConfig(name="", quant_type=qt, quant_group_size=group_size, max_length=0, bos_token_id=0, eos_token_id=0, pad_token_id=0,
                     rms_norm_eps=0.0, rope_theta=10000.0)
    mapping = {"_name_or_path": "name", "vocab_size": "vocab_size", "hidden_size": "dim", "intermediate_size": "hidden_dim",
               "num_attention_heads": "n_heads", "num_key_value_heads": "n_kv_heads", "num_hidden_layers": "n_layers",
               "max_position_embeddings": "max_length"}
    for k, v in conf.items():
        if k in ("bos_token_id", "eos_token_id", "pad_token_id") and isinstance(v, int):
            setattr(c, k, v)
        elif k in ("rms_norm_eps", "rope_theta") and isinstance(v, float):
            setattr(c, k, v)
        elif k in mapping and isinstance(v, type(getattr(c, mapping[k]))):
            setattr(c, mapping[k], v)
    if isinstance(conf.get("hidden_act"), str):
        c.act_type_str = conf["hidden_act"]            # stored as a string, exactly like the reference tool does
    return c


def load_spm_tokenizer(hf_dir: str, conf: dict) -> ff.FlmTokenizer:
    from sentencepiece import SentencePieceProcessor
    sp = SentencePieceProcessor(os.path.join(hf_dir, "tokenizer.model"))
    t = ff.FlmTokenizer(vocab_type=2)
    for i in range(sp.vocab_size()):
        tt = _TT_NORMAL
        if sp.is_unknown(i): tt = _TT_UNKNOWN
        if sp.is_control(i): tt = _TT_CONTROL
        if sp.is_unused(i): tt = _TT_UNUSED
        if sp.is_byte(i): tt = _TT_BYTE
        t.texts.append(sp.id_to_piece(i)); t.scores.append(float(sp.get_score(i))); t.types.append(tt)
    ids = {}
    for name in ("bos", "eos", "pad"):               # (the reference also looks at unk/sep, which its writer cannot store)
        v = conf.get(f"{name}_token_id", -1)
        if isinstance(v, int) and v >= 0:
            ids[name] = v
    t.bos, t.eos, t.pad = ids.get("bos", -1), ids.get("eos", -1), ids.get("pad", -1)
    return t


def iter_checkpoint(hf_dir: str):
    """(name, fp32 ndarray) in checkpoint order"""
    st = sorted(glob.glob(os.path.join(hf_dir, "*.safetensors")))
    if st:
        from safetensors import safe_open
        for path in st:
            with safe_open(path, framework="np") as f:
                for k in f.keys():
                    yield k, np.asarray(f.get_tensor(k), dtype=np.float32)
        return
    files = []
    for pat in ("consolidated.00.pth", "pytorch_model-*-of-*.bin", "*.pt", "pytorch_model.bin"):
        files += sorted(glob.glob(os.path.join(hf_dir, pat)))
    if not files:
        raise FileNotFoundError(f"no checkpoint (*.safetensors, pytorch_model*.bin, *.pt, consolidated.00.pth) in {hf_dir}")
    import torch
    for path in files:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        for k, v in sd.items():
            yield k, v.to(torch.float32).numpy()


def permute_qk(w: np.ndarray, n_heads: int) -> np.ndarray:
    """HF rotary layout -> interleaved pairs: within each head, row j of the first half and row j of the second half become
    rows 2j and 2j+1"""
    rows = w.shape[0]
    return w.reshape(n_heads, 2, rows // n_heads // 2, *w.shape[1:]).swapaxes(1, 2).reshape(w.shape)


def classify(name: str):
    if name in _TOP_KINDS:
        return _TOP_KINDS[name], 0
    if name.startswith("model.layers."):
        parts = name.split(".")
        layer = int(parts[2])
        key = ".".join(parts[3:])
        if key.endswith(".weight"):
            key = key[: -len(".weight")]
        if key in _LAYER_KINDS:
            return _LAYER_KINDS[key], layer
        return None, layer                       # e.g. rotary_emb.inv_freq: skipped with a note, like the reference
    raise ValueError(f"unknown tensor name: {name}")


def convert(hf_dir: str, out_path: str, out_type: str, group_size: int = 64, log=print) -> str:
    qt = {"f32": ff.QT_NONE, "int16": ff.QT_INT16, "int8": ff.QT_INT8}[out_type]
    with open(os.path.join(hf_dir, "config.json")) as f:
        conf = json.load(f)
    cfg = load_config(hf_dir, qt, group_size)
    tok = load_spm_tokenizer(hf_dir, conf)
    with open(out_path, "wb") as f:
        w = ff.FlmWriter(f)
        w.header()
        w.config(cfg)
        w.tokenizer(tok)
        for name, arr in iter_checkpoint(hf_dir):
            kind, layer = classify(name)
            if kind is None:
                log(f"skipping tensor {name} {arr.shape}")
                continue
            if kind == ff.T_ATTN_Q:
                arr = permute_qk(arr, cfg.n_heads)
            elif kind == ff.T_ATTN_K:
                arr = permute_qk(arr, cfg.n_kv_heads)
            needq = qt != ff.QT_NONE and kind != ff.T_TOKEN_EMBD and arr.ndim > 1
            if needq:
                q, s = ff.quantize(np.ascontiguousarray(arr), qt, group_size)
                w.tensor(name, kind, layer, q, s)
            else:
                w.tensor(name, kind, layer, np.ascontiguousarray(arr, dtype=np.float32))
            log(f"{name:50s} {str(tuple(arr.shape)):20s} -> {'int8' if needq and qt == ff.QT_INT8 else 'int16' if needq else 'f32'}")
    return out_path


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="convert a HF LLaMA checkpoint directory to .flm")
    ap.add_argument("-m", "--model-path", required=True)
    ap.add_argument("-t", "--out-type", required=True, choices=["f32", "int16", "int8"])
    ap.add_argument("-g", "--group-size", type=int, default=64, choices=[64], help="the engine only runs group size 64")
    ap.add_argument("-o", "--output-path")
    a = ap.parse_args(argv)
    out = a.output_path or os.path.join(a.model_path, f"{a.out_type}.flm")
    if os.path.isdir(out):
        out = os.path.join(out, f"{a.out_type}.flm")
    convert(a.model_path, out, a.out_type, a.group_size)
    print("wrote", out, file=sys.stderr)
    return 0
