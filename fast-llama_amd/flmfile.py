"""`.flm` model-file writer and reader (numpy), written from the on-disk format.

Format (what the reference's C++ loader accepts, /root/reference paths):
  file header   : u32 0xFA571AEA, u8 v1, u8 v2, u16 v3            (src/model_loaders/flm_loader.cpp:115-123)
  blocks        : BASE_ITEM (scalar + name packed in the header), DICT / STRING / TENSOR "medium"
                  blocks with a 16-byte fixed header, optional header_data, name, head padding so
                  the payload starts aligned, payload, tail padding            (flm_loader.cpp:132-178)
  tensor header : u32 shape[4], u16 tensor_type, u16 layer_id, u32 scales_size; payload = values
                  then fp32 scales                        (flm_loader.cpp:165-177, tensor.cpp:78-102)
  model_config  : DICT of BASE_ITEM / STRING blocks, keys as read by load_config (flm_loader.cpp:390-442)
  tokenizer     : DICT with the blob read by load_tokenizer               (flm_loader.cpp:444-491)

Byte-compatibility with the reference's own writer (tools/convert_flm.py FLFWriter) is pinned by
tests/test_flmfile.py against a golden file generated in the build container.
"""
from __future__ import annotations

import io
import struct
from dataclasses import dataclass, field

import numpy as np

FLM_TAG = 0xFA571AEA
GROUP = 64

# block / data type ids
BT_BASE_ITEM, BT_DICT, BT_TENSOR, BT_ARRAY, BT_STRING, BT_STRING_ARRAY = range(6)
DT_NONE, DT_INT8, DT_INT16, DT_INT32, DT_INT64 = 0, 1, 2, 3, 4
DT_FLOAT32, DT_FLOAT64 = 11, 12
_NP2DT = {"int8": DT_INT8, "int16": DT_INT16, "int32": DT_INT32, "float32": DT_FLOAT32}
_DT2NP = {DT_INT8: np.int8, DT_INT16: np.int16, DT_INT32: np.int32, DT_FLOAT32: np.float32}

# tensor kinds (TensorType)
T_TOKEN_EMBD, T_OUTPUT_NORM, T_CLASSIFIER = 1, 2, 3
T_INPUT_NORM, T_ATTN_Q, T_ATTN_K, T_ATTN_V, T_ATTN_O = 17, 18, 19, 20, 21
T_MLP_GATE, T_MLP_UP, T_MLP_DOWN, T_POST_NORM = 22, 23, 24, 25
LAYER_KINDS = (T_INPUT_NORM, T_ATTN_Q, T_ATTN_K, T_ATTN_V, T_ATTN_O, T_MLP_GATE, T_MLP_UP, T_MLP_DOWN, T_POST_NORM)
KIND_NAMES = {
    T_TOKEN_EMBD: "model.embed_tokens.weight", T_OUTPUT_NORM: "model.norm.weight", T_CLASSIFIER: "lm_head.weight",
    T_INPUT_NORM: "input_layernorm.weight", T_ATTN_Q: "self_attn.q_proj.weight", T_ATTN_K: "self_attn.k_proj.weight",
    T_ATTN_V: "self_attn.v_proj.weight", T_ATTN_O: "self_attn.o_proj.weight", T_MLP_GATE: "mlp.gate_proj.weight",
    T_MLP_UP: "mlp.up_proj.weight", T_MLP_DOWN: "mlp.down_proj.weight", T_POST_NORM: "post_attention_layernorm.weight",
}

QT_NONE, QT_INT16, QT_INT8 = 0, 1, 2
QFACTOR = {QT_INT8: 127.0, QT_INT16: 5792.0}
QDTYPE = {QT_INT8: np.int8, QT_INT16: np.int16}


def quantize(x: np.ndarray, qt: int, gs: int = GROUP):
    """Group quantizer, same arithmetic as quant::quantize (src/blas/quant_operators.cpp:26-47):
    scale = max|x|/F in fp32, q = trunc(x/scale) in fp32; all-zero group -> q = 0, scale = 0."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    g = x.reshape(-1, gs)
    s = (np.abs(g).max(axis=1) / np.float32(QFACTOR[qt])).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = g / s[:, None]
    t = np.where(np.isfinite(t), np.trunc(t), 0.0)
    q = t.astype(QDTYPE[qt]).reshape(x.shape)
    return q, s.reshape(x.shape[:-1] + (x.shape[-1] // gs,))


@dataclass
class FlmConfig:
    name: str = "synthetic"
    model_type: int = 1          # LLAMA
    act_type: int = 2            # SWIGLU
    quant_type: int = QT_INT8
    vocab_size: int = 0
    dim: int = 0
    hidden_dim: int = 0
    n_heads: int = 0
    n_kv_heads: int = 0
    n_layers: int = 0
    max_length: int = 1024
    bos_token_id: int = 1
    eos_token_id: int = 2
    pad_token_id: int = 0
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    quant_group_size: int = GROUP
    act_type_str: str | None = None   # the reference converter stores config.json's "hidden_act" string under act_type (convert_flm.py:369-383)

    @property
    def head_size(self):
        return self.dim // self.n_heads

    @property
    def kv_dim(self):
        return self.head_size * self.n_kv_heads


@dataclass
class FlmTokenizer:
    texts: list = field(default_factory=list)     # index text per token (utf-8 str)
    scores: list = field(default_factory=list)
    types: list = field(default_factory=list)
    vocab_type: int = 2                            # SPM
    bos: int = 1
    eos: int = 2
    pad: int = 0


def _pad_to(b: bytes, a: int) -> bytes:
    r = (-len(b)) % a
    return b + b"\0" * r


class FlmWriter:
    def __init__(self, f):
        self.f = f
        self.pos = 0

    def _w(self, b: bytes):
        self.f.write(b)
        self.pos += len(b)

    def header(self, version=(1, 0, 0)):
        self._w(struct.pack("<IBBH", FLM_TAG, *version))

    @staticmethod
    def base_item(name: str, fmt: str, dt: int, value) -> bytes:
        data = struct.pack("<" + fmt, value)
        nm = name.encode() + b"\0"
        small = len(data) <= 4
        item = (8 if small else 16) + len(nm)
        hs = (item + 7) & ~7
        out = struct.pack("<BBBB", BT_BASE_ITEM, dt, hs, len(data))
        if not small:
            out += struct.pack("<i", 0)
        out += data + b"\0" * ((4 if small else 8) - len(data)) + nm
        return out + b"\0" * (hs - item)

    def _block_bytes(self, pos, name, data_len, bt, dt, align, header_data=b""):
        nm = name.encode() + b"\0" if name else b""
        hd = _pad_to(header_data, 8) if header_data else b""
        name_off = 16 + len(hd)
        hs = name_off + len(nm)
        head_pad = (-(pos + hs)) % align
        hs += head_pad
        assert hs < 256, "block header too large"
        block = (hs + data_len + align - 1) & ~(align - 1)
        tail = block - hs - data_len
        h = struct.pack("<BBBBBBHQ", bt, dt, hs, len(hd), name_off, len(name.encode()) if name else 0, tail, data_len)
        return h + hd + nm + b"\0" * head_pad, tail

    def block(self, name, data: bytes, bt, dt=DT_NONE, align=8, header_data=b""):
        h, tail = self._block_bytes(self.pos, name, len(data), bt, dt, align, header_data)
        self._w(h)
        self._w(data)
        self._w(b"\0" * tail)

    @staticmethod
    def string_block(name, s: str) -> bytes:
        w = FlmWriter(io.BytesIO())
        w.block(name, s.encode() + b"\0", BT_STRING, DT_INT8, 8)
        return w.f.getvalue()

    def config(self, c: FlmConfig):
        b = self.string_block("name", c.name)   # nested blocks are laid out from offset 0 of the dict payload
        for k in ("model_type", "act_type", "quant_type", "vocab_size", "dim", "hidden_dim", "n_heads", "n_kv_heads",
                  "n_layers", "max_length", "bos_token_id", "eos_token_id", "pad_token_id"):
            if k == "act_type" and c.act_type_str is not None:
                b += self.string_block(k, c.act_type_str)
            else:
                b += self.base_item(k, "i", DT_INT32, int(getattr(c, k)))
        b += self.base_item("rms_norm_eps", "f", DT_FLOAT32, float(c.rms_norm_eps))
        b += self.base_item("rope_theta", "f", DT_FLOAT32, float(c.rope_theta))
        b += self.base_item("quant_group_size", "i", DT_INT32, int(c.quant_group_size))
        self.block("model_config", b, BT_DICT)

    def tokenizer(self, t: FlmTokenizer):
        conn = "▁"
        tok = b""
        txt = b""
        for text, score, ty in zip(t.texts, t.scores, t.types):
            ip = len(txt)
            txt += _pad_to(text.encode() + b"\0", 8)
            if text.startswith(conn):
                sp = len(txt)
                txt += _pad_to((" " + text[len(conn):]).encode() + b"\0", 8)
            else:
                sp = ip
            tok += struct.pack("<iiif", ip, sp, int(ty), float(score))
        conn_pos = len(txt)
        txt += _pad_to(conn.encode() + b"\0", 8)
        # SpecialTokenType: NONE 0, BOS 1, EOS 2, PAD 3, MAX 8 (convert_flm.py:50-57)
        special = [-1] * 8
        special[1], special[2], special[3] = t.bos, t.eos, t.pad
        blob = struct.pack("<II", t.vocab_type, conn_pos) + struct.pack("<8i", *special)
        blob += struct.pack("<II", len(t.texts), len(txt)) + tok + txt
        self.block("tokenizer", blob, BT_DICT)

    def tensor(self, name, kind, layer, values: np.ndarray, scales: np.ndarray | None = None):
        shape = list(values.shape) + [0] * (4 - values.ndim)
        hd = struct.pack("<4IHHI", *shape, kind, layer, 0 if scales is None else scales.size)
        data_len = values.nbytes + (0 if scales is None else scales.nbytes)
        h, tail = self._block_bytes(self.pos, name, data_len, BT_TENSOR, _NP2DT[str(values.dtype)], 64, hd)
        self._w(h)
        self._w(np.ascontiguousarray(values).tobytes())
        if scales is not None:
            self._w(np.ascontiguousarray(scales, dtype=np.float32).tobytes())
        self._w(b"\0" * tail)


def write_flm(path, cfg: FlmConfig, tok: FlmTokenizer, tensors: dict):
    """tensors: {(kind, layer): values} for fp32, or {(kind, layer): (q, scales)} for quantized.
    Layer tensors are written layer 0 first (the reference reader requires it, flm_loader.cpp:529-548)."""
    with open(path, "wb") as f:
        w = FlmWriter(f)
        w.header()
        w.config(cfg)
        w.tokenizer(tok)
        order = [(T_TOKEN_EMBD, 0)]
        for l in range(cfg.n_layers):
            order += [(k, l) for k in LAYER_KINDS]
        order += [(T_OUTPUT_NORM, 0), (T_CLASSIFIER, 0)]
        for key in order:
            if key not in tensors:
                continue
            v = tensors[key]
            kind, layer = key
            nm = KIND_NAMES[kind] if kind < 16 else f"model.layers.{layer}.{KIND_NAMES[kind]}"
            if isinstance(v, tuple):
                w.tensor(nm, kind, layer, v[0], v[1])
            else:
                w.tensor(nm, kind, layer, np.asarray(v, dtype=np.float32))


def read_flm(path):
    """-> (FlmConfig, FlmTokenizer, {(kind, layer): ndarray | (q, scales)}).  Reader of this repo's
    Python side (tests, bench); the product's reader is the C++ one in fast-llama_amd/host."""
    buf = np.fromfile(path, dtype=np.uint8)
    mv = memoryview(buf)
    tag, v1, v2, v3 = struct.unpack_from("<IBBH", mv, 0)
    if tag != FLM_TAG:
        raise ValueError("not an .flm file")
    cfg, tok, tensors = FlmConfig(), FlmTokenizer(), {}
    pos = 8

    def parse_block(p):
        bt, dt, hs, hds = struct.unpack_from("<BBBB", mv, p)
        if bt == BT_BASE_ITEM:
            small = hds <= 4
            name = bytes(mv[p + (8 if small else 16): p + hs]).split(b"\0", 1)[0].decode()
            return dict(bt=bt, dt=dt, hs=hs, size=hs, name=name, p=p, hds=hds)
        name_off, name_size, tail, dsz = struct.unpack_from("<BBHQ", mv, p + 4)
        name = bytes(mv[p + name_off: p + name_off + name_size]).decode()
        return dict(bt=bt, dt=dt, hs=hs, size=hs + dsz + tail, name=name, p=p, dsz=dsz, hds=hds)

    def item_value(b):
        p = b["p"]
        small = b["hds"] <= 4
        if b["dt"] in (DT_FLOAT32,):
            return struct.unpack_from("<f", mv, p + 4)[0]
        if b["dt"] == DT_FLOAT64:
            return struct.unpack_from("<d", mv, p + 8)[0]
        return struct.unpack_from("<i" if small else "<q", mv, p + (4 if small else 8))[0]

    while pos < len(buf):
        b = parse_block(pos)
        if b["name"] == "model_config":
            q, end = pos + b["hs"], pos + b["hs"] + b["dsz"]
            while q < end:
                c = parse_block(q)
                if c["bt"] == BT_BASE_ITEM:
                    if hasattr(cfg, c["name"]):
                        v = item_value(c)
                        setattr(cfg, c["name"], type(getattr(cfg, c["name"]))(v))
                elif c["bt"] == BT_STRING and c["name"] == "name":
                    cfg.name = bytes(mv[q + c["hs"]: q + c["hs"] + c["dsz"]]).split(b"\0", 1)[0].decode()
                q += c["size"]
            if cfg.n_kv_heads < 1:
                cfg.n_kv_heads = cfg.n_heads
        elif b["name"] == "tokenizer":
            q = pos + b["hs"]
            vt, conn_pos = struct.unpack_from("<II", mv, q)
            special = struct.unpack_from("<8i", mv, q + 8)
            n, tsz = struct.unpack_from("<II", mv, q + 40)
            items = np.frombuffer(mv, dtype=np.dtype([("ip", "<u4"), ("sp", "<u4"), ("ty", "<u4"), ("sc", "<f4")]), count=n, offset=q + 48)
            tb = q + 48 + 16 * n
            text = bytes(mv[tb: tb + tsz])
            tok = FlmTokenizer(vocab_type=vt, bos=special[1], eos=special[2], pad=special[3])
            for it in items:
                tok.texts.append(text[it["ip"]:].split(b"\0", 1)[0].decode(errors="replace"))
                tok.scores.append(float(it["sc"]))
                tok.types.append(int(it["ty"]))
        elif b["bt"] == BT_TENSOR:
            shape = [s for s in struct.unpack_from("<4I", mv, pos + 16) if s > 0]
            kind, layer, ssz = struct.unpack_from("<HHI", mv, pos + 32)
            npdt = _DT2NP[b["dt"]]
            n = int(np.prod(shape))
            d0 = pos + b["hs"]
            vals = np.frombuffer(mv, dtype=npdt, count=n, offset=d0).reshape(shape)
            if ssz:
                sc = np.frombuffer(mv, dtype=np.float32, count=ssz, offset=d0 + vals.nbytes)
                tensors[(kind, layer)] = (vals, sc.reshape(shape[:-1] + [shape[-1] // cfg.quant_group_size]))
            else:
                tensors[(kind, layer)] = vals
        pos += b["size"]
    return cfg, tok, tensors
