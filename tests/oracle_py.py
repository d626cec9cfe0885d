"""ctypes bindings of the checkers (TEST INFRASTRUCTURE):
  * oracle/liboracle.so       -- this repo's plain-C restatement of the reference hot path
  * oracle/_ref/libflref.so   -- the reference itself behind oracle/ref_harness.cpp (optional; it is
                                 built only where /root/reference exists and travels to the GPU box
                                 as a prebuilt file)
Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libflref.so")
REF_MAIN = os.path.join(ROOT, "oracle", "_ref", "main")

QT_NONE, QT_INT16, QT_INT8 = 0, 1, 2
_orc = None
_ref = None


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _tame_openmp():
    """The checker's OpenMP loops are tiny; on a many-core host (the GPU box has 256 hardware threads and other tenants) the
    default -- one spinning thread per core -- made every oracle call take seconds (a 0.05 s test took 42 s).  Cap the team and
    make idle threads sleep; must run before libgomp starts (it reads the environment once)."""
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, os.cpu_count() or 1))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    os.environ.setdefault("OMP_DYNAMIC", "false")


def orc():
    global _orc
    if _orc is None:
        _tame_openmp()
        if not os.path.exists(ORACLE_SO):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
        _orc = C.CDLL(ORACLE_SO)
        _orc.orc_model_create.restype = C.c_void_p
        _orc.orc_dot_f32.restype = C.c_float
        _orc.orc_square_sum.restype = C.c_float
    return _orc


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        _tame_openmp()
        _ref = C.CDLL(REF_SO)
        _ref.ref_model_load.restype = C.c_void_p
        _ref.ref_dot_f32.restype = C.c_float
        _ref.ref_square_sum.restype = C.c_float
    return _ref


def _qdt(qt):
    return np.int8 if qt == QT_INT8 else np.int16


# ---- op level (oracle) ------------------------------------------------------------------------
def quantize(x, qt, gs=64, lib=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.zeros(x.size, dtype=_qdt(qt)); s = np.zeros(x.size // gs, dtype=np.float32)
    if lib is None:
        orc().orc_quantize(qt, _p(q), _p(s), _p(x), C.c_size_t(x.size), gs)
    else:
        lib.ref_quantize(qt, _p(q), _p(s), _p(x), C.c_size_t(x.size), gs)
    return q, s


def matmul_q(qt, W, sW, X, sX, gs=64, lib=None):
    W = np.ascontiguousarray(W); X = np.ascontiguousarray(X)
    sW = np.ascontiguousarray(sW, dtype=np.float32); sX = np.ascontiguousarray(sX, dtype=np.float32)
    m, n = W.shape; w = X.shape[0]
    out = np.zeros((w, m), dtype=np.float32)
    if lib is None:
        orc().orc_matmul_q(qt, _p(out), _p(W), _p(sW), _p(X), _p(sX), m, n, w, gs)
    else:
        lib.ref_matmul(qt, _p(out), _p(W), _p(sW), _p(X), _p(sX), m, n, w, gs)
    return out


def square_sum(x, lib=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return np.float32((orc().orc_square_sum if lib is None else lib.ref_square_sum)(_p(x), C.c_size_t(x.size)))


def rmsnorm(x, w, lib=None):
    x = np.ascontiguousarray(x, dtype=np.float32); w = np.ascontiguousarray(w, dtype=np.float32)
    o = np.zeros_like(x)
    (orc().orc_rmsnorm if lib is None else lib.ref_rmsnorm)(_p(o), _p(x), _p(w), C.c_size_t(x.size))
    return o


def swiglu(xo, xr, lib=None):
    a = np.array(xo, dtype=np.float32, copy=True); b = np.ascontiguousarray(xr, dtype=np.float32)
    (orc().orc_swiglu if lib is None else lib.ref_swiglu)(_p(a), _p(b), C.c_size_t(a.size))
    return a


def softmax(x, n=None, lib=None):
    a = np.array(x, dtype=np.float32, copy=True)
    (orc().orc_softmax if lib is None else lib.ref_softmax)(_p(a), int(a.size if n is None else n))
    return a


def rope(x, pos, lib=None):
    x = np.ascontiguousarray(x, dtype=np.float32); o = np.zeros_like(x)
    if lib is None:
        orc().orc_rope(_p(o), _p(x), x.size, int(pos))
    else:
        lib.ref_rope_v2(_p(o), _p(x), x.size, 1024, int(pos))
    return o


def weighted_sum(V, att, min_w=1e-15, lib=None):
    V = np.ascontiguousarray(V, dtype=np.float32); att = np.ascontiguousarray(att, dtype=np.float32)
    m, n = V.shape; bs = att.shape[0]
    out = np.zeros((bs, n), dtype=np.float32)
    (orc().orc_weighted_sum if lib is None else lib.ref_weighted_sum)(_p(out), _p(V), _p(att), m, n, bs, C.c_float(min_w))
    return out


def attention_head(kc, vc, q, k, v, pos):
    """oracle ATTN task for one head; kc/vc [max_seq, hs] updated in place; q,k,v [bs, hs] -> out [bs, hs]."""
    q = np.ascontiguousarray(q, dtype=np.float32); k = np.ascontiguousarray(k, dtype=np.float32); v = np.ascontiguousarray(v, dtype=np.float32)
    bs, hs = q.shape
    out = np.zeros((bs, hs), dtype=np.float32)
    scratch = np.zeros(bs * (pos + bs), dtype=np.float32)
    orc().orc_attention_head(_p(out), _p(kc), _p(vc), _p(q), _p(k), _p(v), hs, int(pos), bs, _p(scratch))
    return out


# ---- model level ------------------------------------------------------------------------------
class OracleModel:
    def __init__(self, cfg, tensors, max_seq=1024):
        self.cfg = cfg
        self.h = C.c_void_p(orc().orc_model_create(cfg.dim, cfg.hidden_dim, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads,
                                                   cfg.vocab_size, cfg.quant_type, cfg.quant_group_size, max_seq))
        if not self.h:
            raise RuntimeError("orc_model_create failed")
        for key, v in tensors.items():
            self.set_tensor(key, v)

    def set_tensor(self, key, v):
        kind, layer = key
        if isinstance(v, tuple):
            q = np.ascontiguousarray(v[0]); s = np.ascontiguousarray(v[1], dtype=np.float32)
            qt = QT_INT8 if q.dtype == np.int8 else QT_INT16
            r = orc().orc_model_set_tensor(self.h, kind, layer, qt, _p(q), _p(s), q.shape[0], q.shape[1])
        else:
            a = np.ascontiguousarray(v, dtype=np.float32)
            rows, cols = a.shape if a.ndim == 2 else (1, a.shape[0])
            r = orc().orc_model_set_tensor(self.h, kind, layer, QT_NONE, _p(a), None, rows, cols)
        if r != 0:
            raise RuntimeError(f"orc_model_set_tensor({kind},{layer}) -> {r}")

    def forward(self, tokens, pos):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.zeros(self.cfg.vocab_size, dtype=np.float32)
        r = orc().orc_model_forward(self.h, _p(t), len(t), int(pos), _p(out))
        if r != 0:
            raise RuntimeError(f"orc_model_forward -> {r}")
        return out

    def reset(self):
        orc().orc_model_reset(self.h)

    def __del__(self):
        try:
            if self.h:
                orc().orc_model_free(self.h); self.h = None
        except Exception:
            pass


class RefModel:
    """the reference's ParallelTransformer loaded from a model file (needs oracle/_ref/libflref.so)."""

    def __init__(self, path, qt=QT_INT8, threads=2, max_batch=64):
        self.h = C.c_void_p(ref().ref_model_load(str(path).encode(), b"", qt, threads, max_batch))
        if not self.h:
            raise RuntimeError("reference failed to load " + str(path))
        self.vocab = ref().ref_model_vocab(self.h)

    def forward(self, tokens, pos):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.zeros(self.vocab, dtype=np.float32)
        ref().ref_model_forward(self.h, _p(t), len(t), int(pos), _p(out))
        return out

    def __del__(self):
        try:
            if self.h:
                ref().ref_model_free(self.h); self.h = None
        except Exception:
            pass
