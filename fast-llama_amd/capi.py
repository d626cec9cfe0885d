"""ctypes binding of the C ABI in include/flm_gpu.h (test and bench plumbing; not the product).

The product library is fast-llama_amd/lib/libflm_gpu.so (hand-written HIP, built by
__graft_entry__.build()).  There is NO CPU fallback: if the library is missing or a call fails,
FlmError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLM_GPU_LIB") or os.path.join(_HERE, "lib", "libflm_gpu.so")   # FLM_GPU_LIB: tools/variants.sh builds

QT_NONE, QT_INT16, QT_INT8 = 0, 1, 2
KCLASSES = ("embed", "qkv", "attn", "attn_o", "ffn13", "ffn2", "cls", "argmax", "allreduce", "attn_wo", "ffn", "qkv_attn_wo", "layer", "back", "layers", "token")   # from "attn_wo" on: the fused launches of the single-GPU token path ("layer": k_attn_ffn, the whole decoder layer in one launch; "back": the same without the QKV GEMV; "layers": k_layers, all layers of the token in one launch; "token": k_layers<.., TAIL>, the whole greedy token in one launch)

# every symbol include/flm_gpu.h declares (tests check the library exports all of them)
SYMBOLS = (
    "flm_comm_unique_id", "flm_ctx_create", "flm_ctx_destroy", "flm_p2p_export", "flm_p2p_import", "flm_last_error", "flm_upload_tensor",
    "flm_forward", "flm_forward_argmax", "flm_decode_greedy", "flm_decode_timed", "flm_decode_timed_each", "flm_last_tokens", "flm_reset_kv", "flm_prepare", "flm_sync",
    "flm_kernel_times", "flm_kernel_bytes", "flm_set_option", "flm_query", "flm_debug_read",
    "flm_op_quantize", "flm_op_matmul_q", "flm_op_rmsnorm", "flm_op_swiglu", "flm_op_rope", "flm_op_softmax",
    "flm_op_attention", "flm_op_expf", "flm_op_math", "flm_op_square_sum", "flm_op_argmax", "flm_op_handoff_litmus", "flm_plan_shards",
)


TUNING_KEYS = ("wg_per_cu", "use_mfma", "tok_preq", "tok_nstq", "back_nst13", "back_nst13_head", "back_nst2", "back_pre13", "back_pre2", "back_ao2", "inject_wait_failure")   # csrc/flm_tuning.h


class FlmError(RuntimeError):
    pass


class ModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size",
                                          "max_seq_len", "quant_type", "quant_group_size")]


class ShardPlan(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("head_begin", "head_count", "hidden_begin", "hidden_count", "dim_begin", "dim_count", "vocab_begin", "vocab_count")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FlmError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _lib.flm_last_error.restype = C.c_char_p
        _lib.flm_last_error.argtypes = [C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _check(rc, ctx=None):
    if rc != 0:
        msg = lib().flm_last_error(ctx)
        raise FlmError(f"flm error {rc}: {msg.decode() if msg else ''}")


def desc_from_config(cfg, max_seq_len=1024) -> ModelDesc:
    return ModelDesc(cfg.dim, cfg.hidden_dim, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads, cfg.vocab_size,
                     max_seq_len, cfg.quant_type, cfg.quant_group_size)


def plan_shards(desc: ModelDesc, rank: int, world: int) -> ShardPlan:
    out = ShardPlan()
    _check(lib().flm_plan_shards(C.byref(desc), rank, world, C.byref(out)))
    return out


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _check(lib().flm_comm_unique_id(buf))
    return buf.raw


class Ctx:
    """flm_ctx wrapper.  tensors: {(kind, layer): fp32 ndarray | (q, scales)} as produced by synth/flmfile."""

    def __init__(self, desc: ModelDesc, device=0, rank=0, world=1, comm_id: bytes | None = None):
        self.desc = desc
        self._h = C.c_void_p()
        _check(lib().flm_ctx_create(C.byref(desc), device, rank, world, comm_id, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().flm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def p2p_export(self) -> bytes:
        buf = C.create_string_buffer(128)
        _check(lib().flm_p2p_export(self._h, buf), self._h)
        return buf.raw

    def p2p_import(self, blobs):
        raw = b"".join(blobs)
        _check(lib().flm_p2p_import(self._h, raw, len(blobs)), self._h)

    @staticmethod
    def regroup(ctxs):
        """ranks of one process: exchange the blobs again (after tensor-parallel options changed: the group's launch structure is agreed at import)"""
        blobs = [c.p2p_export() for c in ctxs]
        for c in ctxs:
            c.p2p_import(blobs)

    def upload(self, kind, layer, value):
        if isinstance(value, tuple):
            q, s = value
            q = np.ascontiguousarray(q); s = np.ascontiguousarray(s, dtype=np.float32)
            qt = QT_INT8 if q.dtype == np.int8 else QT_INT16
            _check(lib().flm_upload_tensor(self._h, kind, layer, qt, _p(q), _p(s), q.shape[0], q.shape[1]), self._h)
        else:
            v = np.ascontiguousarray(value, dtype=np.float32)
            rows, cols = v.shape if v.ndim == 2 else (1, v.shape[0])
            _check(lib().flm_upload_tensor(self._h, kind, layer, QT_NONE, _p(v), None, rows, cols), self._h)

    def upload_all(self, tensors):
        for (kind, layer), v in tensors.items():
            self.upload(kind, layer, v)

    def forward(self, tokens, pos) -> np.ndarray:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.empty(self.desc.vocab_size, dtype=np.float32)
        _check(lib().flm_forward(self._h, _p(t), len(t), int(pos), _p(out)), self._h)
        return out

    def forward_argmax(self, tokens, pos) -> int:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        nxt = C.c_int32(-1)
        _check(lib().flm_forward_argmax(self._h, _p(t), len(t), int(pos), C.byref(nxt)), self._h)
        return nxt.value

    def decode_greedy(self, first_token, pos, n_steps) -> np.ndarray:
        out = np.empty(n_steps, dtype=np.int32)
        _check(lib().flm_decode_greedy(self._h, int(first_token), int(pos), int(n_steps), _p(out)), self._h)
        return out

    def decode_timed(self, first_token, pos, n_steps) -> float:
        ms = C.c_float(0)
        _check(lib().flm_decode_timed(self._h, int(first_token), int(pos), int(n_steps), C.byref(ms)), self._h)
        return ms.value

    def decode_timed_each(self, first_token, pos, n_steps) -> np.ndarray:
        ms = np.zeros(n_steps, dtype=np.float32)
        _check(lib().flm_decode_timed_each(self._h, int(first_token), int(pos), int(n_steps), _p(ms)), self._h)
        return ms

    def last_tokens(self, n) -> np.ndarray:
        out = np.empty(n, dtype=np.int32)
        _check(lib().flm_last_tokens(self._h, int(n), _p(out)), self._h)
        return out

    def reset_kv(self):
        _check(lib().flm_reset_kv(self._h), self._h)

    def prepare(self):
        """build every argument block / token graph now (flm_prepare): nothing is left to allocate inside a forward"""
        _check(lib().flm_prepare(self._h), self._h)

    def sync(self):
        _check(lib().flm_sync(self._h), self._h)

    def set_option(self, key, value, unlock=True):
        """flm_set_option.  The experiment dials (csrc/flm_tuning.h) are not part of the boundary: the library refuses them until option "tuning" is 1 -- the tests and
        tools that sweep them go through here, which unlocks them first (unlock=False: the raw call, as a deployment would make it)."""
        if unlock and key in TUNING_KEYS and not self.query("tuning"):
            _check(lib().flm_set_option(self._h, b"tuning", 1), self._h)
        _check(lib().flm_set_option(self._h, key.encode(), int(value)), self._h)

    def query(self, key) -> int:
        v = C.c_int(0)
        _check(lib().flm_query(self._h, key.encode(), C.byref(v)), self._h)
        return int(v.value)

    def debug_read(self, what, layer, n):
        names = {"x1": 0, "q": 1, "att_out": 2, "hd": 3, "kcache": 4, "vcache": 5, "logits": 6, "trace": 7, "trace_abs": 8, "back_trace": 10}
        out = np.empty(n, dtype=np.float32)
        _check(lib().flm_debug_read(self._h, names[what], int(layer), _p(out), C.c_size_t(n)), self._h)
        return out

    def kernel_times(self, pos, iters=3):
        avg = np.zeros(len(KCLASSES), dtype=np.float32); cnt = np.zeros(len(KCLASSES), dtype=np.int32)
        _check(lib().flm_kernel_times(self._h, int(pos), int(iters), _p(avg), _p(cnt)), self._h)
        return {k: (float(avg[i]), int(cnt[i])) for i, k in enumerate(KCLASSES)}

    def kernel_bytes(self, kclass, pos) -> float:
        b = C.c_double(0)
        _check(lib().flm_kernel_bytes(self._h, KCLASSES.index(kclass), int(pos), C.byref(b)), self._h)
        return b.value


# ---- op level ---------------------------------------------------------------------------------
def op_quantize(x, qt, gs=64):
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.empty(x.size, dtype=np.int8 if qt == QT_INT8 else np.int16)
    s = np.empty(x.size // gs, dtype=np.float32)
    _check(lib().flm_op_quantize(qt, _p(q), _p(s), _p(x), C.c_size_t(x.size), gs))
    return q, s


def op_matmul_q(qt, W, sW, X, sX, gs=64):
    W = np.ascontiguousarray(W); X = np.ascontiguousarray(X)
    sW = np.ascontiguousarray(sW, dtype=np.float32); sX = np.ascontiguousarray(sX, dtype=np.float32)
    m, n = W.shape; w = X.shape[0]
    out = np.empty((w, m), dtype=np.float32)
    _check(lib().flm_op_matmul_q(qt, _p(out), _p(W), _p(sW), _p(X), _p(sX), m, n, w, gs))
    return out


def op_argmax(logits) -> int:
    a = np.ascontiguousarray(logits, dtype=np.float32)
    idx = C.c_int32(-1)
    _check(lib().flm_op_argmax(_p(a), int(a.size), C.byref(idx)))
    return idx.value


def op_handoff_litmus(rounds):
    """(wrong values read, waits timed out) after `rounds` rounds of the fused launches' publish / poll / coherent-read sequence"""
    bad, to = C.c_int(0), C.c_int(0)
    _check(lib().flm_op_handoff_litmus(int(rounds), C.byref(bad), C.byref(to)))
    return bad.value, to.value


def op_rmsnorm(x, w):
    x = np.ascontiguousarray(x, dtype=np.float32); w = np.ascontiguousarray(w, dtype=np.float32)
    o = np.empty_like(x)
    _check(lib().flm_op_rmsnorm(_p(o), _p(x), _p(w), C.c_size_t(x.size)))
    return o


def op_square_sum(x):
    """-> (total from the speculative wave evaluation, total from the sequential chains, the 4 strided partial sums)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    o = np.empty(6, dtype=np.float32)
    _check(lib().flm_op_square_sum(_p(x), C.c_size_t(x.size), _p(o)))
    return o[0], o[1], o[2:6]


def op_swiglu(xo, xr):
    a = np.array(xo, dtype=np.float32, copy=True); b = np.ascontiguousarray(xr, dtype=np.float32)
    _check(lib().flm_op_swiglu(_p(a), _p(b), C.c_size_t(a.size)))
    return a


def op_rope(x, pos):
    x = np.ascontiguousarray(x, dtype=np.float32); o = np.empty_like(x)
    _check(lib().flm_op_rope(_p(o), _p(x), x.size, int(pos)))
    return o


def op_softmax(x, n=None):
    a = np.array(x, dtype=np.float32, copy=True)
    _check(lib().flm_op_softmax(_p(a), int(a.size if n is None else n)))
    return a


def op_expf(x):
    a = np.array(x, dtype=np.float32, copy=True).reshape(-1)
    _check(lib().flm_op_expf(_p(a), C.c_size_t(a.size)))
    return a


def op_math(fn, x, y=None):
    a = np.array(x, dtype=np.float32, copy=True).reshape(-1)
    b = None if y is None else np.ascontiguousarray(y, dtype=np.float32).reshape(-1)
    _check(lib().flm_op_math(int(fn), _p(a), _p(b), C.c_size_t(a.size)))
    return a


def op_attention(kc, vc, q, k, v, n_heads, hs, max_seq, pos):
    """kc, vc [n_heads, max_seq, hs] are updated in place; returns out [n_heads*hs]."""
    out = np.empty(n_heads * hs, dtype=np.float32)
    for a in (kc, vc):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    q = np.ascontiguousarray(q, dtype=np.float32); k = np.ascontiguousarray(k, dtype=np.float32); v = np.ascontiguousarray(v, dtype=np.float32)
    _check(lib().flm_op_attention(_p(out), _p(kc), _p(vc), _p(q), _p(k), _p(v), n_heads, hs, max_seq, int(pos)))
    return out
