"""Op-level parity: HIP kernels (through the C ABI) vs the CPU oracle on identical seeded inputs.
Integer/byte results bit-exact; fp32 within the stated tolerance (summation order differs)."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu
QTS = [(O.QT_INT8, np.int8, 127), (O.QT_INT16, np.int16, 5792)]


def _edge_vector(rng, n):
    x = (rng.standard_normal(n) * rng.uniform(0.01, 5)).astype(np.float32)
    if n >= 256:
        x[64:128] = 0.0                               # all-zero group -> q = 0, scale = 0
        x[128:192] = -np.abs(x[128:192])              # group whose max is negative
        x[200] = np.abs(x[192:256]).max() * 1.0       # value exactly at +max
        x[201] = -np.abs(x[192:256]).max()            # and at -max
    return x


@pytest.mark.parametrize("qt,dt,lim", QTS)
@pytest.mark.parametrize("n", [64, 256, 4096, 11008, 16384, 64 * 4000])
def test_quantize_bit_exact(gpu, qt, dt, lim, n):
    rng = np.random.default_rng(n + qt)
    x = _edge_vector(rng, n)
    q, s = gpu.op_quantize(x, qt)
    qo, so = O.quantize(x, qt)
    assert np.array_equal(q, qo)
    assert np.array_equal(s.view(np.uint32), so.view(np.uint32))


@pytest.mark.parametrize("qt,dt,lim", QTS)
@pytest.mark.parametrize("m,n,w", [(96, 256, 1), (96, 256, 3), (64, 11008, 1), (130, 512, 5), (4096, 4096, 1), (7, 64, 2), (1000, 768, 1)])
def test_matmul_q(gpu, qt, dt, lim, m, n, w):
    rng = np.random.default_rng(m * 7 + n + w + qt)
    W = rng.integers(-lim, lim + 1, (m, n)).astype(dt)
    X = rng.integers(-lim, lim + 1, (w, n)).astype(dt)
    sW = rng.uniform(1e-4, 1e-3, (m, n // 64)).astype(np.float32)
    sX = rng.uniform(1e-3, 1e-2, (w, n // 64)).astype(np.float32)
    if n >= 128:
        X[:, 64:128] = 0; sX[:, 1] = 0.0          # an all-zero activation group (scale 0)
    out = gpu.op_matmul_q(qt, W, sW, X, sX)
    ref = O.matmul_q(qt, W, sW, X, sX)
    # identical integer dots and per-group scaling; only the fp32 summation order over groups differs
    denom = np.abs(ref).max()
    assert np.max(np.abs(out - ref)) <= 2e-6 * denom


@pytest.mark.parametrize("n", [64, 768, 4096, 11008])
def test_rmsnorm(gpu, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32) * 3
    w = rng.uniform(0.5, 1.5, n).astype(np.float32)
    o = gpu.op_rmsnorm(x, w)
    ref = O.rmsnorm(x, w)
    np.testing.assert_allclose(o, ref, rtol=1e-6, atol=0)


def test_swiglu(gpu):
    rng = np.random.default_rng(3)
    a = (rng.standard_normal(11008) * 4).astype(np.float32); b = rng.standard_normal(11008).astype(np.float32)
    a[:4] = [0.0, -30.0, 30.0, 1e-8]
    np.testing.assert_allclose(gpu.op_swiglu(a, b), O.swiglu(a, b), rtol=2e-6, atol=1e-30)


@pytest.mark.parametrize("hs", [64, 128])
@pytest.mark.parametrize("pos", [0, 1, 37, 1023])
def test_rope_bit_exact(gpu, hs, pos):
    x = np.random.default_rng(hs + pos).standard_normal(hs).astype(np.float32)
    o = gpu.op_rope(x, pos)
    ref = O.rope(x, pos)
    # the cos/sin table comes from the same host libm recurrence -> bit-identical rotation
    assert np.array_equal(o.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("n,cols", [(1, 1), (5, 3), (64, 64), (1000, 997), (1024, 1024)])
def test_softmax(gpu, n, cols):
    x = (np.random.default_rng(n).standard_normal(n) * 4).astype(np.float32)
    o = gpu.op_softmax(x, cols)[:cols]
    ref = O.softmax(x, cols)[:cols]
    np.testing.assert_allclose(o, ref, rtol=3e-6, atol=1e-12)


@pytest.mark.parametrize("hs,heads", [(64, 4), (128, 2), (128, 32)])
@pytest.mark.parametrize("splits", [1, 0, 4, 8])
def test_attention_decode(gpu, hs, heads, splits):
    """fill the cache token by token with the oracle (prefill + decode), then check every GPU decode step."""
    rng = np.random.default_rng(hs * heads + splits)
    max_seq = 1024
    steps = [0, 1, 2, 5, 63, 64, 65, 130, 257]
    kc_o = np.zeros((heads, max_seq, hs), np.float32); vc_o = np.zeros_like(kc_o)
    kc_g = np.zeros_like(kc_o); vc_g = np.zeros_like(kc_o)
    pos = 0
    for target in steps:
        # advance both caches with the oracle up to `target` (batched prefill path, bs > 1)
        if target > pos:
            bs = target - pos
            q = rng.standard_normal((heads, bs, hs)).astype(np.float32); k = rng.standard_normal((heads, bs, hs)).astype(np.float32)
            v = rng.standard_normal((heads, bs, hs)).astype(np.float32)
            for h in range(heads):
                O.attention_head(kc_o[h], vc_o[h], q[h], k[h], v[h], pos)
            kc_g[:] = kc_o; vc_g[:] = vc_o
            pos = target
        q = rng.standard_normal((heads, hs)).astype(np.float32) * 2; k = rng.standard_normal((heads, hs)).astype(np.float32)
        v = rng.standard_normal((heads, hs)).astype(np.float32)
        ref = np.stack([O.attention_head(kc_o[h], vc_o[h], q[h:h + 1], k[h:h + 1], v[h:h + 1], pos)[0] for h in range(heads)])
        out = gpu.op_attention(kc_g, vc_g, q.reshape(-1), k.reshape(-1), v.reshape(-1), heads, hs, max_seq, pos, splits).reshape(heads, hs)
        # new K row (RoPE) and V row appended bit-exactly
        assert np.array_equal(kc_g[:, pos].view(np.uint32), kc_o[:, pos].view(np.uint32))
        assert np.array_equal(vc_g[:, pos].view(np.uint32), vc_o[:, pos].view(np.uint32))
        np.testing.assert_allclose(out, ref, rtol=0, atol=3e-6 * max(1.0, np.abs(ref).max()))
        pos += 1
