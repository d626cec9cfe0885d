"""Op-level parity: HIP kernels (through the C ABI) vs the CPU oracle on identical seeded inputs.
EVERYTHING is bit-exact, fp32 included: the kernels reproduce the reference's order of operations
(see the design rule at the top of fast-llama_amd/csrc/flm_kernels.h)."""
import ctypes
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
@pytest.mark.parametrize("m,n,w", [(96, 256, 1), (96, 256, 3), (64, 11008, 1), (130, 512, 5), (4096, 4096, 1), (7, 64, 2), (1000, 768, 1), (33000, 128, 1), (20000, 1024, 2)])
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
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))      # same fp32 chain order as quant::matmul


@pytest.mark.parametrize("n", [64, 768, 4096, 11008, 12288, 16384])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_rmsnorm(gpu, n, seed):
    rng = np.random.default_rng(n + seed)
    x = rng.standard_normal(n).astype(np.float32) * 3
    w = rng.uniform(0.5, 1.5, n).astype(np.float32)
    o = gpu.op_rmsnorm(x, w)
    ref = O.rmsnorm(x, w)
    assert np.array_equal(o.view(np.uint32), ref.view(np.uint32))        # 4-lane strided sum of squares, as square_sum_avx128


def test_swiglu(gpu):
    rng = np.random.default_rng(3)
    a = (rng.standard_normal(11008) * 4).astype(np.float32); b = rng.standard_normal(11008).astype(np.float32)
    a[:4] = [0.0, -30.0, 30.0, 1e-8]
    assert np.array_equal(gpu.op_swiglu(a, b).view(np.uint32), O.swiglu(a, b).view(np.uint32))


def test_expf_bit_exact_vs_libm(gpu):
    """the device expf is glibc's algorithm; compare with the host libm the reference links against"""
    libm = ctypes.CDLL("libm.so.6"); libm.expf.restype = ctypes.c_float; libm.expf.argtypes = [ctypes.c_float]
    rng = np.random.default_rng(1)
    x = np.concatenate([-rng.random(200000, dtype=np.float32) * 40, (rng.random(200000, dtype=np.float32) - 0.5) * 200,
                        np.array([0.0, -0.0, 1.0, -1.0, 88.0, 88.7, 88.8, -87.0, -87.4, -100.0, -103.5, -104.0, -150.0, np.inf, -np.inf, 1e-30, -1e-30], np.float32)])
    dev = gpu.op_expf(x)
    ref = np.array([libm.expf(float(v)) for v in x], dtype=np.float32)
    assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("hs", [64, 128])
@pytest.mark.parametrize("pos", [0, 1, 37, 1023])
def test_rope_bit_exact(gpu, hs, pos):
    x = np.random.default_rng(hs + pos).standard_normal(hs).astype(np.float32)
    o = gpu.op_rope(x, pos)
    ref = O.rope(x, pos)
    # the cos/sin table comes from the same host libm recurrence -> bit-identical rotation
    assert np.array_equal(o.view(np.uint32), ref.view(np.uint32))


def test_sqrt_div_rmsscale_ieee(gpu):
    """sqrtf and division must be correctly rounded on the device (HIP's __fsqrt_rn is NOT: it is the native 1-ulp sqrt)"""
    rng = np.random.default_rng(2)
    x = np.abs(rng.standard_normal(300000).astype(np.float32)) * np.float32(10.0) ** rng.integers(-8, 8, 300000).astype(np.float32)
    y = (rng.standard_normal(300000).astype(np.float32) + np.float32(3.0)) * np.float32(7.0)
    assert np.array_equal(gpu.op_math(1, x).view(np.uint32), np.sqrt(x).view(np.uint32))
    assert np.array_equal(gpu.op_math(2, x, y).view(np.uint32), (x / y).view(np.uint32))
    n = np.full(x.size, 4096, np.float32)
    ref = (1.0 / np.sqrt(x / n + np.float32(1e-5)).astype(np.float64)).astype(np.float32)       # x86_simd.cpp:1755
    assert np.array_equal(gpu.op_math(3, x, n).view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("n,cols", [(1, 1), (5, 3), (64, 64), (1000, 997), (1024, 1024)])
def test_softmax(gpu, n, cols):
    x = (np.random.default_rng(n).standard_normal(n) * 4).astype(np.float32)
    o = gpu.op_softmax(x, cols)[:cols]
    ref = O.softmax(x, cols)[:cols]
    assert np.array_equal(o.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("hs,heads", [(64, 4), (128, 2), (128, 32), (32, 3), (96, 2)])
def test_attention_decode(gpu, hs, heads):
    """fill the cache token by token with the oracle (prefill + decode), then check every GPU decode step."""
    rng = np.random.default_rng(hs * heads)
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
        out = gpu.op_attention(kc_g, vc_g, q.reshape(-1), k.reshape(-1), v.reshape(-1), heads, hs, max_seq, pos).reshape(heads, hs)
        # new K row (RoPE) and V row appended bit-exactly
        assert np.array_equal(kc_g[:, pos].view(np.uint32), kc_o[:, pos].view(np.uint32))
        assert np.array_equal(vc_g[:, pos].view(np.uint32), vc_o[:, pos].view(np.uint32))
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), f"pos {pos}"
        pos += 1


@pytest.mark.parametrize("parts", [2, 4])
@pytest.mark.parametrize("hs,heads", [(128, 32), (64, 4), (128, 3)])
def test_attention_decode_split_over_workgroups(gpu, hs, heads, parts, monkeypatch):
    """long contexts spread a head over several workgroups (scores by position tile, weighted sum by output dimension, scores
    exchanged through memory inside the launch): same bits as the oracle at short, tile-boundary and long positions"""
    monkeypatch.setenv("FLM_OP_ATTN_PARTS", str(parts))
    rng = np.random.default_rng(hs * heads + parts)
    max_seq = 1024
    kc_o = np.zeros((heads, max_seq, hs), np.float32); vc_o = np.zeros_like(kc_o)
    pos = 0
    for target in [0, 3, 63, 64, 200, 255, 256, 257, 700, 1023]:
        if target > pos:
            bs = target - pos
            q = rng.standard_normal((heads, bs, hs)).astype(np.float32); k = rng.standard_normal((heads, bs, hs)).astype(np.float32)
            v = rng.standard_normal((heads, bs, hs)).astype(np.float32)
            for h in range(heads):
                O.attention_head(kc_o[h], vc_o[h], q[h], k[h], v[h], pos)
            pos = target
        kc_g = kc_o.copy(); vc_g = vc_o.copy()
        q = rng.standard_normal((heads, hs)).astype(np.float32) * 2; k = rng.standard_normal((heads, hs)).astype(np.float32)
        v = rng.standard_normal((heads, hs)).astype(np.float32)
        if target == 700:
            q *= 40.0                                  # a peaked softmax: most weights fall under the 1e-15 skip threshold
        ref = np.stack([O.attention_head(kc_o[h], vc_o[h], q[h:h + 1], k[h:h + 1], v[h:h + 1], pos)[0] for h in range(heads)])
        out = gpu.op_attention(kc_g, vc_g, q.reshape(-1), k.reshape(-1), v.reshape(-1), heads, hs, max_seq, pos).reshape(heads, hs)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), f"pos {pos}"
        pos += 1


def _sq_cases():
    rng = np.random.default_rng(42)
    cases = []
    for n in (64, 256, 768, 1024, 4096, 5120, 11008, 4096 + 16, 16384):
        cases.append(("normal", rng.standard_normal(n).astype(np.float32)))
        cases.append(("wide range", (rng.standard_normal(n) * np.exp2(rng.integers(-40, 40, n))).astype(np.float32)))
        cases.append(("growing", (np.arange(1, n + 1) * 0.37).astype(np.float32)))
        cases.append(("shrinking", (1000.0 / np.arange(1, n + 1)).astype(np.float32)))
        cases.append(("powers of two (exact ties)", np.exp2(rng.integers(-14, 3, n)).astype(np.float32)))
        cases.append(("ones", np.ones(n, np.float32)))
        cases.append(("small integers", rng.integers(-5, 6, n).astype(np.float32)))
        z = rng.standard_normal(n).astype(np.float32); z[rng.random(n) < 0.7] = 0
        cases.append(("mostly zeros", z))
        cases.append(("all zeros", np.zeros(n, np.float32)))
        cases.append(("tiny / denormal squares", (rng.standard_normal(n) * 1e-22).astype(np.float32)))
        cases.append(("huge", (rng.standard_normal(n) * 1e18).astype(np.float32)))
        h = rng.standard_normal(n).astype(np.float32); h[n // 2] = 3e19                 # a square that overflows fp32
        cases.append(("overflow to inf", h))
        big = rng.standard_normal(n).astype(np.float32) * 1e-3; big[5] = 4096.0        # one dominant early term: later ones are below half an ulp
        cases.append(("dominant first term", big))
        t = np.full(n, 2.0 ** -12, np.float32); t[0] = 1.0                              # every later term is EXACTLY half an ulp of the sum: ties all the way
        cases.append(("all ties", t))
    return cases


def test_square_sum_speculative_is_bit_exact(gpu):
    """the speculative wave evaluation of the rmsnorm sum of squares (flm_gemv.h: sq_chain_spec, the one the prologue runs) == the
    sequential chains on the GPU == the CPU oracle's restatement of the reference, on friendly and on adversarial data
    (tools/chain_emul.c fuzzes the same algorithm on the build host)"""
    for name, x in _sq_cases():
        fast, seq, lanes = gpu.op_square_sum(x)
        want = O.square_sum(x)
        f, s, w = np.float32(fast).view(np.uint32), np.float32(seq).view(np.uint32), np.float32(want).view(np.uint32)
        assert s == w, (name, x.size, seq, want)
        assert f == w, (name, x.size, fast, want)
    rng = np.random.default_rng(7)
    for it in range(300):                                   # random magnitudes, a share of exactly representable "round" values
        n = int(rng.choice([256, 1024, 4096, 8192]))
        x = (rng.standard_normal(n) * np.exp2(rng.integers(-12, 12) + rng.integers(-6, 7, n) * (it % 3))).astype(np.float32)
        if it % 4 == 0:
            idx = rng.random(n) < 0.3
            x[idx] = np.exp2(rng.integers(-13, 4, idx.sum())).astype(np.float32) * rng.choice([1.0, 1.5, 3.0], idx.sum()).astype(np.float32)
        fast, seq, lanes = gpu.op_square_sum(x)
        want = O.square_sum(x)
        assert np.float32(fast).view(np.uint32) == np.float32(want).view(np.uint32) == np.float32(seq).view(np.uint32), (it, n, fast, seq, want)


def test_handoff_litmus(gpu):
    """The fused launches hand activations from workgroup to workgroup INSIDE a launch: write-through stores -> s_waitcnt vmcnt(0) -> relaxed flag store; relaxed polls ->
    coherent loads (flm_layer.h).  k_handoff_litmus runs exactly that sequence 10^6 times between one workgroup per CU with data-dependent payloads and checks every value
    read -- a compiler / firmware / memory-model change that reorders it fails here, by name, not as a wrong token id three layers later."""
    bad, timed_out = gpu.op_handoff_litmus(1_000_000)
    assert timed_out == 0, "a flag round never completed (workgroups not co-resident?)"
    assert bad == 0, f"{bad} payload values were read before they were visible"


def test_quantize_shared_reciprocal_equals_ieee_division(gpu):
    """quant_elems4 (flm_math.h): the four divisions x / scale of a quantizer round share the refined reciprocal of the scale -- the instructions an IEEE fp32 division is lowered
    to, minus the operand scaling that does nothing in the range the fast path accepts.  Against quant_elem (the plain IEEE division) on 10^8 pairs: quotients ON and one ulp
    off every truncation boundary k = 0 .. 127 (int8) and around 5792 (int16), random quotients, scales from 2^-126 to 2^126 (beyond 2^+-100 the wave takes the plain
    division), denormals, zeros, infinities."""
    rng = np.random.default_rng(5)
    n = 1 << 22
    for rep in range(24):
        e = rng.integers(-126, 127, n) if rep % 4 == 3 else rng.integers(-40, 40, n)
        sc = (rng.random(n, dtype=np.float32) + np.float32(1.0)) * np.exp2(e).astype(np.float32)
        if rep % 3 == 0:      # on and around the boundaries: x = fl(k sc) nudged by -2 .. +2 ulps
            k = rng.integers(0, 129 if rep % 2 else 5795, n).astype(np.float32)
            x = (k * sc).astype(np.float32)
            x = (x.view(np.int32) + rng.integers(-2, 3, n).astype(np.int32)).view(np.float32)
        elif rep % 3 == 1:    # random quotients in (-130, 130) / (-5800, 5800)
            x = ((rng.random(n, dtype=np.float32) * 2 - 1) * np.float32(130 if rep % 2 else 5800) * sc).astype(np.float32)
        else:                 # anything: tiny, huge, denormal, zero
            x = (rng.standard_normal(n).astype(np.float32) * np.exp2(rng.integers(-149, 120, n)).astype(np.float32))
            x = np.clip(x, -8000 * sc, 8000 * sc).astype(np.float32)      # (the quantizer's precondition: |x| <= the group's maximum = F scale)
            x[::97] = 0.0; sc[::101] = 0.0; x[::103] = np.float32(1e-42)
        x = np.where(np.isfinite(x), x, np.float32(1.0)).astype(np.float32)
        fast = gpu.op_math(4, x, sc)
        ieee = gpu.op_math(5, x, sc)
        bad = np.nonzero(fast.view(np.uint32) != ieee.view(np.uint32))[0]
        assert bad.size == 0, (rep, bad.size, x[bad[:4]], sc[bad[:4]], fast[bad[:4]], ieee[bad[:4]])
