"""A fixed subset of tools/fuzz_shapes.py inside the GPU suite: seeded random model shapes (odd group counts, head sizes 32 / 64 / 128, both
quant types) against the CPU oracle, logits bit for bit -- short prompts (token by token), long prompts (the batched prompt kernels, then decode steps
with the heads spread over workgroups and QKV in the attention's launch where the shape allows it) -- plus the zero-column case of the prompt path's
weighted sum on the matrix cores."""
import numpy as np
import pytest

import oracle_py as O
from fast_llama_amd import flmfile as ff, synth

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def _shape(rng):
    hs = int(rng.choice([32, 64, 128])); heads = int(rng.integers(1, 9))
    dim = hs * heads
    if dim % 64:
        dim = (dim + 63) // 64 * 64; heads = dim // hs
    return dict(dim=dim, hidden_dim=int(rng.integers(1, 24)) * 64, n_heads=heads, n_kv_heads=heads, n_layers=2, vocab_size=int(rng.integers(300, 1500)))


# (seed, long prompt?): the shapes tools/fuzz_shapes.py draws first for these seeds; the long ones exercise k_gemm_q8_mfma / k_gemm_q16_mfma,
# k_qk_mfma, k_attn_pv_mfma and the split-head decode path
CASES = [(1, False), (2, False), (3, False), (4, False), (5, False), (11, True), (12, True), (13, True)]


@pytest.mark.parametrize("seed,long_prompt", CASES)
def test_random_shapes_vs_oracle(gpu, seed, long_prompt):
    rng = np.random.default_rng(seed)
    kw = _shape(rng)
    qt = ff.QT_INT8 if rng.random() < 0.6 else ff.QT_INT16
    cfg = synth.make_config("tiny", qt, **kw)
    tensors = synth.make_tensors(cfg, seed=100 + seed)
    om = O.OracleModel(cfg, tensors)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    npr = int(rng.integers(100, 700)) if long_prompt else int(rng.integers(1, 40))
    prompt = np.array([1] + [int(x) for x in rng.integers(2, cfg.vocab_size, npr - 1)], dtype=np.int32)
    lg = ctx.forward(prompt, 0); lo = om.forward(prompt, 0)
    assert bits_equal(lg, lo), (kw, qt, npr)
    cur, pos = int(np.argmax(lo)), len(prompt)
    for i in range(3):
        t = np.array([cur], np.int32); lg = ctx.forward(t, pos); lo = om.forward(t, pos)
        assert bits_equal(lg, lo), (kw, qt, npr, i)
        cur = int(np.argmax(lo)); pos += 1
    # ... and the device-resident greedy loop from there: the one-launch token where the shape takes it (the embedding row, the classifier and the argmax inside the launch,
    # odd vocabulary sizes -- classifier workgroups with one, two or no rows of their own), graphs of eight tokens + single ones
    n = 11
    want, oc = [], cur
    for k in range(n):
        oc = int(np.argmax(om.forward(np.array([oc], np.int32), pos + k))); want.append(oc)
    assert list(ctx.decode_greedy(cur, pos, n)) == want, (kw, qt, npr)
    assert ctx.query("fallback") == 0
    ctx.close()


@pytest.mark.parametrize("qt", [ff.QT_INT8, ff.QT_INT16])
def test_prompt_weighted_sum_with_all_zero_v_columns(gpu, qt):
    """k_attn_pv_mfma starts its accumulators at -0 and feeds skipped rows with weight +0: an output that is exactly zero can differ from
    the reference's in its SIGN only.  Rows of Wv set to zero make whole columns of V zero for every position, i.e. exactly-zero attention outputs in
    every head; nothing downstream may see a difference: the cache rows of the next layer and the logits are the token-by-token path's (VALU chains)
    and the oracle's bits, and the decode path's attention output (flm_debug_read 2) has +0 there like the reference."""
    cfg = synth.make_config("small", qt)
    tensors = synth.make_tensors(cfg, seed=77)
    hs = cfg.head_size
    zero_rows = [h * hs + d for h in range(cfg.n_heads) for d in (0, 5, hs - 1)]
    for l in range(cfg.n_layers):
        q, s = tensors[(ff.T_ATTN_V, l)]
        q = q.copy(); q[zero_rows, :] = 0
        tensors[(ff.T_ATTN_V, l)] = (q, s)
    om = O.OracleModel(cfg, tensors)
    prompt = np.array([1] + [int(x) for x in (np.arange(1, 90) * 7919) % cfg.vocab_size], dtype=np.int32)
    want = om.forward(prompt, 0)
    out = {}
    for batched in (1, 0):
        ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
        ctx.set_option("use_prefill", batched)
        lg = ctx.forward(prompt, 0)
        att = ctx.debug_read("att_out", 0, cfg.dim)
        kv = [ctx.debug_read("kcache", l, cfg.n_heads * cfg.max_length * hs).reshape(cfg.n_heads, cfg.max_length, hs)[:, :len(prompt)].copy() for l in range(cfg.n_layers)]
        vv = [ctx.debug_read("vcache", l, cfg.n_heads * cfg.max_length * hs).reshape(cfg.n_heads, cfg.max_length, hs)[:, :len(prompt)].copy() for l in range(cfg.n_layers)]
        out[batched] = (lg, att, kv, vv)
        ctx.close()
    assert bits_equal(out[1][0], want) and bits_equal(out[0][0], want)
    assert bits_equal(out[1][1], out[0][1])
    assert np.all(out[1][1].view(np.uint32)[zero_rows] == 0), "an exactly-zero attention output must be +0, as the reference's chain leaves it"
    for l in range(cfg.n_layers):
        assert bits_equal(out[1][2][l], out[0][2][l]), f"K cache rows of layer {l}"
        assert bits_equal(out[1][3][l], out[0][3][l]), f"V cache rows of layer {l}"
        assert np.all(out[1][3][l][:, :, [0, 5, hs - 1]].view(np.uint32) == 0)
