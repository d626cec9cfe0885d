"""The CPU oracle (oracle/flm_oracle.c) against the committed golden vectors that were produced by the
reference itself (tests/golden/make_golden.py).  CPU only.  Bit-exact unless stated."""
import os

import numpy as np
import pytest

import oracle_py as O
from fast_llama_amd import flmfile as ff, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(GOLD, "ops.npz"))


@pytest.mark.parametrize("qt,name", [(O.QT_INT8, "i8"), (O.QT_INT16, "i16")])
def test_g1_quantize(ops, qt, name):
    x = ops[f"quant_{name}_x"]
    q, s = O.quantize(x, qt)
    assert np.array_equal(q, ops[f"quant_{name}_q"])
    assert np.array_equal(bits(s), bits(ops[f"quant_{name}_s"]))
    assert np.all(q[64:128] == 0) and s[1] == 0.0                      # all-zero group
    lim = 127 if qt == O.QT_INT8 else 5792
    assert q[200] in (lim, lim - 1) and q[201] in (-lim, -lim + 1)     # truncation at +-max
    # the numpy quantizer used by the .flm writer reproduces the same bytes
    q2, s2 = ff.quantize(x, qt)
    assert np.array_equal(q2, q) and np.array_equal(bits(s2), bits(s))


@pytest.mark.parametrize("qt,name,lim,dt", [(O.QT_INT8, "i8", 127, np.int8), (O.QT_INT16, "i16", 5792, np.int16)])
@pytest.mark.parametrize("m,n,w", [(96, 256, 1), (96, 256, 3), (64, 11008, 1)])
def test_g2_matmul(ops, qt, name, lim, dt, m, n, w):
    mr = np.random.default_rng([qt, m, n, w])
    W = mr.integers(-lim, lim + 1, (m, n)).astype(dt); X = mr.integers(-lim, lim + 1, (w, n)).astype(dt)
    sW = mr.uniform(1e-4, 1e-3, (m, n // 64)).astype(np.float32); sX = mr.uniform(1e-3, 1e-2, (w, n // 64)).astype(np.float32)
    out = O.matmul_q(qt, W, sW, X, sX)
    assert np.array_equal(bits(out), bits(ops[f"matmul_{name}_{m}_{n}_{w}"]))


def test_g3_float_ops(ops):
    for n in (64, 768, 4096):
        assert np.array_equal(bits(O.rmsnorm(ops[f"rms_{n}_x"], ops[f"rms_{n}_w"])), bits(ops[f"rms_{n}_o"]))
    assert np.array_equal(bits(O.swiglu(ops["swiglu_a"], ops["swiglu_b"])), bits(ops["swiglu_o"]))
    assert np.array_equal(bits(O.softmax(ops["softmax_x"], int(ops["softmax_cols"]))), bits(ops["softmax_o"]))
    for hs in (64, 128):
        for pos in (0, 1, 37, 1023):
            assert np.array_equal(bits(O.rope(ops[f"rope_{hs}_{pos}_x"], pos)), bits(ops[f"rope_{hs}_{pos}_o"]))
    assert np.array_equal(bits(O.weighted_sum(ops["wsum_V"], ops["wsum_att"], 1e-15)), bits(ops["wsum_o"]))


@pytest.mark.parametrize("hs", [64, 128])
def test_g4_attention(hs):
    g = np.load(os.path.join(GOLD, "attention.npz"))
    kc = np.zeros((1024, hs), np.float32); vc = np.zeros_like(kc)
    o1 = O.attention_head(kc, vc, g[f"hs{hs}_q"], g[f"hs{hs}_k"], g[f"hs{hs}_v"], 0)          # prefill, bs = 5, causal
    o2 = O.attention_head(kc, vc, g[f"hs{hs}_q2"], g[f"hs{hs}_k2"], g[f"hs{hs}_v2"], 5)       # decode
    assert np.array_equal(bits(o1), bits(g[f"hs{hs}_o"]))
    assert np.array_equal(bits(o2), bits(g[f"hs{hs}_o2"]))
    assert np.array_equal(bits(kc[:6]), bits(g[f"hs{hs}_kc"])) and np.array_equal(bits(vc[:6]), bits(g[f"hs{hs}_vc"]))


def weights_checksum(tensors):
    chk = 0
    for key in sorted(tensors):
        v = tensors[key]
        for a in (v if isinstance(v, tuple) else (v,)):
            chk = (chk * 1000003 + int(np.frombuffer(np.ascontiguousarray(a).tobytes(), dtype=np.uint8).astype(np.uint64).sum())) % (1 << 61)
    return chk


@pytest.mark.parametrize("name,shape,qt,f32", [("model_tiny_int8", "tiny", O.QT_INT8, False), ("model_tiny_int16", "tiny", O.QT_INT16, False),
                                               ("model_tiny128_int8", "tiny128", O.QT_INT8, False), ("model_tiny_int8_f32master", "tiny", O.QT_INT8, True),
                                               ("model_small_int8", "small", O.QT_INT8, False)])
def test_g5_model_logits_and_ids(name, shape, qt, f32):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = synth.make_config(shape, qt)
    tensors = synth.make_tensors(cfg, seed=int(g["seed"]), fp32_master=f32)
    assert weights_checksum(tensors) == int(g["weights_checksum"]), "synthetic weight generator drifted"
    om = O.OracleModel(cfg, tensors)
    pos, cur = 0, g["prompt"]
    for i in range(len(g["ids"])):
        l = om.forward(cur, pos)                       # first step = batched prefill (bs = 8), then decode
        assert np.array_equal(bits(l), bits(g["logits"][i])), f"step {i}"
        assert int(np.argmax(l)) == int(g["ids"][i])
        pos += len(cur); cur = np.array([g["ids"][i]], np.int32)


def test_prefill_equals_token_by_token():
    """row i of the reference's batched prefill only depends on rows <= i: feeding the prompt one token
    at a time gives bit-identical last-row logits (this is what the GPU path relies on)."""
    cfg = synth.make_config("tiny", O.QT_INT8)
    tensors = synth.make_tensors(cfg, seed=8)
    a, b = O.OracleModel(cfg, tensors), O.OracleModel(cfg, tensors)
    prompt = np.array([1, 5, 99, 310, 42, 7, 7, 250], np.int32)
    la = a.forward(prompt, 0)
    for i, t in enumerate(prompt):
        lb = b.forward(np.array([t], np.int32), i)
    assert np.array_equal(bits(la), bits(lb))


def test_argmax_first_max_wins():
    g = np.load(os.path.join(GOLD, "tokenizer_sampler.npz"))
    lg = np.zeros(320, np.float32); lg[g["tie_logits_idx"]] = 5.0
    assert O.orc().orc_argmax(O._p(lg), 320) == int(g["tie_argmax"]) == 3
    r = g["rand_logits"]
    assert O.orc().orc_argmax(O._p(np.ascontiguousarray(r)), r.size) == int(g["rand_t0"]) == int(g["rand_topp"])   # seed-0 top-p == greedy
