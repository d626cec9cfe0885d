"""Pin the oracle against the REAL reference (oracle/_ref/libflref.so = reference sources compiled by
oracle/Makefile + a thin shim).  Runs wherever the prebuilt artefact exists (build container; it also
travels to the GPU box); skipped otherwise -- the committed goldens cover that case."""
import numpy as np
import pytest

import oracle_py as O
from fast_llama_amd import flmfile as ff, synth

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libflref.so not built (needs /root/reference)")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("seed", range(4))
def test_ops_random(seed):
    R = O.ref()
    rng = np.random.default_rng(seed)
    for qt, dt, lim in ((O.QT_INT8, np.int8, 127), (O.QT_INT16, np.int16, 5792)):
        x = (rng.standard_normal(64 * 64) * rng.uniform(1e-3, 50)).astype(np.float32)
        x[rng.integers(0, 64) * 64:][:64] = 0
        q, s = O.quantize(x, qt); q2, s2 = O.quantize(x, qt, lib=R)
        assert np.array_equal(q, q2) and np.array_equal(bits(s), bits(s2))
        m, n, w = int(rng.integers(1, 200)), 64 * int(rng.integers(1, 40)), int(rng.integers(1, 20))
        W = rng.integers(-lim, lim + 1, (m, n)).astype(dt); X = rng.integers(-lim, lim + 1, (w, n)).astype(dt)
        sW = rng.uniform(1e-4, 1e-3, (m, n // 64)).astype(np.float32); sX = rng.uniform(1e-3, 1e-2, (w, n // 64)).astype(np.float32)
        assert np.array_equal(bits(O.matmul_q(qt, W, sW, X, sX)), bits(O.matmul_q(qt, W, sW, X, sX, lib=R)))
    n = 64 * int(rng.integers(1, 100))
    x = (rng.standard_normal(n) * 3).astype(np.float32); w = rng.uniform(0.5, 1.5, n).astype(np.float32)
    assert np.array_equal(bits(O.rmsnorm(x, w)), bits(O.rmsnorm(x, w, lib=R)))
    assert np.array_equal(bits(O.swiglu(x, w)), bits(O.swiglu(x, w, lib=R)))
    c = int(rng.integers(1, n))
    assert np.array_equal(bits(O.softmax(x, c)[:c]), bits(O.softmax(x, c, lib=R)[:c]))
    for hs in (64, 128):
        xx = rng.standard_normal(hs).astype(np.float32); pos = int(rng.integers(0, 1024))
        assert np.array_equal(bits(O.rope(xx, pos)), bits(O.rope(xx, pos, lib=R)))


@pytest.mark.parametrize("shape,qt,f32,threads", [("tiny", O.QT_INT8, False, 2), ("tiny", O.QT_INT16, False, 1), ("tiny", O.QT_INT8, True, 2),
                                                  ("small", O.QT_INT16, False, 4)])
def test_model_bit_exact(tmp_path, shape, qt, f32, threads):
    cfg = synth.make_config(shape, qt)
    path = tmp_path / "m.flm"
    tensors = synth.write_synthetic_flm(str(path), cfg, seed=31, fp32_master=f32)
    rm = O.RefModel(path, qt, threads=threads)
    om = O.OracleModel(cfg, tensors)
    prompt = np.array([1] + [int(x) for x in (np.arange(1, 11) * 7919) % cfg.vocab_size], np.int32)
    pos, cur = 0, prompt
    for _ in range(10):
        lr = rm.forward(cur, pos); lo = om.forward(cur, pos)
        assert np.array_equal(bits(lr), bits(lo))
        pos += len(cur); cur = np.array([int(np.argmax(lr))], np.int32)
