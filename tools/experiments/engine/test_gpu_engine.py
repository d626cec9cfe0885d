"""The weight-streaming engine (fast-llama_amd/csrc/flm_engine.h; option "engine"): loader waves feeding an LDS ring with LDS-DMA, consumer waves
doing the dots, the reference's chains and the cross-CU hand-offs.  Off by default; whatever it computes must be the oracle's bits."""
import numpy as np
import pytest

import oracle_py as O
from fast_llama_amd import flmfile as ff, synth

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


def _prompt(vocab, n):
    return np.array([1] + [int(x) for x in (np.arange(1, n) * 7919) % vocab], dtype=np.int32)


@pytest.mark.parametrize("mode", [1, 2])
def test_engine_7b_width_layers_vs_oracle(gpu, mode):
    """two LLaMA2-7B-width layers + the classifier through the engine: mode 1 = {FFN13, FFN2} per launch (SwiGLU and the quantizer at the
    quant-group owners, hd crossing as granules), mode 2 = {Wo, FFN13, FFN2, next QKV | classifier} per launch (the residual stream crossing
    as granules, rmsnorm chains on the consumer waves, RoPE + cache append as an epilogue); logits of a prompt and of 3 decode steps"""
    cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = 2
    tensors = synth.make_tensors(cfg, seed=31)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 5)
    want = [om.forward(prompt, 0)]
    cur, pos = int(np.argmax(want[0])), len(prompt)
    for _ in range(3):
        want.append(om.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(want[-1])); pos += 1
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    ctx.set_option("engine", mode); ctx.set_option("use_prefill", 0)          # (the prompt token by token: every token through the engine)
    lg = ctx.forward(prompt, 0)
    assert bits_equal(lg, want[0])
    cur, pos = int(np.argmax(lg)), len(prompt)
    for i in range(3):
        lg = ctx.forward(np.array([cur], np.int32), pos)
        assert bits_equal(lg, want[i + 1]), i
        cur = int(np.argmax(lg)); pos += 1
    ctx.close()


@pytest.mark.parametrize("shape,layers", [("small", None), ("7B", 3)])
def test_engine_greedy_ids_equal_the_per_phase_kernels(gpu, shape, layers):
    """graph-replayed greedy decode, 40 tokens, engine modes 1 and 2 against the default kernels (which the other tests pin to the oracle);
    `small` (dim 512, hidden 1536): units shorter than a ring fill, workgroups without rows -- the loaders' general path"""
    cfg = synth.make_config(shape, ff.QT_INT8)
    if layers:
        cfg.n_layers = layers
    tensors = synth.make_tensors(cfg, seed=5)
    prompt = _prompt(cfg.vocab_size, 9)
    ids = {}
    for mode in (0, 1, 2):
        ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
        ctx.set_option("engine", mode)
        first = ctx.forward_argmax(prompt, 0)
        ids[mode] = [first] + [int(x) for x in ctx.decode_greedy(first, len(prompt), 40)]
        x1 = ctx.debug_read("x1", 0, cfg.dim).copy()
        ids[mode].append(x1.view(np.uint32).tobytes())
        ctx.close()
    assert ids[1] == ids[0]
    assert ids[2] == ids[0]


def test_engine_long_context_split_heads(gpu):
    """mode 2 at a position where a head is spread over 4 workgroups: the heads then hand their output over as fp32 (the Wo phase quantizes)"""
    cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = 2
    tensors = synth.make_tensors(cfg, seed=7)
    prompt = _prompt(cfg.vocab_size, 150)
    out = {}
    for mode in (0, 2):
        ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
        ctx.set_option("engine", mode)
        first = ctx.forward_argmax(prompt, 0)
        out[mode] = [first] + [int(x) for x in ctx.decode_greedy(first, len(prompt), 6)]
        ctx.close()
    assert out[2] == out[0]


def test_option_and_query_surface_of_round_3(gpu):
    """flm_query reads back every option and the path flags; unknown keys and out-of-range values are errors, not silent no-ops;
    the engine is refused where it cannot run (int16 models fall back to the per-phase kernels by themselves)"""
    cfg = synth.make_config("tiny", ff.QT_INT8)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(synth.make_tensors(cfg, seed=5))
    assert ctx.query("resident") == 1 and ctx.query("fallback") == 0
    tp = ctx.query("token_path")
    assert tp & 1 and tp & 2                                   # attention + Wo and FFN13 + FFN2 fused on a whole device
    for key in ("engine", "fold_xchg", "cu_parts", "fuse_attn_o", "fuse_ffn", "fuse_qkv", "attn_split", "use_graph", "use_mfma"):
        ctx.query(key)
    with pytest.raises(gpu.FlmError):
        ctx.query("no_such_key")
    with pytest.raises(gpu.FlmError):
        ctx.set_option("no_such_key", 1)
    with pytest.raises(gpu.FlmError):
        ctx.set_option("cu_parts", 3)                          # 1, 2, 4 or 8
    with pytest.raises(gpu.FlmError):
        ctx.set_option("engine", 3)                            # 0, 1 or 2
    ctx.set_option("engine", 2)
    assert ctx.query("engine") == 2 and (ctx.query("token_path") >> 4) & 3 == 2
    ctx.set_option("engine", 0)
    ctx.close()
    # an int16 model: the engine option is accepted and the token path stays on the per-phase kernels (the engine is int8 only), results unchanged
    cfg16 = synth.make_config("tiny", ff.QT_INT16)
    t16 = synth.make_tensors(cfg16, seed=5)
    c16 = gpu.Ctx(gpu.desc_from_config(cfg16)); c16.upload_all(t16)
    om = O.OracleModel(cfg16, t16)
    prompt = np.array([1, 5, 9], np.int32)
    ref = om.forward(prompt, 0)
    c16.set_option("engine", 2)
    assert np.array_equal(c16.forward(prompt, 0).view(np.uint32), ref.view(np.uint32))
    c16.close()
