"""Model-level parity on the GPU.  The north_star bound is 1e-3 relative on the logits with identical greedy
ids; the kernels are built to be BIT-IDENTICAL to the reference CPU path, so these tests assert exact
equality of every logit against the CPU oracle (itself bit-exact with the reference,
tests/test_oracle_vs_reference.py) and against the committed golden logits produced by the reference."""
import os

import numpy as np
import pytest

import oracle_py as O
from fast_llama_amd import flmfile as ff, synth

pytestmark = pytest.mark.gpu
REL_TOL = 1e-3      # BASELINE.json north_star: logits within 1e-3 relative fp32 tolerance
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def _prompt(V, n):
    return np.array([1] + [int(x) for x in (np.arange(1, n) * 7919) % V], dtype=np.int32)


@pytest.mark.parametrize("shape,qt,fp32_master", [("tiny", ff.QT_INT8, False), ("tiny", ff.QT_INT16, False), ("tiny128", ff.QT_INT8, False),
                                                  ("tiny", ff.QT_INT8, True), ("small", ff.QT_INT8, False), ("small", ff.QT_INT16, False)])
def test_logits_and_greedy_vs_oracle(gpu, shape, qt, fp32_master):
    cfg = synth.make_config(shape, qt)
    tensors = synth.make_tensors(cfg, seed=1234, fp32_master=fp32_master)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg))
    ctx.upload_all(tensors)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 8)
    lg = ctx.forward(prompt, 0); lo = om.forward(prompt, 0)
    assert rel_err(lg, lo) < REL_TOL
    assert bits_equal(lg, lo)
    pos, cur = len(prompt), int(np.argmax(lo))
    for _ in range(16):
        t = np.array([cur], dtype=np.int32)
        lg = ctx.forward(t, pos); lo = om.forward(t, pos)
        assert bits_equal(lg, lo), f"pos {pos}: rel {rel_err(lg, lo):.3e}"
        cur = int(np.argmax(lo)); pos += 1
    ctx.close()


def test_device_greedy_loop_matches_stepwise(gpu):
    """flm_decode_greedy (hipGraph replay, device-resident state) == step-by-step flm_forward_argmax == oracle."""
    cfg = synth.make_config("small", ff.QT_INT8)
    tensors = synth.make_tensors(cfg, seed=77)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 6)
    first = ctx.forward_argmax(prompt, 0)
    assert first == int(np.argmax(om.forward(prompt, 0)))
    n = 40
    ids = ctx.decode_greedy(first, len(prompt), n)
    # stepwise on a fresh cache
    ctx.reset_kv()
    assert ctx.forward_argmax(prompt, 0) == first
    cur, pos, step_ids, orc_ids = first, len(prompt), [], []
    ocur = first
    for _ in range(n):
        cur = ctx.forward_argmax(np.array([cur], np.int32), pos); step_ids.append(cur)
        ocur = int(np.argmax(om.forward(np.array([ocur], np.int32), pos))); orc_ids.append(ocur)
        pos += 1
    assert list(ids) == step_ids
    assert list(ids) == orc_ids
    # graph on/off and attention split counts give identical tokens
    for key, val in (("use_graph", 0), ("wg_per_cu", 1), ("wg_per_cu", 4), ("use_graph", 1), ("fuse_attn_o", 0), ("fuse_ffn", 0), ("fuse_attn_o", 1), ("fuse_ffn", 1)):
        ctx.set_option(key, val); ctx.reset_kv()
        assert ctx.forward_argmax(prompt, 0) == first
        assert list(ctx.decode_greedy(first, len(prompt), n)) == list(ids)
    ctx.close()


@pytest.mark.parametrize("qt", [ff.QT_INT8, ff.QT_INT16])
def test_7b_width_layer_every_code_path_vs_oracle(gpu, qt):
    """one LLaMA2-7B-width layer + the 32000-row classifier: the production pass geometry, step numbering and LDS layouts
    (the tiny shapes use other ones), with attention + Wo and FFN13 + FFN2 fused into one launch each and as two launches, and with
    QKV in the attention's launch as well (the default only at long contexts)."""
    cfg = synth.make_config("7B", qt); cfg.n_layers = 1
    tensors = synth.make_tensors(cfg, seed=31)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 5)
    want = [om.forward(prompt, 0)]
    cur, pos = int(np.argmax(want[0])), len(prompt)
    for _ in range(3):
        want.append(om.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(want[-1])); pos += 1
    for opts in ({}, {"fuse_token": 0}, {"fuse_layer": 0}, {"fuse_back": 0}, {"fuse_attn_o": 0}, {"fuse_ffn": 0}, {"fuse_attn_o": 0, "fuse_ffn": 0}, {"fuse_qkv": 2}, {"fuse_qkv": 2, "use_prefill": 0}):
        ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
        for k, v in opts.items():
            ctx.set_option(k, v)
        lg = ctx.forward(prompt, 0)
        assert bits_equal(lg, want[0]), (opts, rel_err(lg, want[0]))
        cur, pos = int(np.argmax(lg)), len(prompt)
        for i in range(3):
            lg = ctx.forward(np.array([cur], np.int32), pos)
            assert bits_equal(lg, want[i + 1]), (opts, i, rel_err(lg, want[i + 1]))
            cur = int(np.argmax(lg)); pos += 1
        ctx.close()


@pytest.mark.parametrize("shape,qt,layers", [("7B", ff.QT_INT8, 3), ("7B", ff.QT_INT16, 2), ("small", ff.QT_INT8, None), ("tiny128", ff.QT_INT16, None),
                                             ((512, 1408, 3, 8, 320), ff.QT_INT8, None), ((1024, 2752, 2, 8, 320), ff.QT_INT16, None)])
def test_back_half_of_a_layer_in_one_launch_vs_oracle(gpu, shape, qt, layers):
    """k_layers / k_attn_ffn: all layers of the token in one launch (the next layer's [Wq; Wk; Wv] requested in front of the flag round between two layers: every
    way of sizing that request), one launch per layer, and attention + Wo + FFN13 + FFN2 as one launch (head sizes that are multiples of 64; the other shapes must fall back to the two
    launches by themselves), with every way of filling the LDS stash -- [W1; W3] under the attention, by the head workgroups behind their head, W2 behind
    a workgroup's rows of hd, the first register set in front of the x1 flag round --: logits of a prompt (token by token) and of graph-replayed greedy
    steps are the oracle's bits, whatever the stash holds."""
    cfg = synth.make_config(shape, qt)
    if layers:
        cfg.n_layers = layers
    tensors = synth.make_tensors(cfg, seed=41)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 5)
    want = [om.forward(prompt, 0)]
    cur, pos = int(np.argmax(want[0])), len(prompt)
    for _ in range(4):
        want.append(om.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(want[-1])); pos += 1
    variants = ({}, {"fuse_token": 0}, {"tok_preq": 0, "tok_nstq": 0}, {"tok_preq": 5, "tok_nstq": -1}, {"tok_preq": 16, "tok_nstq": 9, "back_nst13": 2, "use_graph": 0}, {"fuse_layer": 0}, {"back_nst13": 0}, {"back_nst13": 3, "fuse_layer": 0}, {"back_nst13": 17, "back_nst13_head": 5}, {"back_nst13_head": -1, "back_nst2": -1},
                {"back_nst13": 0, "back_nst2": 7, "back_pre13": 1}, {"back_nst13": -1, "back_nst13_head": -1, "back_nst2": -1, "back_pre13": 1, "use_graph": 0},
                # round 5: the Wo / FFN2 hand-offs consumed in arrival order (GemvCtx::run_ao) -- off, one at a time, every way of requesting W2 around the first look
                {"back_ao": 0}, {"back_ao": 1}, {"back_ao": 2, "back_ao2": 2}, {"back_ao2": 1}, {"back_ao": 3, "back_ao2": 3}, {"back_ao": 3, "fuse_token": 0}, {"back_ao": 2, "back_ao2": 2, "fuse_layer": 0, "use_graph": 0})
    for opts in variants:
        ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
        ctx.set_option("use_prefill", 0)
        for k, v in opts.items():
            ctx.set_option(k, v)
        lg = ctx.forward(prompt, 0)
        assert bits_equal(lg, want[0]), (opts, rel_err(lg, want[0]))
        cur, pos = int(np.argmax(lg)), len(prompt)
        for i in range(4):
            lg = ctx.forward(np.array([cur], np.int32), pos)
            assert bits_equal(lg, want[i + 1]), (opts, i, rel_err(lg, want[i + 1]))
            cur = int(np.argmax(lg)); pos += 1
        assert ctx.query("fallback") == 0
        ctx.close()


@pytest.mark.parametrize("shape,qt,layers,n", [("tiny", ff.QT_INT8, None, 70), ("tiny128", ff.QT_INT16, None, 33), ("small", ff.QT_INT8, None, 129),
                                               ("small", ff.QT_INT16, None, 65), ("7B", ff.QT_INT8, 2, 67), ("small", ff.QT_INT8, None, 800),
                                               ("7B", ff.QT_INT8, 1, 200)])
def test_batched_prefill_is_bit_identical_to_token_by_token(gpu, shape, qt, layers, n):
    """prompts go through the batched kernels (GEMM tiles, per-row prologues, causal attention); the cache rows they leave and
    the logits of the last token must be the bits of the token-by-token path (and of the oracle)"""
    cfg = synth.make_config(shape, qt)
    if layers:
        cfg.n_layers = layers
    tensors = synth.make_tensors(cfg, seed=8)
    prompt = _prompt(cfg.vocab_size, n)
    outs = {}
    for mode in (0, 1, 2, 3):                                   # 0 token by token, 1 batched (tile shape by size), 2 batched, 64 x 64 tiles forced, 3 batched, 128 x 128 tiles + fused SwiGLU forced
        ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
        ctx.set_option("use_prefill", 1 if mode else 0); ctx.set_option("use_mfma", {0: 1, 1: 1, 2: 2, 3: 3}[mode])
        lg = ctx.forward(prompt[:5], 0)                       # short prompts stay on the token-by-token path
        lg = ctx.forward(prompt[5:], 5)                       # the batch starts at a non-zero position
        kv = [ctx.debug_read("kcache", l, cfg.n_heads * cfg.max_length * cfg.head_size).reshape(cfg.n_heads, cfg.max_length, -1)[:, :n].copy()
              for l in range(cfg.n_layers)]
        vv = [ctx.debug_read("vcache", l, cfg.n_heads * cfg.max_length * cfg.head_size).reshape(cfg.n_heads, cfg.max_length, -1)[:, :n].copy()
              for l in range(cfg.n_layers)]
        nxt = ctx.forward(np.array([int(np.argmax(lg))], np.int32), n)
        outs[mode] = (lg, kv, vv, nxt)
        ctx.close()
    for m in (1, 2, 3):
        assert bits_equal(outs[0][0], outs[m][0]), m
        for l in range(cfg.n_layers):
            assert bits_equal(outs[0][1][l], outs[m][1][l]), f"mode {m}: K cache layer {l}"
            assert bits_equal(outs[0][2][l], outs[m][2][l]), f"mode {m}: V cache layer {l}"
        assert bits_equal(outs[0][3], outs[m][3]), m
    om = O.OracleModel(cfg, tensors)
    assert bits_equal(outs[1][0], om.forward(prompt, 0))


@pytest.mark.parametrize("shape,qt,n", [("small", ff.QT_INT8, 200), ("tiny128", ff.QT_INT8, 90), ("tiny", ff.QT_INT16, 333),
                                        ((256, 512, 2, 8, 320), ff.QT_INT8, 130), ((384, 768, 2, 4, 320), ff.QT_INT8, 77), ("tiny128", ff.QT_INT8, 1000)])
def test_prefill_attention_on_fp32_mfma_is_the_valu_bits(gpu, shape, qt, n):
    """QK^T of the batched prefill on v_mfma_f32_16x16x4_f32 (eight accumulators = the reference's eight strided lanes, head dimension
    permuted into the k-slots) and the weighted sum on the same instruction (an accumulator element = the chain of one (query,
    dimension), first row by multiplication = an accumulator that starts at -0) against the VALU chains: same cache rows, same logits,
    all equal to the oracle.  Head sizes 32, 64, 96, 128.  If an f32 MFMA were not a k-ordered fmaf chain on this GPU
    (MI355X_MICROARCH / cdna_hip_programming say it is) this is the test that fails."""
    cfg = synth.make_config(shape, qt)
    tensors = synth.make_tensors(cfg, seed=17)
    prompt = _prompt(cfg.vocab_size, n)
    res = []
    for qk, pv in ((1, 1), (1, 0), (0, 0)):
        ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
        ctx.set_option("use_qk_mfma", qk); ctx.set_option("use_pv_mfma", pv)
        lg = ctx.forward(prompt[:7], 0)
        lg = ctx.forward(prompt[7:], 7)                       # a batch that starts at a non-zero position
        kv = [ctx.debug_read(w, cfg.n_layers - 1, cfg.n_heads * cfg.max_length * cfg.head_size).copy() for w in ("kcache", "vcache")]
        res.append((lg.copy(), kv))
        ctx.close()
    for r in res[1:]:
        assert bits_equal(res[0][0], r[0]) and bits_equal(res[0][1][0], r[1][0]) and bits_equal(res[0][1][1], r[1][1])
    assert bits_equal(res[0][0], O.OracleModel(cfg, tensors).forward(prompt, 0))


@pytest.mark.parametrize("n", [3000, 5500])
def test_prompts_longer_than_one_tile_of_exps_fits_the_lds(gpu, n):
    """a prompt of more than ~2500 tokens: 16 queries' exps (64 bytes per position) no longer fit the 160 KiB of LDS, the weighted-sum
    kernel takes 8 (4) queries per workgroup; with the matrix-core weighted sum switched off the 8-query VALU kernel does not fit
    either (max_seq_len 6000) and the one-query-per-workgroup kernel runs.  Same cache rows and logits on every path, equal to the
    token-by-token path's and to the oracle's."""
    cfg = synth.make_config("tiny128", ff.QT_INT8, max_length=6000)
    tensors = synth.make_tensors(cfg, seed=23)
    prompt = _prompt(cfg.vocab_size, n)
    res = []
    for prefill, pv in ((1, 1), (1, 0), (0, 1)) if n == 3000 else ((1, 1), (0, 1)):
        ctx = gpu.Ctx(gpu.desc_from_config(cfg, max_seq_len=cfg.max_length)); ctx.upload_all(tensors)
        ctx.set_option("use_prefill", prefill); ctx.set_option("use_pv_mfma", pv)
        lg = ctx.forward(prompt, 0)
        kv = [ctx.debug_read(w, cfg.n_layers - 1, cfg.n_heads * cfg.max_length * cfg.head_size).copy() for w in ("kcache", "vcache")]
        nxt = ctx.forward(np.array([int(np.argmax(lg))], np.int32), n)
        res.append((lg.copy(), kv, nxt.copy()))
        ctx.close()
    for r in res[1:]:
        assert bits_equal(res[0][0], r[0]) and bits_equal(res[0][1][0], r[1][0]) and bits_equal(res[0][1][1], r[1][1]) and bits_equal(res[0][2], r[2])
    assert bits_equal(res[0][0], O.OracleModel(cfg, tensors, max_seq=cfg.max_length).forward(prompt, 0))


def test_max_seq_len_beyond_the_lds_is_refused_at_create(gpu):
    cfg = synth.make_config("tiny128", ff.QT_INT8, max_length=40000)
    with pytest.raises(Exception, match="max_seq_len"):
        gpu.Ctx(gpu.desc_from_config(cfg, max_seq_len=cfg.max_length))


def test_long_context_positions(gpu):
    """positions up to max_seq_len-1 (1024 clamp, transformer.cpp:32): prefill 1000 tokens on the GPU and the oracle."""
    cfg = synth.make_config("tiny", ff.QT_INT8)
    tensors = synth.make_tensors(cfg, seed=5)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 1000)
    lg = ctx.forward(prompt, 0); lo = om.forward(prompt, 0)
    assert bits_equal(lg, lo)
    t = np.array([int(np.argmax(lo))], np.int32)
    for pos in range(1000, 1024):
        lg = ctx.forward(t, pos); lo = om.forward(t, pos)
        assert bits_equal(lg, lo)
        t = np.array([int(np.argmax(lo))], np.int32)
    with pytest.raises(gpu.FlmError):
        ctx.forward(t, 1024)                      # beyond max_seq_len is an error, not a silent wrap
    ctx.close()


def test_fused_attention_decode_over_the_whole_context_matches_two_launches(gpu):
    """7B width, one layer: 1000 greedy tokens (positions 8..1007: 1 to 16 K/V tiles per head) with attention + Wo as one launch
    and as two, and with QKV in the same launch at every length / never -- the same ids and, at the end, the same logits bits; no
    cross-workgroup wait may time out on the way."""
    cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = 1
    tensors = synth.make_tensors(cfg, seed=77)
    prompt = _prompt(cfg.vocab_size, 8)
    res = []
    for fuse, fq in ((1, 1), (0, 1), (1, 2), (1, 0)):
        ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
        ctx.set_option("fuse_attn_o", fuse); ctx.set_option("fuse_qkv", fq)
        first = ctx.forward_argmax(prompt, 0)
        ids = list(ctx.decode_greedy(first, len(prompt), 1000))
        lg = ctx.forward(np.array([ids[-1]], np.int32), len(prompt) + 1000)
        res.append((first, ids, lg.copy()))
        ctx.close()
    for r in res[1:]:
        assert res[0][0] == r[0] and res[0][1] == r[1]
        assert bits_equal(res[0][2], r[2])


def test_long_context_decode_with_split_heads_vs_oracle(gpu):
    """7B width, two layers (k_layers: SPLIT heads across the edge between layers): a 600-token prompt, then decode steps at positions 600.. with every head spread over 4 workgroups
    (the default from 128 positions on) and over 1 and 2 -- logits bit-equal to the oracle's"""
    cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = 2
    tensors = synth.make_tensors(cfg, seed=19)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 600)
    want = [om.forward(prompt, 0)]
    cur, pos = int(np.argmax(want[0])), len(prompt)
    for _ in range(3):
        want.append(om.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(want[-1])); pos += 1
    for opts in ({}, {"fuse_token": 0}, {"tok_preq": 3, "tok_nstq": 7}, {"fuse_layer": 0}, {"fuse_back": 0}, {"back_nst13": 5, "back_pre13": 3}, {"attn_split": 0}, {"attn_split": 2}, {"fuse_attn_o": 0}, {"use_graph": 0}, {"fuse_qkv": 0}, {"fuse_qkv": 2, "attn_split": 0}):
        ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
        for k, v in opts.items():
            ctx.set_option(k, v)
        lg = ctx.forward(prompt, 0)
        assert bits_equal(lg, want[0]), opts
        cur, pos = int(np.argmax(lg)), len(prompt)
        for i in range(3):
            lg = ctx.forward(np.array([cur], np.int32), pos)
            assert bits_equal(lg, want[i + 1]), (opts, i)
            cur = int(np.argmax(lg)); pos += 1
        ctx.close()


def test_cross_workgroup_handoff_is_stable(gpu):
    """the fused attention + Wo launch hands the heads' output to the GEMV workgroups inside one kernel.  100 replays of a 5-token
    prompt fed token by token plus three more tokens (int16, 7B width: the configuration in which a hand-off that did not wait for the
    heads' stores failed about one replay in fifty) must give the same logits bits every time (tools/stress2.py runs more)."""
    cfg = synth.make_config("7B", ff.QT_INT16); cfg.n_layers = 1
    tensors = synth.make_tensors(cfg, seed=31)
    prompt = _prompt(cfg.vocab_size, 5)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    ctx.set_option("use_prefill", 0)
    ref = None
    for rep in range(100):
        ctx.reset_kv()
        out = [ctx.forward(prompt, 0).copy()]
        cur, pos = int(np.argmax(out[0])), len(prompt)
        for _ in range(3):
            out.append(ctx.forward(np.array([cur], np.int32), pos).copy()); cur = int(np.argmax(out[-1])); pos += 1
        if ref is None:
            ref = out
        for i, (a, b) in enumerate(zip(out, ref)):
            assert bits_equal(a, b), f"replay {rep}, forward {i}"
    ctx.close()


def test_errors(gpu):
    cfg = synth.make_config("tiny", ff.QT_INT8)
    d = gpu.desc_from_config(cfg)
    ctx = gpu.Ctx(d)
    with pytest.raises(gpu.FlmError):             # forward before upload
        ctx.forward(np.array([1], np.int32), 0)
    ctx.close()
    d2 = gpu.desc_from_config(cfg); d2.n_kv_heads = 2
    with pytest.raises(gpu.FlmError):             # GQA unsupported (reference path is broken)
        gpu.Ctx(d2)
    d3 = gpu.desc_from_config(cfg); d3.quant_group_size = 32
    with pytest.raises(gpu.FlmError):
        gpu.Ctx(d3)


@pytest.mark.parametrize("name,shape,qt,f32", [("model_tiny_int8", "tiny", ff.QT_INT8, False), ("model_tiny_int16", "tiny", ff.QT_INT16, False),
                                               ("model_tiny128_int8", "tiny128", ff.QT_INT8, False), ("model_tiny_int8_f32master", "tiny", ff.QT_INT8, True),
                                               ("model_small_int8", "small", ff.QT_INT8, False)])
def test_golden_logits_from_reference(gpu, name, shape, qt, f32):
    """committed golden vectors produced by the reference binary itself (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = synth.make_config(shape, qt)
    tensors = synth.make_tensors(cfg, seed=int(g["seed"]), fp32_master=f32)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    prompt = g["prompt"]
    lg = ctx.forward(prompt, 0)
    assert bits_equal(lg, g["logits"][0])
    pos = len(prompt)
    for i, tok in enumerate(g["ids"][:-1]):
        lg = ctx.forward(np.array([tok], np.int32), pos)
        assert bits_equal(lg, g["logits"][i + 1])
        assert int(np.argmax(lg)) == int(g["ids"][i + 1])
        pos += 1
    ctx.close()


@pytest.mark.parametrize("qt", [ff.QT_INT8, ff.QT_INT16])
@pytest.mark.parametrize("pos0", [5, 130])
def test_greedy_token_as_one_launch_vs_oracle(gpu, pos0, qt):
    """k_layers<.., TAIL>: a greedy decode token as ONE launch -- the embedding row read by the first layer's prologue and Wo epilogue, all layers, the classifier as a phase
    behind the last layer's flag round, the argmax over the classifier workgroups' candidates and the decode state's advance; flag values count from a per-token epoch base
    (nobody clears the lines).  Ids and the last token's logits are the oracle's, with the one-launch token, with the four-launch one, after switching back and forth, eager and
    from the graph; short contexts and split heads."""
    cfg = synth.make_config("7B", qt); cfg.n_layers = 2             # (int16: W2's share is not resident -> the round-4 hand-offs inside the one-launch token)
    tensors = synth.make_tensors(cfg, seed=53)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, pos0)
    lo = om.forward(prompt, 0)
    first = int(np.argmax(lo)); n = 6
    want_ids, cur, pos, last = [], first, len(prompt), None
    for _ in range(n):
        last = om.forward(np.array([cur], np.int32), pos); cur = int(np.argmax(last)); want_ids.append(cur); pos += 1
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    assert ctx.forward_argmax(prompt, 0) == first
    assert ctx.query("token_path") & 1024                    # the one-launch token is what a greedy step runs
    first_n = first, n
    # (graphs of 8 tokens: 19 steps = two chunks + three single tokens; "graph_chunks" 0: a graph launch per token)
    n = 19
    want_ids, cur, pos = [], first, len(prompt)
    for _ in range(n):
        last = om.forward(np.array([cur], np.int32), pos); cur = int(np.argmax(last)); want_ids.append(cur); pos += 1
    # round 6: the one-launch token hands the residual stream over as data-tagged granules ("gr_edges", a tuning dial; default on): both forms, switching back and forth
    for opts in ({}, {"fuse_tail": 0}, {"fuse_tail": 1, "graph_chunks": 0}, {"use_graph": 0}, {"fuse_tail": 0, "graph_chunks": 1}, {"use_graph": 1, "fuse_tail": 1}, {"back_ao": 0},
                 {"tuning": 1, "gr_edges": 0}, {"back_ao": 3}, {"gr_edges": 1, "graph_chunks": 0}, {"graph_chunks": 1}):
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.reset_kv()
        assert ctx.forward_argmax(prompt, 0) == first
        ids = list(ctx.decode_greedy(first, len(prompt), n))
        assert ids == want_ids, (opts, ids, want_ids)
        assert bits_equal(ctx.debug_read("logits", 0, cfg.vocab_size), last), opts
        assert ctx.query("fallback") == 0
        if ctx.query("fuse_tail"):
            assert bool(ctx.query("gr_active")) == bool(ctx.query("gr_edges")), opts      # (the granule form is what the one-launch token ran, unless switched off)
            if pos0 < 128: assert (ctx.query("preq_active"), ctx.query("pre13_active")) == ((12, 10) if ctx.query("gr_edges") else (16, 16)), opts   # (plan_layer's by-launch early sets are what ran)
    ctx.close()


@pytest.mark.parametrize("shape,pos0", [("7B", 700), ("7B", 1000), ((2048, 5504, 2, 32, 32000), 700)])
def test_one_launch_token_with_split_heads_deep_in_the_context_vs_oracle(gpu, shape, pos0):
    """k_layers<.., SPLIT, .., TAIL> from position 512 on: q and this token's K / V row reach the parts of a split head as granules (attn_head<.., SPLIT, .., GRIN>) -- the K piece is
    patched into a pre-landed tile (positions 700..703), into a ring tile when it is parked (704..: the part's third tile; 1000..: its fourth), into a tile requested inside the
    steps loop (head size 64: two parts of six tiles); the V piece lands in the second half of the slice (rows >= 512).  19 greedy ids and the last logits are the oracle's, with
    the granule form and with the flag form."""
    cfg = synth.make_config(shape, ff.QT_INT8); cfg.n_layers = 2
    tensors = synth.make_tensors(cfg, seed=61)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, pos0)
    first = int(np.argmax(om.forward(prompt, 0))); n = 19
    want_ids, cur, pos, last = [], first, len(prompt), None
    for _ in range(n):
        last = om.forward(np.array([cur], np.int32), pos); cur = int(np.argmax(last)); want_ids.append(cur); pos += 1
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    for opts in ({}, {"tuning": 1, "gr_edges": 0}, {"gr_edges": 1, "attn_kpre": 0}, {"attn_kpre": 1, "use_graph": 0}):
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.reset_kv()
        assert ctx.forward_argmax(prompt, 0) == first
        ids = list(ctx.decode_greedy(first, len(prompt), n))
        assert ids == want_ids, (opts, ids, want_ids)
        assert bits_equal(ctx.debug_read("logits", 0, cfg.vocab_size), last), opts
        assert ctx.query("fallback") == 0
        assert bool(ctx.query("gr_active") & 2) == bool(ctx.query("gr_edges")), opts      # (bit 1: the split heads' one-launch token ran, on granules)
        if shape == "7B": assert (ctx.query("preq_active_split"), ctx.query("pre13_active_split")) == ((12, 16) if ctx.query("gr_edges") else (16, 16)), opts   # (plan_layer's by-launch early sets are what ran)
        if not ctx.query("attn_kpre"): assert ctx.query("kpre_active") == 0
        elif shape == "7B" and ctx.query("gr_edges"): assert ctx.query("kpre_active") > 0, opts      # (the pre-landed tiles are what ran)
    ctx.close()


def test_one_launch_token_after_the_embedding_table_is_replaced(gpu):
    """the one-launch token reads the embedding row through a pointer in a device-resident argument block: uploading another table (a new allocation) must rebuild that block"""
    cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = 1
    tensors = dict(synth.make_tensors(cfg, seed=61))
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    prompt = _prompt(cfg.vocab_size, 4)
    for scale in (1.0, -0.5):
        if scale != 1.0:
            tensors[(ff.T_TOKEN_EMBD, 0)] = (tensors[(ff.T_TOKEN_EMBD, 0)] * np.float32(scale)).astype(np.float32)
            ctx.upload(ff.T_TOKEN_EMBD, 0, tensors[(ff.T_TOKEN_EMBD, 0)])
        om = O.OracleModel(cfg, tensors)
        first = int(np.argmax(om.forward(prompt, 0)))
        want, cur, pos = [], first, len(prompt)
        for _ in range(5):
            cur = int(np.argmax(om.forward(np.array([cur], np.int32), pos))); want.append(cur); pos += 1
        ctx.reset_kv()
        assert ctx.forward_argmax(prompt, 0) == first
        assert list(ctx.decode_greedy(first, len(prompt), 5)) == want, scale
    ctx.close()


def test_a_wait_that_gives_up_is_retried_on_one_kernel_per_phase(gpu):
    """the error path of the in-launch hand-offs: a wait that times out (20 ms) raises a flag, the launch runs through, the host re-runs the call on one kernel per phase and returns
    CORRECT results with FLM_OK; the context stays on the per-phase kernels for 64 tokens ("fallback" counts the episodes, "fallback_active" 1), then takes the census again and returns.  The flag is raised by hand here ("inject_wait_failure", a tuning-mode dial): every poll of
    the next call's launches returns at once, so what they compute is garbage -- ids, logits and cache rows must nevertheless be the oracle's after the call."""
    cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = 2
    tensors = synth.make_tensors(cfg, seed=59)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 5)
    first = int(np.argmax(om.forward(prompt, 0)))
    want_ids, cur, pos, last = [], first, len(prompt), None
    for _ in range(10):
        last = om.forward(np.array([cur], np.int32), pos); cur = int(np.argmax(last)); want_ids.append(cur); pos += 1
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    assert ctx.forward_argmax(prompt, 0) == first
    assert list(ctx.decode_greedy(first, len(prompt), 10)) == want_ids and ctx.query("fallback") == 0
    ctx.reset_kv()
    assert ctx.forward_argmax(prompt, 0) == first
    ctx.set_option("inject_wait_failure", 1)
    assert list(ctx.decode_greedy(first, len(prompt), 10)) == want_ids          # the one-launch tokens ran through on garbage, the call was repeated
    assert ctx.query("fallback") == 1 and not (ctx.query("token_path") & 1024)
    assert bits_equal(ctx.debug_read("logits", 0, cfg.vocab_size), last)
    ctx.reset_kv()
    assert ctx.forward_argmax(prompt, 0) == first
    assert list(ctx.decode_greedy(first, len(prompt), 10)) == want_ids          # ... and the context goes on, on one kernel per phase
    assert ctx.query("fallback_active") == 1
    # ... but not for good: after 64 tokens on that path the census runs again, and a context whose workgroups are all resident returns to the launch structure it had
    for _ in range(6):
        ctx.reset_kv()
        assert ctx.forward_argmax(prompt, 0) == first
        assert list(ctx.decode_greedy(first, len(prompt), 10)) == want_ids
    assert ctx.query("fallback") == 1 and ctx.query("fallback_active") == 0 and (ctx.query("token_path") & 1024)     # one episode, over: the one-launch token is back
    ctx.reset_kv()
    assert ctx.forward_argmax(prompt, 0) == first
    assert list(ctx.decode_greedy(first, len(prompt), 10)) == want_ids
    assert bits_equal(ctx.debug_read("logits", 0, cfg.vocab_size), last)
    ctx.close()


def test_option_and_query_surface(gpu):
    """flm_query reads back every option and the path flags; unknown keys and out-of-range values are errors, not silent no-ops; every legal on / off combination
    of the launch-structure options on a tiny model gives the oracle's bits (the options choose launches, never arithmetic)"""
    cfg = synth.make_config("tiny", ff.QT_INT8)
    tensors = synth.make_tensors(cfg, seed=5)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    assert ctx.query("resident") == 1 and ctx.query("fallback") == 0
    tp = ctx.query("token_path")
    assert tp & 1 and tp & 2 and tp & 128 and tp & 256 and tp & 512        # attention + Wo, FFN13 + FFN2, both in one launch, with the QKV GEMV in front, all layers in one launch
    for key in ("fold_xchg", "cu_parts", "fuse_attn_o", "fuse_ffn", "fuse_qkv", "fuse_back", "fuse_layer", "fuse_token", "fuse_tail", "tok_preq", "tok_nstq", "back_nst13", "back_nst13_head", "back_nst2", "back_pre13", "back_ao", "back_ao2", "ao_active",
                "attn_split", "use_graph", "graph_chunks", "use_mfma", "use_prefill", "wg_per_cu"):
        ctx.query(key)
    with pytest.raises(gpu.FlmError):
        ctx.query("no_such_key")
    with pytest.raises(gpu.FlmError):
        ctx.set_option("no_such_key", 1)
    with pytest.raises(gpu.FlmError):
        ctx.set_option("cu_parts", 3)                          # 1, 2, 4 or 8
    with pytest.raises(gpu.FlmError):
        ctx.set_option("engine", 1)                            # round 3's engine left the library
    # the experiment dials (csrc/flm_tuning.h) are not part of the boundary: refused until "tuning" is set
    assert ctx.query("tuning") == 0
    for key in gpu.TUNING_KEYS:
        with pytest.raises(gpu.FlmError):
            ctx.set_option(key, 1, unlock=False)
    ctx.set_option("tuning", 1)
    for key in gpu.TUNING_KEYS:
        if key != "inject_wait_failure":                       # (an action, not a value)
            ctx.set_option(key, ctx.query(key), unlock=False)
    ctx.close()
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 4)
    want = [om.forward(prompt, 0)]
    cur, pos = int(np.argmax(want[0])), len(prompt)
    for _ in range(2):
        want.append(om.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(want[-1])); pos += 1
    import itertools
    keys = ("fuse_token", "fuse_layer", "fuse_back", "fuse_attn_o", "fuse_ffn", "use_graph")
    for vals in itertools.product((0, 1), repeat=len(keys)):
        for fq in (0, 2):
            ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
            for k, v in zip(keys, vals):
                ctx.set_option(k, v)
            ctx.set_option("fuse_qkv", fq)
            lg = ctx.forward(prompt, 0)
            assert bits_equal(lg, want[0]), (vals, fq)
            cur, pos = int(np.argmax(lg)), len(prompt)
            for i in range(2):
                lg = ctx.forward(np.array([cur], np.int32), pos)
                assert bits_equal(lg, want[i + 1]), (vals, fq, i)
                cur = int(np.argmax(lg)); pos += 1
            ctx.close()
