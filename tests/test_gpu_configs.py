"""BASELINE.json's configurations on the GPU, each against the reference or the (reference-pinned) oracle, bit for bit:
  config 1  stories110M-shaped llama2.c checkpoint through the drop-in CLI        vs a transcript of the reference CLI
  config 2  Chinese-LLaMA-1.3B shape (4 layers of 7B width, 55296-row classifier) vs the oracle
  config 3  the FULL 32-layer LLaMA2-7B int8 model bench.py times                   vs golden digests produced by the reference
  config 5  7B-width int16 + 512-token batched prefill                              vs the oracle
plus the seams the round-1 review found untested: the tile GEMM kernels through the op-level matmul, the argmax tie on
the device, the quantized-embedding branch of the embedding gather."""
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

import __graft_entry__ as graft
import oracle_py as O
from fast_llama_amd import flmfile as ff, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
MAIN = os.path.join(graft.PKG_DIR, "bin", "main")


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def _prompt(V, n):
    return np.array([1] + [int(x) for x in (np.arange(1, n) * 7919) % V], dtype=np.int32)


def test_config2_1p3B_shape_logits_vs_oracle(gpu):
    """L 4, dim 4096, hidden 11008, V 55296: 8-token prompt + 8 greedy steps"""
    cfg = synth.make_config("1.3B", ff.QT_INT8)
    tensors = synth.make_tensors(cfg, seed=13)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 8)
    lg = ctx.forward(prompt, 0); lo = om.forward(prompt, 0)
    assert bits_equal(lg, lo)
    cur, pos = int(np.argmax(lo)), len(prompt)
    for _ in range(8):
        t = np.array([cur], np.int32)
        lg = ctx.forward(t, pos); lo = om.forward(t, pos)
        assert bits_equal(lg, lo), pos
        cur = int(np.argmax(lo)); pos += 1
    ctx.close()


def _strip_timing(b: bytes) -> bytes:
    b = re.sub(rb"total_latancy:.*", b"total_latancy:<t>", b)
    return re.sub(rb"num_threads:\x1b\[33m *-?\d+\x1b\[0m", b"num_threads:<n>", b)


def test_config1_llama2c_110M_cli_matches_reference_transcript(gpu, tmp_path):
    """stories110M shape (dim 768, hs 64, 12 layers) as a llama2.c .bin: fp32 file quantized to int8 at load, int8 embedding
    table shared with the classifier and dequantized in forward (llama2c_loader.cpp:83,117-124,189-190, transformer.cpp:115-122)"""
    src = open(os.path.join(GOLD, "make_golden_r2.py")).read()
    ns = {}
    exec(re.search(r"LLAMA2C = dict\(.*?\)\n", src, re.S).group(0), ns)
    case = ns["LLAMA2C"]
    cfg = synth.make_config(case["shape"], ff.QT_INT8)
    ck, tk = str(tmp_path / "m.bin"), str(tmp_path / "tokenizer.bin")
    synth.write_llama2c(ck, tk, cfg, np.random.default_rng(case["seed"]))
    if not os.path.exists(MAIN):
        graft.build()
    r = subprocess.run([MAIN, "-c", ck, "-z", tk, "-j", "4", *case["args"]], capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")
    want = bytes(np.load(os.path.join(GOLD, "cli_llama2c_110M.npz"))["transcript"])
    assert _strip_timing(r.stdout) == _strip_timing(want)


def test_config3_full_32_layer_7B_int8_vs_reference_digests(gpu):
    """the model bench.py times, at its full depth: logits of the prompt and of 16 greedy steps hash to the digests of the
    reference's logits (tests/golden/make_golden_r2.py), then the device greedy loop reproduces the reference's ids"""
    g = np.load(os.path.join(GOLD, "model_7B_int8_L32.npz"))
    cfg = synth.make_config("7B", ff.QT_INT8)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg))
    for (kind, layer), v in synth.iter_portable(cfg):
        ctx.upload(kind, layer, v)
    prompt, ids = g["prompt"], g["ids"]

    def dig(l):
        return np.frombuffer(hashlib.sha256(np.ascontiguousarray(l).tobytes()).digest(), dtype=np.uint8)

    lg = ctx.forward(prompt, 0)
    assert bits_equal(lg[:16], g["head"][0]), (lg[:4], g["head"][0][:4])
    assert np.array_equal(dig(lg), g["sha256"][0])
    pos = len(prompt)
    for i in range(16):
        assert int(np.argmax(lg)) == int(ids[i])
        lg = ctx.forward(np.array([ids[i]], np.int32), pos)
        assert bits_equal(lg[:16], g["head"][i + 1]), i
        assert np.array_equal(dig(lg), g["sha256"][i + 1]), i
        pos += 1
    # the graph-replayed device loop from the first generated token on
    rest = ctx.decode_greedy(int(ids[0]), len(prompt), len(ids) - 1)
    assert list(rest) == [int(x) for x in ids[1:]]
    ctx.close()


def _dig(l):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(l).tobytes()).digest(), dtype=np.uint8)


def _upload_portable(gpu, cfg):
    ctx = gpu.Ctx(gpu.desc_from_config(cfg))
    for (kind, layer), v in synth.iter_portable(cfg):
        ctx.upload(kind, layer, v)
    return ctx


def test_config5_full_32_layer_7B_int16_512_token_prefill_vs_reference_digests(gpu):
    """BASELINE config 5 as stated and at full depth: the 32-layer int16 model, the 512-token prompt through the batched path (matrix-core GEMM tiles on hi / lo byte
    planes, QK^T and softmax x V on fp32 MFMA, last row only through the tail: transformer.cpp:92-94,140-142), then 4 greedy steps: every step's logits hash to the
    digests of the REFERENCE's logits (tests/golden/make_golden_r4.py: oracle/_ref/libflref.so with max_batch_size = 512); the 9-token prompt's greedy ids as well"""
    g = np.load(os.path.join(GOLD, "model_7B_int16_L32.npz"))
    cfg = synth.make_config("7B", ff.QT_INT16)
    ctx = _upload_portable(gpu, cfg)
    prompt, ids = g["p512_prompt"], g["p512_ids"]
    lg = ctx.forward(prompt, 0)
    assert bits_equal(lg[:16], g["p512_head"][0]), (lg[:4], g["p512_head"][0][:4])
    assert np.array_equal(_dig(lg), g["p512_sha256"][0])
    pos = len(prompt)
    for i in range(len(ids) - 1):
        assert int(np.argmax(lg)) == int(ids[i])
        lg = ctx.forward(np.array([ids[i]], np.int32), pos)
        assert np.array_equal(_dig(lg), g["p512_sha256"][i + 1]), i
        pos += 1
    # the short prompt of bench.py --quant int16: prompt logits, then the graph-replayed greedy ids
    ctx.reset_kv()
    p9, ids9 = g["p9_prompt"], g["p9_ids"]
    lg = ctx.forward(p9, 0)
    assert np.array_equal(_dig(lg), g["p9_sha256"][0])
    assert list(ctx.decode_greedy(int(ids9[0]), len(p9), len(ids9) - 1)) == [int(x) for x in ids9[1:]]
    ctx.close()


def test_config3_long_prompt_and_long_context_decode_vs_reference_digests(gpu):
    """the int8 model with the 512-token prompt (bench.py's long_context / --pos 512 / prefill512-int8 modes): the batched forward's logits and 26 greedy steps at
    positions 512.. (a head over 4 workgroups, the whole layer in one launch) hash to the reference's digests"""
    g = np.load(os.path.join(GOLD, "model_7B_int8_L32_p512.npz"))
    cfg = synth.make_config("7B", ff.QT_INT8)
    ctx = _upload_portable(gpu, cfg)
    prompt, ids = g["prompt"], g["ids"]
    lg = ctx.forward(prompt, 0)
    assert np.array_equal(_dig(lg), g["sha256"][0])
    pos = len(prompt)
    for i in range(6):
        assert int(np.argmax(lg)) == int(ids[i])
        lg = ctx.forward(np.array([ids[i]], np.int32), pos)
        assert np.array_equal(_dig(lg), g["sha256"][i + 1]), i
        pos += 1
    ctx.reset_kv()
    first = ctx.forward_argmax(prompt, 0)
    assert first == int(ids[0])
    assert list(ctx.decode_greedy(first, len(prompt), len(ids) - 1)) == [int(x) for x in ids[1:]]
    ctx.close()


def test_config2_1p3B_shape_int8_vs_reference_digests(gpu):
    """BASELINE config 2's shape (4 layers, vocabulary 55296): prompt logits and 26 greedy steps against the reference's digests / ids"""
    g = np.load(os.path.join(GOLD, "model_1p3B_int8.npz"))
    cfg = synth.make_config("1.3B", ff.QT_INT8)
    ctx = _upload_portable(gpu, cfg)
    prompt, ids = g["prompt"], g["ids"]
    lg = ctx.forward(prompt, 0)
    assert np.array_equal(_dig(lg), g["sha256"][0])
    pos = len(prompt)
    for i in range(8):
        lg = ctx.forward(np.array([ids[i]], np.int32), pos)
        assert np.array_equal(_dig(lg), g["sha256"][i + 1]), i
        pos += 1
    ctx.reset_kv()
    assert ctx.forward_argmax(prompt, 0) == int(ids[0])
    assert list(ctx.decode_greedy(int(ids[0]), len(prompt), len(ids) - 1)) == [int(x) for x in ids[1:]]
    ctx.close()


def test_config5_7B_width_int16_512_token_prefill_vs_oracle(gpu):
    """7B width, int16, 2 layers, a 512-token prompt through the batched kernels: last-token logits and the next decode step"""
    cfg = synth.make_config("7B", ff.QT_INT16); cfg.n_layers = 2
    tensors = synth.make_tensors(cfg, seed=55)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 512)
    lg = ctx.forward(prompt, 0); lo = om.forward(prompt, 0)
    assert bits_equal(lg, lo)
    t = np.array([int(np.argmax(lo))], np.int32)
    assert bits_equal(ctx.forward(t, 512), om.forward(t, 512))
    ctx.close()


@pytest.mark.parametrize("qt,dt,lim", [(O.QT_INT8, np.int8, 127), (O.QT_INT16, np.int16, 5792)])
@pytest.mark.parametrize("m,n,w", [(200, 256, 16), (64, 11008, 17), (1000, 4096, 64), (130, 512, 100), (4100, 1024, 200), (96, 256, 800),
                                   (70, 320, 33), (129, 64, 65), (33, 704, 130)])
def test_op_matmul_batched_runs_the_tile_kernels(gpu, qt, dt, lim, m, n, w, monkeypatch):
    """flm_op_matmul_q with w >= 16 goes through k_gemm_q8_mfma / k_gemm_q16_mfma -- the kernels the batched prompt path
    uses -- and must equal quant::matmul's chain bit for bit, as the per-row GEMV does (odd group counts: the MFMA kernels' half
    stage; ragged row / token tiles)"""
    rng = np.random.default_rng(m * 7 + n + w + qt)
    W = rng.integers(-lim, lim + 1, (m, n)).astype(dt); X = rng.integers(-lim, lim + 1, (w, n)).astype(dt)
    sW = rng.uniform(1e-4, 1e-3, (m, n // 64)).astype(np.float32); sX = rng.uniform(1e-3, 1e-2, (w, n // 64)).astype(np.float32)
    if n >= 128: X[:, 64:128] = 0; sX[:, 1] = 0.0
    ref = O.matmul_q(qt, W, sW, X, sX)
    for variant in ("1", "2", "3", "gemv"):          # matrix cores (tile shape by size / 64 x 64 / 128 x 128), a GEMV per row
        monkeypatch.setenv("FLM_OP_GEMM", variant)
        out = gpu.op_matmul_q(qt, W, sW, X, sX)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), variant


def test_argmax_ties_first_maximum_wins(gpu):
    """sample_argmax (sampler.cpp:36-47): the lowest index among equal maxima, whichever thread / wave holds it"""
    rng = np.random.default_rng(4)
    for n in (5, 320, 4097, 32000, 55296):
        lg = rng.standard_normal(n).astype(np.float32)
        assert gpu.op_argmax(lg) == int(np.argmax(lg))
        for idxs in ([n - 1, 0], [7 % n, 3 % n, (n // 2)], [n - 1, n - 2], list(range(0, n, max(1, n // 50)))):
            t = lg.copy(); t[idxs] = 9.0
            assert gpu.op_argmax(t) == min(idxs), (n, idxs)
    assert gpu.op_argmax(np.full(100, -np.inf, np.float32)) == 0
    assert gpu.op_argmax(np.full(100, np.nan, np.float32)) == 0


@pytest.mark.parametrize("qt", [ff.QT_INT8, ff.QT_INT16])
def test_quantized_embedding_table_is_dequantized_like_the_reference(gpu, qt):
    """the llama2.c path keeps the embedding table quantized and dequantizes the token's row in forward (transformer.cpp:115-122):
    decode and batched prefill, vs the oracle"""
    cfg = synth.make_config("tiny", qt)
    tensors = synth.make_tensors(cfg, seed=3)
    emb = tensors[(ff.T_TOKEN_EMBD, 0)]
    q, s = O.quantize(emb.reshape(-1), qt)
    tensors[(ff.T_TOKEN_EMBD, 0)] = (q.reshape(emb.shape), s.reshape(emb.shape[0], -1))
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    om = O.OracleModel(cfg, tensors)
    for prompt in (_prompt(cfg.vocab_size, 3), _prompt(cfg.vocab_size, 40)):
        ctx.reset_kv(); om.reset()
        lg = ctx.forward(prompt, 0); lo = om.forward(prompt, 0)
        assert bits_equal(lg, lo)
        t = np.array([int(np.argmax(lo))], np.int32)
        assert bits_equal(ctx.forward(t, len(prompt)), om.forward(t, len(prompt)))
    ctx.close()


def test_no_skippable_work_in_the_product_library(gpu):
    """the product build refuses the perf-exploration switches (they exist only with -DFLM_ABLATE=1)"""
    if os.environ.get("FLM_ABLATE"):
        pytest.skip("ablation build")
    cfg = synth.make_config("tiny", ff.QT_INT8)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg))
    for key in ("ablate", "trace", "use_mega"):
        with pytest.raises(gpu.FlmError):
            ctx.set_option(key, 1)
    ctx.close()


_ALLOC_CHILD = r"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.environ["FLM_ROOT"])
import __graft_entry__ as graft
graft.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
cnt = ctypes.CDLL(None)                      # the LD_PRELOADed interposer (tests/helpers/hipcount.c)
cnt.hipcount_allocs.restype = ctypes.c_long
hip = ctypes.CDLL("libamdhip64.so")
def free_bytes():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value
out = {}
for shape, qt, layers, nprompt in (("7B", ff.QT_INT8, 2, 9), ("small", ff.QT_INT16, None, 140), ("tiny", ff.QT_INT8, None, 3)):
    cfg = synth.make_config(shape, qt)
    if layers: cfg.n_layers = layers
    tensors = synth.make_tensors(cfg, seed=3)
    ctx = capi.Ctx(capi.desc_from_config(cfg)); ctx.upload_all(tensors)       # (the last tensor's arrival builds the argument blocks and the token graphs)
    prompt = np.array([1] + [int(x) for x in (np.arange(1, nprompt) * 7919) % cfg.vocab_size], np.int32)
    a0, f0 = cnt.hipcount_allocs(), free_bytes()
    lg = ctx.forward(prompt, 0)                                               # the context's FIRST forward: batched prompt kernels + a token with logits
    first = ctx.forward_argmax(prompt, 0)
    ids = ctx.decode_greedy(first, len(prompt), 40)                           # ... and its first greedy loop: graph chunks of 16 / 8 / ..., crossing into split heads for the long prompt
    one = ctx.forward(np.array([int(ids[-1])], np.int32), len(prompt) + 40)
    a1, f1 = cnt.hipcount_allocs(), free_bytes()
    out[shape] = {"allocs": a1 - a0, "free_delta": f0 - f1, "first": int(first), "argmax": int(np.argmax(lg)), "counted_before": a0}
    ctx.close()
print("ALLOC " + json.dumps(out))
"""


def test_nothing_is_allocated_inside_forward_and_decode(gpu):
    """include/flm_gpu.h: "nothing is allocated inside flm_forward* / flm_decode_*" (the reference carves its scratch from two arenas made at load time: transformer.cpp:110-130).
    A child process under an LD_PRELOAD interposer (tests/helpers/hipcount.c: hipMalloc, hipExtMallocWithFlags, hipHostMalloc, hipMallocManaged, hipMallocAsync, hipMallocPitch)
    creates a context, uploads a model -- the last tensor's arrival builds k_layers' argument blocks and instantiates every token graph (flm_prepare) -- and then brackets the context's
    FIRST flm_forward (a batched prompt), flm_forward_argmax, flm_decode_greedy (chunk graphs, the crossing into split heads) and a single-token forward with the interposer's count and
    hipMemGetInfo: no allocation call, no byte less free."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "tests", "helpers", "libhipcount.so")
    assert os.path.exists(so), "tests/helpers/libhipcount.so missing: run __graft_entry__.build()"
    env = dict(os.environ, LD_PRELOAD=so, FLM_ROOT=root)
    r = subprocess.run([sys.executable, "-c", _ALLOC_CHILD], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("ALLOC ")][-1][6:])
    for shape, v in res.items():
        assert v["counted_before"] > 20, f"{shape}: the interposer saw no allocation at create / upload -- it is not interposing"
        assert v["first"] == v["argmax"]
        assert v["allocs"] == 0, f"{shape}: {v['allocs']} allocation calls inside forward / decode"
        assert v["free_delta"] <= 0, f"{shape}: {v['free_delta']} bytes less free device memory after the first forward / decode"
