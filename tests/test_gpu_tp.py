"""Tensor parallelism with the one-shot peer-to-peer exchange, on ONE GPU: `world` ranks = `world` contexts on device 0, one host
thread each (the C ABI releases the GIL), their exchange buffers shared by pointer (same process; separate processes use the IPC
handles of the same blobs).  Every rank streams its row shard of every matmul, stores its slice of each activation vector into
all ranks' buffers and meets the others in k_xchg -- the data flow of BASELINE.json's config 4 -- and every rank's logits must be
the single-GPU / reference bits.  (Real multi-GPU runs are the driver's; tests/test_tp_gloo.py covers the same flow on the CPU.)"""
import threading

import numpy as np
import pytest

import oracle_py as O
from fast_llama_amd import flmfile as ff, synth

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def _prompt(V, n):
    return np.array([1] + [int(x) for x in (np.arange(1, n) * 7919) % V], dtype=np.int32)


def _n_devices():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        return 1


def _dev(r, world):
    """rank r's device: a GPU of its own where the box has enough of them (the first run on a multi-GPU node exercises xGMI by itself), else device 0 for everybody"""
    return r if _n_devices() >= world else 0


def _run_ranks(ctxs, fn):
    out, err = [None] * len(ctxs), [None] * len(ctxs)

    def work(r):
        try:
            out[r] = fn(ctxs[r])
        except Exception as e:  # noqa: BLE001
            err[r] = e
    th = [threading.Thread(target=work, args=(r,)) for r in range(len(ctxs))]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not any(t.is_alive() for t in th), "a rank hung"
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("shape,qt,layers,world", [("small", ff.QT_INT8, None, 2), ("small", ff.QT_INT16, None, 2), ("small", ff.QT_INT8, None, 4), ("small", ff.QT_INT16, None, 4), ("7B", ff.QT_INT8, 1, 2), ("7B", ff.QT_INT8, 1, 8)])
def test_tensor_parallel_p2p_is_bit_identical(gpu, shape, qt, layers, world):
    cfg = synth.make_config(shape, qt)
    if layers:
        cfg.n_layers = layers
    tensors = synth.make_tensors(cfg, seed=41)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 7)
    want = [om.forward(prompt, 0)]
    cur, pos = int(np.argmax(want[0])), len(prompt)
    for _ in range(5):
        want.append(om.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(want[-1])); pos += 1
    ids_want = [int(np.argmax(w)) for w in want]
    desc = gpu.desc_from_config(cfg)
    ctxs = [gpu.Ctx(desc, device=_dev(r, world), rank=r, world=world, comm_id=None) for r in range(world)]
    for c in ctxs:
        c.upload_all(tensors)                       # the FULL tensors: each context keeps its row shard
    blobs = [c.p2p_export() for c in ctxs]
    for c in ctxs:
        c.p2p_import(blobs)

    def rank_main(c):
        got = [c.forward(prompt, 0)]
        cur, pos = int(np.argmax(got[0])), len(prompt)
        for _ in range(5):
            got.append(c.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(got[-1])); pos += 1
        c.reset_kv()
        first = c.forward_argmax(prompt, 0)
        ids = [first] + [int(x) for x in c.decode_greedy(first, len(prompt), 5)]      # hipGraph replay of the sharded token
        return got, ids

    for r, (got, ids) in enumerate(_run_ranks(ctxs, rank_main)):
        for i, (a, b) in enumerate(zip(got, want)):
            assert bits_equal(a, b), f"rank {r}, forward {i}"
        assert ids == ids_want, f"rank {r}"
    for c in ctxs:
        c.close()


def test_tensor_parallel_long_context_split_heads(gpu):
    """under TP the attention runs as its own launch: long contexts spread the LOCAL heads over hs/32 workgroups each"""
    cfg = synth.make_config("small", ff.QT_INT8)
    tensors = synth.make_tensors(cfg, seed=43)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 300)
    want = om.forward(prompt, 0)
    t = np.array([int(np.argmax(want))], np.int32)
    want2 = om.forward(t, 300)
    world = 2
    ctxs = [gpu.Ctx(gpu.desc_from_config(cfg), device=_dev(r, world), rank=r, world=world, comm_id=None) for r in range(world)]
    for c in ctxs:
        c.upload_all(tensors)
    blobs = [c.p2p_export() for c in ctxs]
    for c in ctxs:
        c.p2p_import(blobs)
    for got, got2 in _run_ranks(ctxs, lambda c: (c.forward(prompt, 0), c.forward(t, 300))):
        assert bits_equal(got, want) and bits_equal(got2, want2)
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("shape,qt,layers,world,n", [("small", ff.QT_INT8, None, 2, 140), ("small", ff.QT_INT8, None, 4, 33), ("7B", ff.QT_INT8, 2, 2, 150),
                                                     ("7B", ff.QT_INT8, 2, 8, 70), ("small", ff.QT_INT16, None, 2, 75), ("7B", ff.QT_INT16, 2, 4, 90)])
def test_tensor_parallel_batched_prompt(gpu, shape, qt, layers, world, n):
    """prompts under tensor parallelism go through the batched kernels too (peer to peer): every rank runs its heads / rows / hidden
    slice of every step, the kernels store their column slices of the attention output, the residual stream and hd into every rank's
    exchange region, a flag round closes each step.  Logits of the prompt and of the next token = the oracle's bits on every rank, and
    the same as with the prompt fed token by token."""
    cfg = synth.make_config(shape, qt)
    if layers:
        cfg.n_layers = layers
    tensors = synth.make_tensors(cfg, seed=47)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, n)
    want = om.forward(prompt, 0)
    t = np.array([int(np.argmax(want))], np.int32)
    want2 = om.forward(t, n)
    ctxs = [gpu.Ctx(gpu.desc_from_config(cfg), device=_dev(r, world), rank=r, world=world, comm_id=None) for r in range(world)]
    for c in ctxs:
        c.upload_all(tensors)
    blobs = [c.p2p_export() for c in ctxs]
    for c in ctxs:
        c.p2p_import(blobs)

    def rank_main(c):
        a = c.forward(prompt, 0); b = c.forward(t, n)
        c.set_option("use_prefill", 0); c.reset_kv()
        a0 = c.forward(prompt, 0)
        return a, b, a0

    for r, (a, b, a0) in enumerate(_run_ranks(ctxs, rank_main)):
        assert bits_equal(a, want), f"rank {r}: prompt logits"
        assert bits_equal(b, want2), f"rank {r}: next token"
        assert bits_equal(a0, want), f"rank {r}: token by token"
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("qt", [ff.QT_INT8, ff.QT_INT16])
def test_rccl_exchange_branch_on_a_one_rank_communicator(gpu, qt):
    """The RCCL fallback of the exchange (flm_gpu.hip `exchange`: ncclAllGather on the ctx stream; ncclCommInitRank at create) had never executed anywhere:
    a context created with a real ncclUniqueId -- world 1 is enough on a 1-GPU box -- runs the SHARDED token path (one kernel per phase, an all-gather
    behind each of the four activation vectors and the logits, eager launches) over a 1-rank communicator.  Init, the in-place all-gathers and the
    results (the oracle's bits) are what is checked; the collective's cost on xGMI is the driver's to measure."""
    cfg = synth.make_config("small", qt)
    tensors = synth.make_tensors(cfg, seed=43)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 3)                    # (short: fed token by token, every token through the exchanges)
    ctx = gpu.Ctx(gpu.desc_from_config(cfg), device=0, rank=0, world=1, comm_id=gpu.comm_unique_id())
    ctx.set_option("force_tp", 1)               # (a caller that merely passes an id keeps the single-GPU fast path)
    ctx.upload_all(tensors)
    lg = ctx.forward(prompt, 0)
    assert bits_equal(lg, om.forward(prompt, 0))
    cur, pos = int(np.argmax(lg)), len(prompt)
    for _ in range(4):
        t = np.array([cur], np.int32)
        lg = ctx.forward(t, pos)
        assert bits_equal(lg, om.forward(t, pos))
        cur = int(np.argmax(lg)); pos += 1
    ids = ctx.decode_greedy(cur, pos, 5)
    ref, c2, p2 = [], cur, pos
    for _ in range(5):
        c2 = int(np.argmax(om.forward(np.array([c2], np.int32), p2))); p2 += 1; ref.append(c2)
    assert list(ids) == ref
    ctx.close()


@pytest.mark.parametrize("shape,qt,layers,world,fuse", [("small", ff.QT_INT8, None, 2, 1), ("small", ff.QT_INT16, None, 4, 1), ("7B", ff.QT_INT8, 2, 2, 1), ("7B", ff.QT_INT8, 1, 4, 1),
                                                        ("small", ff.QT_INT8, None, 2, 0), ("7B", ff.QT_INT8, 2, 2, 0),
                                                        ("small", ff.QT_INT8, None, 2, 2), ("small", ff.QT_INT16, None, 4, 2), ("7B", ff.QT_INT8, 2, 2, 2), ("7B", ff.QT_INT8, 1, 4, 2)])
def test_folded_exchange_under_cu_masks(gpu, shape, qt, layers, world, fuse):
    """The latency path of the sharded token: every exchange's flag round inside the GEMV launch that consumes the vector ("fold_xchg", the default when
    every rank has CUs of its own) -- 5 launches per layer instead of 9.  On one GPU the ranks get disjoint CU masks ("cu_parts": 2 x 128 or 4 x 64 CUs,
    launches sized to the mask), so a consumer that polls cannot keep its peers' producers off the device.  Logits and graph-replayed greedy ids of
    every rank = the oracle's bits; without the partition the contexts fall back to k_xchg launches by themselves (fold_active 0).
    fuse 1 (the default): attention and the Wo GEMV are ONE launch across the ranks ("tp_fuse_attn": every rank's heads raise their lines in every rank's
    array, the Wo workgroups of every rank wait for all heads of the model), and so are FFN13 and FFN2 ("tp_fuse_ffn": a rank's last workgroup raises the
    rank's line everywhere) -- 3 launches per layer; fuse 0: five; fuse 2: the QKV GEMV joins the attention's launch too (its rows are the rank's own heads) -- 2 per layer."""
    cfg = synth.make_config(shape, qt)
    if layers:
        cfg.n_layers = layers
    tensors = synth.make_tensors(cfg, seed=53)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 4)
    want = [om.forward(prompt, 0)]
    cur, pos = int(np.argmax(want[0])), len(prompt)
    for _ in range(6):
        want.append(om.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(want[-1])); pos += 1
    ids_want = [int(np.argmax(w)) for w in want]
    ctxs = [gpu.Ctx(gpu.desc_from_config(cfg), device=0, rank=r, world=world, comm_id=None) for r in range(world)]
    for c in ctxs:
        c.upload_all(tensors)
    blobs = [c.p2p_export() for c in ctxs]
    for c in ctxs:
        c.p2p_import(blobs)
        assert c.query("fold_active") == 0          # ranks share the device and have no partition yet
        c.set_option("cu_parts", world)
        assert c.query("fold_active") == 0          # ... and what the group runs changes only when the whole group exchanges its blobs again
        c.set_option("tp_fuse_attn", fuse); c.set_option("tp_fuse_ffn", 1 if fuse else 0); c.set_option("tp_fuse_layers", 0)   # (the per-layer structures; the rank-spanning k_layers has its own test below)
    gpu.Ctx.regroup(ctxs)
    for c in ctxs:
        assert c.query("fold_active") == 1 and c.query("span_active") == 1
        assert c.query("grp_tp_fuse_attn") == fuse and c.query("grp_tp_fuse_ffn") == (1 if fuse else 0)

    def rank_main(c):
        lg = [c.forward(prompt, 0)]
        cur, pos = int(np.argmax(lg[0])), len(prompt)
        for _ in range(2):
            lg.append(c.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(lg[-1])); pos += 1
        ids = c.decode_greedy(cur, pos, 4)
        return lg, [int(x) for x in ids]

    for r, (lg, ids) in enumerate(_run_ranks(ctxs, rank_main)):
        for i, l in enumerate(lg):
            assert bits_equal(l, want[i]), f"rank {r}: logits of step {i}"
        assert ids == ids_want[3:7], f"rank {r}: greedy ids"
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("fuse", [1, 0, 2])
def test_fused_attention_across_ranks_with_split_heads(gpu, fuse):
    """long contexts under tensor parallelism with folded exchanges: a rank's heads are spread over hs/32 workgroups each inside the attention + Wo launch that
    spans the ranks (a part raises its own line; the Wo workgroups wait for every part of every head of the model); CU-masked ranks on one GPU, oracle bits"""
    cfg = synth.make_config("small", ff.QT_INT8)
    tensors = synth.make_tensors(cfg, seed=47)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 200)
    want = [om.forward(prompt, 0)]
    cur, pos = int(np.argmax(want[0])), len(prompt)
    for _ in range(4):
        want.append(om.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(want[-1])); pos += 1
    ids_want = [int(np.argmax(w)) for w in want]
    world = 2
    ctxs = [gpu.Ctx(gpu.desc_from_config(cfg), device=0, rank=r, world=world, comm_id=None) for r in range(world)]
    for c in ctxs:
        c.upload_all(tensors)
    for c in ctxs:
        c.set_option("cu_parts", world)
        c.set_option("tp_fuse_attn", fuse); c.set_option("tp_fuse_ffn", 1 if fuse else 0); c.set_option("tp_fuse_layers", 0)
    gpu.Ctx.regroup(ctxs)                           # (the group's launch structure is agreed when the blobs are exchanged)

    def rank_main(c):
        lg = [c.forward(prompt, 0)]
        cur, pos = int(np.argmax(lg[0])), len(prompt)
        lg.append(c.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(lg[-1])); pos += 1
        return lg, [int(x) for x in c.decode_greedy(cur, pos, 3)]

    for r, (lg, ids) in enumerate(_run_ranks(ctxs, rank_main)):
        assert bits_equal(lg[0], want[0]) and bits_equal(lg[1], want[1]), f"rank {r}"
        assert ids == ids_want[2:5], f"rank {r}: greedy ids"
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("shape,qt,layers,world,nprompt", [("small", ff.QT_INT8, None, 2, 4), ("small", ff.QT_INT16, None, 4, 4), ("7B", ff.QT_INT8, 3, 2, 4), ("7B", ff.QT_INT16, 2, 4, 4), ("7B", ff.QT_INT8, 2, 8, 4),
                                                           ("small", ff.QT_INT8, None, 2, 200), ("7B", ff.QT_INT8, 2, 4, 150), ("7B", ff.QT_INT16, 2, 2, 140)])
def test_rank_spanning_layers_under_cu_masks(gpu, shape, qt, layers, world, nprompt):
    """Round 6: ALL layers of a sharded token as ONE launch per rank (k_layers<.., TP>, option "tp_fuse_layers", the default where the group can span): the single-GPU persistent
    launch with the reference's row split (transformer.cpp:264-287) across the ranks -- the four all-to-all hand-offs of a layer (heads -> Wo, x1 -> FFN13, hd -> FFN2, x -> the next
    layer's QKV: transformer.cpp:386-505's tasks) are flag rounds between the ranks' workgroups, every producer storing its slice into every rank's buffer and raising its line in
    every rank's array.  CU-masked ranks on one GPU (2 x 128, 4 x 64, 8 x 32 CUs); short prompts and long ones (split heads: a head over hs / 32 workgroups of its rank);
    logits and graph-replayed greedy ids of every rank = the oracle's bits, and the launch is what ran (tp_layers_active).
    Both forms of the hand-offs: the default -- every cross-rank vector (heads' output, x1, hd, x) as data-tagged 8-byte granules, no flag line and no fence ("gr_edges" 1,
    gr_active) -- and the flag rounds ("gr_edges" 0 on ONE rank: the whole group is back on flags), with and without the fences around the lines."""
    cfg = synth.make_config(shape, qt)
    if layers:
        cfg.n_layers = layers
    tensors = synth.make_tensors(cfg, seed=59)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, nprompt)
    want = [om.forward(prompt, 0)]
    cur, pos = int(np.argmax(want[0])), len(prompt)
    for _ in range(6):
        want.append(om.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(want[-1])); pos += 1
    ids_want = [int(np.argmax(w)) for w in want]
    ctxs = [gpu.Ctx(gpu.desc_from_config(cfg), device=0, rank=r, world=world, comm_id=None) for r in range(world)]
    for c in ctxs:
        c.upload_all(tensors)
        c.set_option("cu_parts", world)
    gpu.Ctx.regroup(ctxs)
    for c in ctxs:
        assert c.query("span_active") == 1 and c.query("grp_tp_fuse_layers") == 1

    def rank_main(c):
        lg = [c.forward(prompt, 0)]
        cur, pos = int(np.argmax(lg[0])), len(prompt)
        for _ in range(2):
            lg.append(c.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(lg[-1])); pos += 1
        ids = c.decode_greedy(cur, pos, 4)
        return lg, [int(x) for x in ids], c.query("tp_layers_active")

    for form in ("granules", "flags", "fenced flags", "granules again"):
        if form == "flags": ctxs[0].set_option("gr_edges", 0)
        elif form == "fenced flags":
            for c in ctxs: c.set_option("tuning", 1); c.set_option("tp_fence", 3)
        elif form == "granules again":
            for c in ctxs: c.set_option("tp_fence", -1)
            ctxs[0].set_option("gr_edges", 1)
        if form != "granules":
            gpu.Ctx.regroup(ctxs)
            for c in ctxs: c.reset_kv()
        for c in ctxs:
            assert c.query("grp_tp_fuse_layers") == 1 and c.query("grp_gr") == (1 if form.startswith("granules") else 0)
        for r, (lg, ids, act) in enumerate(_run_ranks(ctxs, rank_main)):
            for i, l in enumerate(lg):
                assert bits_equal(l, want[i]), f"{form}, rank {r}: logits of step {i}"
            assert ids == ids_want[3:7], f"{form}, rank {r}: greedy ids"
            assert act >= 1 and (nprompt < 128 or act & 2), f"{form}, rank {r}: the rank-spanning launch did not run (tp_layers_active {act})"
        for c in ctxs:
            assert bool(c.query("gr_active")) == form.startswith("granules"), form
    # the structure is an option like the others: off on one rank and the whole group is back on the per-layer launches, same bits
    ctxs[-1].set_option("tp_fuse_layers", 0)
    gpu.Ctx.regroup(ctxs)
    for c in ctxs:
        assert c.query("grp_tp_fuse_layers") == 0
        c.reset_kv()
    for r, lg in enumerate(_run_ranks(ctxs, lambda c: c.forward(prompt, 0))):
        assert bits_equal(lg, want[0]), f"rank {r}: per-layer launches"
    for c in ctxs:
        c.close()


def test_group_agrees_on_the_weakest_launch_structure(gpu):
    """ranks whose options differ must not pick different hand-off protocols (they would wait on flags nobody raises): flm_p2p_import derives the group's
    structure from all blobs -- one rank without folded exchanges, or with another attn_split, and nobody folds / splits; results stay the oracle's bits"""
    cfg = synth.make_config("small", ff.QT_INT8)
    tensors = synth.make_tensors(cfg, seed=77)
    om = O.OracleModel(cfg, tensors)
    prompt = _prompt(cfg.vocab_size, 5)
    want = om.forward(prompt, 0)
    world = 2
    ctxs = [gpu.Ctx(gpu.desc_from_config(cfg), device=0, rank=r, world=world, comm_id=None) for r in range(world)]
    for c in ctxs:
        c.upload_all(tensors); c.set_option("cu_parts", world)
    ctxs[1].set_option("fold_xchg", 0); ctxs[0].set_option("attn_split", 2); ctxs[1].set_option("tp_fuse_ffn", 1)
    gpu.Ctx.regroup(ctxs)
    for c in ctxs:
        assert c.query("fold_active") == 0 and c.query("span_active") == 0 and c.query("grp_attn_split") == 0
        assert c.query("grp_tp_fuse_attn") == 0 and c.query("grp_tp_fuse_ffn") == 0
    for r, lg in enumerate(_run_ranks(ctxs, lambda c: c.forward(prompt, 0))):
        assert bits_equal(lg, want), f"rank {r}"
    for c in ctxs:
        c.close()


# every launch structure of the sharded token, as bench.py's TP_STRUCTURES / host/engine.cpp's candidates name them: (tp_trust_fused, tp_fuse_ffn, tp_fuse_layers, tp_fence, gr_edges)
_TWO_GPU_STRUCTURES = [("xchg-launches", 0, 0, 0, -1, 0), ("folded", 1, 0, 0, -1, 0), ("rank-spanning launch, fenced flags", 1, 0, 1, 3, 0), ("rank-spanning launch, flags", 1, 0, 1, 0, 0),
                       ("rank-spanning launch, granules", 1, 0, 1, -1, 1), ("folded + FFN pair across ranks", 1, 1, 0, -1, 0)]


@pytest.mark.parametrize("qt,rehearsal", [(ff.QT_INT8, False), (ff.QT_INT16, False), (ff.QT_INT8, True)])
def test_two_gpus_p2p_and_trusted_fused_launches(gpu, qt, rehearsal):
    """two ranks on two GPUs (lights up on the first multi-GPU box): by default the group takes the k_xchg launches (a flag round behind a kernel boundary); with "tp_trust_fused" on
    every rank the folded exchanges and the rank-spanning launches over xGMI -- fenced flags, bare flags, and the data-tagged granules that need neither (round 6) --: every
    structure in one run, oracle bits each time (logits of a prompt + two single tokens, graph-replayed greedy ids), short prompts and one long enough for split heads.
    rehearsal: the same test body with both ranks on ONE GPU under CU masks (what a 1-GPU box can run of it: there the group folds without being asked to trust anything)."""
    if not rehearsal and _n_devices() < 2:
        pytest.skip("needs two GPUs")
    cfg = synth.make_config("7B", qt); cfg.n_layers = 2
    tensors = synth.make_tensors(cfg, seed=61)
    om = O.OracleModel(cfg, tensors)
    world = 2
    ctxs = [gpu.Ctx(gpu.desc_from_config(cfg), device=(0 if rehearsal else r), rank=r, world=world, comm_id=None) for r in range(world)]
    for c in ctxs:
        c.upload_all(tensors)
        if rehearsal: c.set_option("cu_parts", world)
    for nprompt in (6, 140):
        prompt = _prompt(cfg.vocab_size, nprompt)
        want = [om.forward(prompt, 0)]
        cur, pos = int(np.argmax(want[0])), len(prompt)
        for _ in range(6):
            want.append(om.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(want[-1])); pos += 1
        ids_want = [int(np.argmax(w)) for w in want]

        def rank_main(c):
            lg = [c.forward(prompt, 0)]
            cur, pos = int(np.argmax(lg[0])), len(prompt)
            for _ in range(2):
                lg.append(c.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(lg[-1])); pos += 1
            return lg, [int(x) for x in c.decode_greedy(cur, pos, 4)]

        for name, trust, ffn, layers, fence, gr in _TWO_GPU_STRUCTURES:
            for c in ctxs:
                c.set_option("tp_trust_fused", trust); c.set_option("tp_fuse_ffn", ffn); c.set_option("tp_fuse_layers", layers); c.set_option("tp_fence", fence); c.set_option("gr_edges", gr)
                c.reset_kv()
            gpu.Ctx.regroup(ctxs)
            for c in ctxs:
                if not rehearsal:
                    assert c.query("fold_active") == trust and c.query("span_active") == trust, name
                assert c.query("grp_tp_fuse_layers") == (1 if (trust or rehearsal) and layers else 0) and c.query("grp_gr") == gr, name
            for r, (lg, ids) in enumerate(_run_ranks(ctxs, rank_main)):
                for i, l in enumerate(lg):
                    assert bits_equal(l, want[i]), f"{name}, prompt of {nprompt}, rank {r}: logits of step {i}"
                assert ids == ids_want[3:7], f"{name}, prompt of {nprompt}, rank {r}: greedy ids"
            for c in ctxs:
                if layers and (trust or rehearsal):
                    assert c.query("tp_layers_active") >= 1 and bool(c.query("gr_active")) == bool(gr), name
    for c in ctxs:
        c.close()


@pytest.mark.skipif(_n_devices() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("mode", ["p2p", "rccl"])
def test_two_gpus_bench_processes(gpu, mode):
    """the driver's N = 2 command line on two real GPUs (one process per GPU over torch.distributed.run): peer to peer, and with the peer mapping refused so that
    the RCCL all-gather branch runs with a 2-rank communicator; the line must carry the reference's ids"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    if mode == "rccl":
        env["FLM_BENCH_NO_P2P"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--shape", "1.3B", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["parity"]["match"] in (True, None) and line["parity"]["replay_identical"]
    if mode == "rccl":
        assert "rccl" in line["config"]["parallelism"]
