"""Tensor-parallel data flow on CPU, world_size 2, gloo.

Each rank owns the row shards `flm_plan_shards` (C library, pure host arithmetic) assigns to it, computes
them with the CPU oracle's operators, and exchanges activation slices with all_gather exactly where the
HIP path calls ncclAllGather (fast-llama_amd/csrc/flm_gpu.hip, enqueue_token).  Because every output row
is reduced on one rank in the reference's order, the sharded logits must be BIT-IDENTICAL to the
unsharded oracle forward -- that is the property that lets the multi-GPU path keep the parity bound.
"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):          # spawned workers import this module without conftest
    if _p not in sys.path:
        sys.path.insert(0, _p)
import __graft_entry__ as _graft  # noqa: E402

_graft.load_package()

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_py as O
from fast_llama_amd import capi, flmfile as ff, synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _gather(local: np.ndarray) -> np.ndarray:
    t = torch.from_numpy(np.ascontiguousarray(local))
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out).numpy()


def _rows(t, begin, count):
    q, s = t
    return np.ascontiguousarray(q[begin:begin + count]), np.ascontiguousarray(s[begin:begin + count])


def sharded_forward(cfg, tensors, plan, tokens, pos0, kc, vc):
    """token-by-token forward of this rank's shard; kc/vc: [layers][local heads][max_seq][hs]"""
    qt, hs, dim = cfg.quant_type, cfg.head_size, cfg.dim
    hb, hn = plan.head_begin, plan.head_count
    logits = None
    for i, tok in enumerate(tokens):
        pos = pos0 + i
        x1 = np.array(tensors[(ff.T_TOKEN_EMBD, 0)][tok], dtype=np.float32)
        for l in range(cfg.n_layers):
            qx, sx = O.quantize(O.rmsnorm(x1, tensors[(ff.T_INPUT_NORM, l)]), qt)
            att_local = np.zeros(hn * hs, np.float32)
            q = O.matmul_q(qt, *_rows(tensors[(ff.T_ATTN_Q, l)], hb * hs, hn * hs), qx[None], sx[None])[0]
            k = O.matmul_q(qt, *_rows(tensors[(ff.T_ATTN_K, l)], hb * hs, hn * hs), qx[None], sx[None])[0]
            v = O.matmul_q(qt, *_rows(tensors[(ff.T_ATTN_V, l)], hb * hs, hn * hs), qx[None], sx[None])[0]
            for h in range(hn):
                sl = slice(h * hs, (h + 1) * hs)
                att_local[sl] = O.attention_head(kc[l][h], vc[l][h], q[sl][None], k[sl][None], v[sl][None], pos)[0]
            att = _gather(att_local)                                                   # ncclAllGather(att_out)
            qa, sa = O.quantize(att, qt)
            db, dn = plan.dim_begin, plan.dim_count
            x1_local = x1[db:db + dn] + O.matmul_q(qt, *_rows(tensors[(ff.T_ATTN_O, l)], db, dn), qa[None], sa[None])[0]
            x1 = _gather(x1_local)                                                     # ncclAllGather(x1)
            qx, sx = O.quantize(O.rmsnorm(x1, tensors[(ff.T_POST_NORM, l)]), qt)
            fb, fn = plan.hidden_begin, plan.hidden_count
            g = O.matmul_q(qt, *_rows(tensors[(ff.T_MLP_GATE, l)], fb, fn), qx[None], sx[None])[0]
            u = O.matmul_q(qt, *_rows(tensors[(ff.T_MLP_UP, l)], fb, fn), qx[None], sx[None])[0]
            hd = _gather(O.swiglu(g, u))                                               # ncclAllGather(hd)
            qh, sh = O.quantize(hd, qt)
            x1_local = x1[db:db + dn] + O.matmul_q(qt, *_rows(tensors[(ff.T_MLP_DOWN, l)], db, dn), qh[None], sh[None])[0]
            x1 = _gather(x1_local)                                                     # ncclAllGather(x1)
        if i == len(tokens) - 1:
            qx, sx = O.quantize(O.rmsnorm(x1, tensors[(ff.T_OUTPUT_NORM, 0)]), qt)
            vb, vn = plan.vocab_begin, plan.vocab_count
            logits = _gather(O.matmul_q(qt, *_rows(tensors[(ff.T_CLASSIFIER, 0)], vb, vn), qx[None], sx[None])[0])   # ncclAllGather(logits)
    return logits


def _worker(rank, world, port, shape, qt, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = synth.make_config(shape, qt)
        tensors = synth.make_tensors(cfg, seed=2024)
        plan = capi.plan_shards(capi.desc_from_config(cfg), rank, world)
        hs = cfg.head_size
        kc = np.zeros((cfg.n_layers, plan.head_count, 1024, hs), np.float32); vc = np.zeros_like(kc)
        prompt = [1, 7, 300, 42, 99]
        res = [sharded_forward(cfg, tensors, plan, prompt, 0, kc, vc)]
        cur, pos = int(np.argmax(res[0])), len(prompt)
        for _ in range(3):
            res.append(sharded_forward(cfg, tensors, plan, [cur], pos, kc, vc)); cur = int(np.argmax(res[-1])); pos += 1
        np.save(os.path.join(out_dir, f"logits_rank{rank}.npy"), np.stack(res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,qt", [("tiny", ff.QT_INT8), ("tiny128", ff.QT_INT16)])
def test_row_sharded_forward_is_bit_identical(tmp_path, shape, qt):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), shape, qt, str(tmp_path)), nprocs=world, join=True)
    cfg = synth.make_config(shape, qt)
    tensors = synth.make_tensors(cfg, seed=2024)
    om = O.OracleModel(cfg, tensors)
    prompt = np.array([1, 7, 300, 42, 99], np.int32)
    ref = [om.forward(prompt, 0)]
    cur, pos = int(np.argmax(ref[0])), len(prompt)
    for _ in range(3):
        ref.append(om.forward(np.array([cur], np.int32), pos)); cur = int(np.argmax(ref[-1])); pos += 1
    ref = np.stack(ref)
    for r in range(world):
        got = np.load(tmp_path / f"logits_rank{r}.npy")
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"rank {r}"
