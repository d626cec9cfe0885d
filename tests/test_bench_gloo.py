"""bench.py's multi-process bookkeeping on CPU, world_size 2, gloo: the default N > 1 mode runs one independent
sequence per rank (replicas, no data-path collective) and reports N*K tokens over the MAX wall time of the ranks;
--parallel tp reports K tokens over the same MAX.  Each rank here decodes its own sequence with the CPU oracle."""
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):          # spawned workers import this module without conftest
    if _p not in sys.path:
        sys.path.insert(0, _p)
import __graft_entry__ as _graft  # noqa: E402

_graft.load_package()

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

import bench
import oracle_py as O
from fast_llama_amd import flmfile as ff, synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.make_config("tiny", ff.QT_INT8)
    om = O.OracleModel(cfg, synth.make_tensors(cfg, seed=3))
    prompt = np.array([1, 5 + rank, 9], np.int32)          # replicas: a different sequence on every rank
    cur, pos, ids, steps = int(np.argmax(om.forward(prompt, 0))), 3, [], 6
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        cur = int(np.argmax(om.forward(np.array([cur], np.int32), pos))); pos += 1; ids.append(cur)
    time.sleep(0.05 * (rank + 1))                           # rank 1 is the slow one
    local = time.perf_counter() - t0
    rep, e_rep = bench.job_throughput(local, steps, world, "replicas")
    tp, e_tp = bench.job_throughput(local, steps, world, "tp")
    q.put((rank, local, rep, e_rep, tp, e_tp, ids))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_and_tp_accounting_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps: p.join(timeout=60)
    (r0, l0, rep0, e0, tp0, et0, ids0), (r1, l1, rep1, e1, tp1, et1, ids1) = res
    slow = max(l0, l1)
    assert abs(e0 - slow) < 1e-9 and abs(e1 - slow) < 1e-9 and e0 == et0          # every rank sees the MAX
    assert abs(rep0 - 2 * 6 / slow) < 1e-6 and rep0 == rep1                        # N*K tokens in the slowest rank's time
    assert abs(tp0 - 6 / slow) < 1e-6 and tp0 == tp1                               # K tokens (one sequence) in that time
    assert ids0 != ids1                                                            # the replicas really decoded different sequences


def test_single_process_is_identity():
    assert bench.max_over_ranks(1.25) == 1.25
    v, e = bench.job_throughput(2.0, 128, 1, "single")
    assert v == 64.0 and e == 2.0


def test_pmc_traffic_refuses_a_summary_whose_kernel_is_not_in_the_library(tmp_path):
    """roofline.traffic comes from a committed PMC summary: it is reported only while the kernel the summary names is still a symbol of the
    library the run loaded (round-2 review: a stale file must not be presented as current)"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    assert b._mangled_fragment("k_ffn<2,") == b"5k_ffnILi2E"
    assert b._mangled_fragment("k_gemv<2, 2, 2,") == b"6k_gemvILi2ELi2ELi2E"
    assert b._mangled_fragment("no template") is None
    lib_with = tmp_path / "with.so"; lib_with.write_bytes(b"\0\0_ZN3flm5k_ffnILi2ELi3EEEvNS_8GemvArgsES1_iiPjjPi\0")
    lib_without = tmp_path / "without.so"; lib_without.write_bytes(b"\0\0_ZN3flm6k_gemvILi2ELi2ELi2ELi0ELb0EEEvNS_8GemvArgsE\0")
    got, src, note = b.pmc_traffic(r"k_ffn<2,", str(lib_with))
    assert got and got > 100e6 and src.endswith(".json") and "found" in note        # the committed profiles/r*_pmc_*.json
    got, src, note = b.pmc_traffic(r"k_ffn<2,", str(lib_without))
    assert got is None and src is None and "stale" in note
    got, src, note = b.pmc_traffic(r"k_ffn<2,", str(tmp_path / "missing.so"))
    assert got is None and "cannot read" in note


# ---- the selection of the sharded token's launch structure (bench.run_tp_structures): every structure timed and verified on every rank, the fastest VERIFIED one wins ----
class _FakeTpCtx:
    def __init__(self): self.opts = {"tp_trust_fused": 0, "tp_fuse_ffn": 0}; self.tp_info = {}; self.connects = 0
    def set_option(self, k, v): self.opts[k] = v
    def query(self, k): return 0


def _struct_worker(rank, world, port, q, scenario):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = _FakeTpCtx()

    def fake_connect(capi, c, rank_, world_, device, dist_, torch_, comm_id, first=False):
        c.connects += 1
        tpl = int(bool(c.opts["tp_trust_fused"] and c.opts.get("tp_fuse_layers")))
        per_layer = 9 if not c.opts["tp_trust_fused"] else 0 if tpl else (2 if c.opts["tp_fuse_ffn"] else 3)
        c.tp_info = {"transport": "p2p", "launches_per_sharded_layer": per_layer, "fold_active": int(per_layer != 9), "tp_fuse_attn": 2, "tp_fuse_ffn": c.opts["tp_fuse_ffn"],
                     "tp_fuse_layers": tpl, "tp_fence": c.opts.get("tp_fence") if tpl else None, "granules": (int(bool(c.opts.get("gr_edges"))) if tpl else None)}

    def fake_time(c, cfg, args, prompt, barrier, gold):
        per_layer = c.tp_info["launches_per_sharded_layer"]
        wall = ({9: 0.9, 3: 0.5, 2: 0.4}[per_layer] if per_layer else (0.28 if c.tp_info["granules"] else 0.6 if c.tp_info["tp_fence"] else 0.3)) + 0.01 * rank      # (the rank-spanning launch: slow with fenced flags, faster without, the fastest on granules)
        match = True
        if scenario == "ffn_structure_mismatches_on_rank1" and per_layer == 2 and rank == 1:
            match = False
        if scenario == "fused_raises_on_rank0" and per_layer == 3 and rank == 0:
            raise RuntimeError("timeout in a folded exchange")
        return {"wall_s": wall, "p50_ms": wall * 100, "parity": {"match": match, "first_mismatch": None if match else 3}}

    bench.tp_connect, bench.time_decode = fake_connect, fake_time
    args = type("A", (), {"steps": 10})()
    m, results = bench.run_tp_structures(None, ctx, None, args, None, lambda: None, [1, 2, 3], rank, world, 0, dist, torch)
    q.put((rank, None if m is None else m["wall_s"], [(r["name"], r.get("verified"), r.get("ms_per_step")) for r in results], dict(ctx.opts)))
    dist.barrier()
    dist.destroy_process_group()


def _run_structs(scenario):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_struct_worker, args=(r, 2, port, q, scenario)) for r in range(2)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps: p.join(timeout=60)
    return res


def test_tp_structure_selection_takes_the_fastest_verified_structure_world2():
    (r0, w0, res0, o0), (r1, w1, res1, o1) = _run_structs("all_good")
    assert res0 == res1                                                   # every rank holds the same table (max-over-ranks times, MIN-over-ranks verdicts)
    assert [v for _, v, _ in res0] == [True, True, True, True, True, True]
    assert res0[5][2] == round(1000 * 0.41 / 10, 4)                       # the slowest rank's wall time (FFN13 + FFN2 across ranks: the last structure tried)
    assert abs(w0 - 0.28) < 1e-9 and abs(w1 - 0.29) < 1e-9                # the measurement of the winner: all layers in one rank-spanning launch, on data-tagged granules
    assert o0["tp_trust_fused"] == 1 and o0["tp_fuse_layers"] == 1 and o0["gr_edges"] == 1 and o0["tp_fuse_ffn"] == 0 and o0 == o1     # ... and the group was put back on it


def test_tp_structure_that_mismatches_on_one_rank_is_reported_not_chosen_world2():
    (r0, w0, res0, o0), (r1, w1, res1, o1) = _run_structs("ffn_structure_mismatches_on_rank1")
    assert [v for _, v, _ in res0] == [True, True, True, True, True, False] and res0 == res1
    assert abs(w0 - 0.28) < 1e-9                                          # the rank-spanning launch stays the winner
    assert o0["tp_trust_fused"] == 1 and o0["tp_fuse_ffn"] == 0 and o0["tp_fuse_layers"] == 1 and o0 == o1     # ... and the group was put back on it


def test_tp_structure_that_gives_up_leaves_the_conservative_one_world2():
    (r0, w0, res0, o0), (r1, w1, res1, o1) = _run_structs("fused_raises_on_rank0")
    assert [v for _, v, _ in res0] == [True, False] and res0 == res1     # nothing is built on a structure that failed
    assert abs(w0 - 0.90) < 1e-9 and o0["tp_trust_fused"] == 0 and o0 == o1
