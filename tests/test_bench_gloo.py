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
