#!/usr/bin/env python3
"""bench.py -- decode tokens/s of the fast-llama per-token hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One step = one greedy decode token of LLaMA2-7B int8 (synthetic weights, configs[2] of BASELINE.json)
through the HIP path (fast-llama_amd/lib/libflm_gpu.so, C ABI include/flm_gpu.h).  All weights, the
KV cache and the decode state are resident in HBM before the timed region; the K timed tokens run
back to back from a hipGraph with no host round trip.
N > 1 (one process per GPU): by default N REPLICAS, each GPU decoding its own sequence -- single-stream
decode is a chain of dependent ~20 us kernels, so splitting one sequence over GPUs buys capacity, not speed
(DESIGN.md section 8); whole-job tokens/s = N*K / max over ranks of the wall time, "scaling": "weak", no
data-path collective.  --parallel tp runs ONE sequence tensor-parallel instead (every matmul split by
output rows + RCCL all-gather of the activations, bit-identical to the single-GPU result; "strong").

Rank 0 prints ONE JSON line; besides the contract's keys it carries
  roofline     : dominant kernel (ffn13 GEMV) algorithmic bytes / its mean launch time measured live
                 with HIP events on the ctx stream, against the 8 TB/s HBM3E peak
  cpu_baseline : the reference's own CPU path (oracle/_ref/main, built from /root/reference by
                 oracle/Makefile) timed on this host's cores on a bounded sample (N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def token_bytes(cfg, pos, esz=1):
    """algorithmic bytes per decoded token: weights + fp32 scales + norms + emb row + fp32 KV."""
    L, dim, hid, V, kvd = cfg.n_layers, cfg.dim, cfg.hidden_dim, cfg.vocab_size, cfg.kv_dim
    eq = L * ((dim + 2 * kvd) * dim + dim * dim + 3 * dim * hid) + V * dim
    return eq * esz + eq / 64 * 4 + (2 * L + 1) * dim * 4 + dim * 4 + 2 * L * kvd * 4 * (pos + 1)


def upload_synthetic(ctx, cfg, seed=20260928, log_every=8):
    """stream the synthetic checkpoint tensor by tensor (peak host RAM = one tensor)."""
    from fast_llama_amd import flmfile as ff, synth
    gs = cfg.quant_group_size
    t0 = time.time()
    rng = np.random.default_rng(seed)
    ctx.upload(ff.T_TOKEN_EMBD, 0, rng.standard_normal((cfg.vocab_size, cfg.dim), dtype=np.float32))
    for l in range(cfg.n_layers):
        lr = np.random.default_rng([seed, 1000 + l])
        ctx.upload(ff.T_INPUT_NORM, l, (0.8 + 0.4 * lr.random(cfg.dim, dtype=np.float32)).astype(np.float32))
        ctx.upload(ff.T_POST_NORM, l, (0.8 + 0.4 * lr.random(cfg.dim, dtype=np.float32)).astype(np.float32))
        for kind, (r, k) in synth.linear_shapes(cfg).items():
            ctx.upload(kind, l, synth._qweights(lr, r, k, cfg.quant_type, gs))
        if l % log_every == 0:
            log(f"  uploaded layer {l}/{cfg.n_layers} ({time.time() - t0:.1f}s)")
    fr = np.random.default_rng([seed, 999])
    ctx.upload(ff.T_OUTPUT_NORM, 0, (0.8 + 0.4 * fr.random(cfg.dim, dtype=np.float32)).astype(np.float32))
    ctx.upload(ff.T_CLASSIFIER, 0, synth._qweights(fr, cfg.vocab_size, cfg.dim, cfg.quant_type, gs))
    log(f"  synthetic checkpoint resident in HBM after {time.time() - t0:.1f}s")


def host_cores():
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True).stdout
        sockets = int(re.search(r"Socket\(s\):\s+(\d+)", out).group(1))
        cps = int(re.search(r"Core\(s\) per socket:\s+(\d+)", out).group(1))
        model = re.search(r"Model name:\s+(.*)", out).group(1).strip()
        return sockets * cps, model
    except Exception:
        return os.cpu_count() or 1, "unknown"


def pmc_traffic(kernel_regex):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary under profiles/
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE x2 is the gfx950 correction of
    MI355X_MICROARCH.md; counters are in KiB).  None when no summary is there."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        for name, c in d.items():
            if re.search(kernel_regex, name) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return int((2 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024), os.path.basename(f)
    return None, None


def cpu_baseline(cfg, budget_s=40.0):
    """Reference CPU path on a bounded sample: the reference binary decodes synthetic 7B-WIDTH models with
    4 and 12 layers (same tensor shapes as the 32-layer model, so the same per-layer and classifier
    work per token); per-token time is t(L) = t_cls + L * t_layer, fitted from the two runs and
    evaluated at L = 32.  Falls back to the C restatement (kind "port") if the binary cannot run."""
    from fast_llama_amd import flmfile as ff, synth
    import copy
    cores, model = host_cores()
    threads = max(1, min(cores, 64))
    ref_main = os.path.join(ROOT, "oracle", "_ref", "main")
    res = {"unit": "tokens/s", "cores": threads, "host": model}

    def run_ref(L, n_tokens):
        c = copy.copy(cfg); c.n_layers = L; c.name = f"synthetic-7Bwidth-L{L}"
        path = f"/tmp/flm-bench-L{L}.flm"
        tensors = synth.make_tensors(c, seed=7, share_layers=True)
        ff.write_flm(path, c, synth.make_tokenizer(c.vocab_size), tensors)
        del tensors
        cmd = ["timeout", "300", ref_main, "-c", path, "-j", str(threads), "-q", "int8", "-n", str(n_tokens), "-t", "0",
               "--mode", "bm", "--rounds", "1", "--uma", "-i", "the shape of it"]
        t0 = time.time()
        out = subprocess.run(cmd, capture_output=True, text=True)
        os.remove(path)
        m = re.search(r"output_token_latancy:(?:\x1b\[[0-9;]*m)?\s*([0-9.]+)", out.stdout)
        if out.returncode != 0 or not m:
            raise RuntimeError(f"reference binary failed rc={out.returncode}: {out.stdout[-300:]} {out.stderr[-300:]}")
        return float(m.group(1)), time.time() - t0

    if os.path.exists(ref_main):
        try:
            import shutil
            free_gb = shutil.disk_usage("/tmp").free / 1e9
            need_gb = token_bytes(cfg, 0) / 1e9 * 1.15
            if free_gb > need_gb + 2:
                # the whole model: the reference binary decodes the same 32-layer shape, no extrapolation
                ntok = 160
                t_tok, wall = run_ref(cfg.n_layers, ntok)
                res.update(value=1000.0 / t_tok, kind="reference",
                           sample=(f"reference binary (oracle/_ref/main, -O3 -march=x86-64-v3 -mfma, AVX2 kernels) -j {threads} -t 0 --mode bm --uma, the full "
                                   f"{cfg.n_layers}-layer LLaMA2-7B-shaped int8 .flm (synthetic weights, identical tensors in every layer), prompt 13 tokens + {ntok} "
                                   f"decode tokens: {t_tok:.2f} ms per output token (wall {wall:.0f}s incl. writing and loading the file)"))
                return res
            La, Lb, ntok = 4, 12, 48
            t2, w2 = run_ref(La, ntok)
            t4, w4 = run_ref(Lb, ntok)
            t_layer = max((t4 - t2) / (Lb - La), 1e-6)
            t_cls = max(t2 - La * t_layer, 0.0)
            t_tok = t_cls + cfg.n_layers * t_layer
            res.update(value=1000.0 / t_tok, kind="reference",
                       sample=(f"reference binary (oracle/_ref/main, -O3 -march=x86-64-v3 -mfma, AVX2 kernels) -j {threads} -t 0 --mode bm, int8 .flm, "
                               f"7B-width synthetic models with {La} and {Lb} layers, {ntok} decode tokens each: {t2:.2f} / {t4:.2f} ms per token; "
                               f"t_layer={t_layer:.3f} ms, t_cls={t_cls:.3f} ms, extrapolated to 32 layers = {t_tok:.1f} ms/token "
                               f"(wall {w2 + w4:.0f}s; /tmp too small for the full model)"))
            return res
        except Exception as e:  # noqa: BLE001
            log("cpu_baseline: reference binary unusable here:", e)
    # fallback: the C restatement of the reference (OpenMP over rows), same bounded-sample scheme
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as O

    def run_port(L, n_tokens):
        c = copy.copy(cfg); c.n_layers = L
        tensors = synth.make_tensors(c, seed=7, share_layers=True)
        om = O.OracleModel(c, tensors)
        om.forward(np.array([1, 5, 9], np.int32), 0)
        t0 = time.time()
        for i in range(n_tokens):
            om.forward(np.array([7], np.int32), 3 + i)
        return (time.time() - t0) * 1000.0 / n_tokens

    t1 = run_port(1, 4); t3 = run_port(3, 4)
    t_layer = max((t3 - t1) / 2.0, 1e-6); t_cls = max(t1 - t_layer, 0.0)
    t_tok = t_cls + cfg.n_layers * t_layer
    res.update(value=1000.0 / t_tok, kind="port", cores=os.cpu_count() or 1,
               sample=f"oracle/flm_oracle.c (C restatement, OpenMP) 7B-width models with 1 and 3 layers, 4 tokens each, extrapolated to 32 layers = {t_tok:.1f} ms/token")
    return res


def max_over_ranks(x: float) -> float:
    """MAX of a per-rank scalar over the process group (1-process runs: identity)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(x)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(local_elapsed_s: float, steps: int, world: int, mode: str):
    """whole-job tokens/s from each rank's own wall time for its K timed steps.
    replicas: every rank decoded K tokens of its own sequence  -> N*K tokens in max(elapsed)   (weak scaling)
    tp      : all ranks decoded the same K tokens together     ->   K tokens in max(elapsed)   (strong scaling)"""
    elapsed = max_over_ranks(local_elapsed_s)
    tokens = steps * (world if mode == "replicas" else 1)
    return tokens / elapsed, elapsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--shape", default="7B")
    ap.add_argument("--quant", default="int8", choices=["int8", "int16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--wg-per-cu", type=int, default=0)
    ap.add_argument("--prompt-len", type=int, default=9)
    ap.add_argument("--parallel", default="replicas", choices=["replicas", "tp"],
                    help="N > 1: 'replicas' = one independent sequence per GPU, no data-path collective (default); "
                         "'tp' = one sequence, every matmul split by output rows over the GPUs, RCCL all-gathers")
    ap.add_argument("--mega", type=int, default=-1, help="1/0: force the persistent whole-token kernel on/off (default: library default)")
    args = ap.parse_args()

    import torch
    graft.load_package()
    from fast_llama_amd import capi, flmfile as ff, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    comm_id = None
    mode = "single" if world == 1 else args.parallel
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if mode == "tp":
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            comm_id = bytes(idt.cpu().numpy().tobytes())
    tp = world if mode == "tp" else 1

    qt = ff.QT_INT8 if args.quant == "int8" else ff.QT_INT16
    cfg = synth.make_config(args.shape, qt)
    if rank == 0:
        log(f"bench: {args.shape} {args.quant}, world={world} ({mode}), steps={args.steps}, warmup={args.warmup}")
    ctx = capi.Ctx(capi.desc_from_config(cfg), device=local_rank, rank=rank if tp > 1 else 0, world=tp, comm_id=comm_id)
    if args.wg_per_cu:
        ctx.set_option("wg_per_cu", args.wg_per_cu)
    if args.mega >= 0:
        ctx.set_option("use_mega", args.mega)
    upload_synthetic(ctx, cfg)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # prompt (untimed): BOS + 8 tokens, then W warm-up decode steps (also captures the hipGraphs)
    V = cfg.vocab_size
    seq = rank if mode == "replicas" else 0            # replicas decode different sequences
    prompt = np.array([1] + [int(x) for x in ((np.arange(1, args.prompt_len) + 131 * seq) * 7919) % V], dtype=np.int32)
    first = ctx.forward_argmax(prompt, 0)
    pos = len(prompt)
    if args.warmup > 0:
        ids = ctx.decode_greedy(first, pos, args.warmup)
        first = int(ids[-1]); pos += args.warmup
    barrier()
    t0 = time.perf_counter()
    ms_dev = ctx.decode_timed(first, pos, args.steps)      # enqueues EXACTLY K tokens and waits for the last one
    torch.cuda.synchronize()
    barrier_t = time.perf_counter() - t0
    barrier()
    tok_s, elapsed = job_throughput(barrier_t, args.steps, world, mode)
    mid_pos = pos + args.steps // 2
    esz = 1 if qt == ff.QT_INT8 else 2

    # per-kernel times, live, HIP events on the ctx stream (eager launches, same kernels as the graph)
    kt = ctx.kernel_times(mid_pos, iters=3)
    dom = "ffn13"
    dom_us, dom_cnt = kt[dom]
    dom_bytes = ctx.kernel_bytes(dom, mid_pos)
    achieved = dom_bytes / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
    kernels = {k: {"us": round(v[0], 2), "per_token": v[1], "GBps": round(ctx.kernel_bytes(k, mid_pos) / (v[0] * 1e-6) / 1e9, 1) if v[0] > 0 else 0.0}
               for k, v in kt.items() if v[1] > 0}

    traffic, traffic_src = pmc_traffic(r"k_gemv<2, 2, 2," if qt == ff.QT_INT8 else r"k_gemv<1, 2, 2,")
    if rank == 0:
        line = {
            "metric": "decode tokens/s LLaMA2-7B int8" if args.shape == "7B" and qt == ff.QT_INT8 else f"decode tokens/s {args.shape} {args.quant}",
            "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "strong" if mode == "tp" else "weak", "vs_baseline": None, "dtype": "int8" if qt == ff.QT_INT8 else "int16", "data": "synthetic",
            "config": {"workload": f"LLaMA2-{args.shape} {args.quant} .flm-layout synthetic weights, single-stream greedy decode, "
                                   f"prompt {len(prompt)} tokens, positions {pos}..{pos + args.steps - 1}, fp32 KV cache, max_seq 1024",
                       "parallelism": {"single": "single-gpu", "tp": f"tp{world}: one sequence, matmuls split by output rows, RCCL all-gathers",
                                       "replicas": f"{world} replicas: one independent sequence per GPU, no data-path collective"}[mode],
                       "device_ms_per_step": round(ms_dev / args.steps, 4)},
            # per GPU: bytes each GPU streams per token it works on, at the rate it produces them
            "token_roofline": {"bytes_per_token": int(token_bytes(cfg, mid_pos, esz) / tp), "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                               "achieved": round(token_bytes(cfg, mid_pos, esz) / tp * (tok_s / (world / tp)) / 1e9, 1),
                               "frac": round(token_bytes(cfg, mid_pos, esz) / tp * (tok_s / (world / tp)) / 1e9 / HBM_PEAK_GBS, 4)},
            "roofline": {"kernel": "k_gemv<int8,rmsnorm+quantize,swiglu> (ffn13)" if qt == ff.QT_INT8 else "k_gemv<int16,...> (ffn13)",
                         "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": int(dom_bytes), "avg_launch_us": round(dom_us, 2), "launches_per_token": dom_cnt},
            "kernels": kernels,
            "kernels_note": "per-class times of the stand-alone kernels (back-to-back launches); the decode loop runs attention + attn_o as one launch (k_attn_o) on a single GPU",
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
