#!/usr/bin/env python3
"""bench.py -- decode tokens/s of the fast-llama per-token hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--pos P]      (N > 1: launched by torch.distributed.run, one process per GPU)
  python bench.py --config prefill512-int16                     (BASELINE config 5: the batched prompt path)

One step = one greedy decode token of LLaMA2-7B int8 (synthetic weights, configs[2] of BASELINE.json) through the HIP path
(fast-llama_amd/lib/libflm_gpu.so, C ABI include/flm_gpu.h).  All weights, the KV cache and the decode state are resident in HBM
before the timed region; the K timed tokens run back to back from a hipGraph with no host round trip.  The checkpoint is the
portable splitmix64 one of fast_llama_amd/synth.py (norm weights 1.0), the same tensors the reference decoded when
tests/golden/model_7B_int8_L32.npz was made -- the line's `parity` field says whether the ids decoded here equal the reference's.

N > 1: the headline `value` is ONE sequence sharded over the N GPUs ("scaling": "strong"): every matmul split by output rows
(bit-identical to one GPU), activation slices exchanged peer to peer over xGMI (RCCL all-gathers if the peer mapping fails); the
replicas figure (N independent sequences, no data-path collective, "weak") is reported beside it in `replicas`.  If the sharded
run fails or decodes other ids than the reference on any rank, the line falls back to replicas and says so in `tp_note`.
--parallel replicas measures the replicas only.

Rank 0 prints ONE JSON line; besides the contract's keys it carries
  roofline       : dominant kernel (ffn13 GEMV) algorithmic bytes / its mean launch time measured live with HIP events on the ctx
                   stream, against the 8 TB/s HBM3E peak; traffic from the committed PMC summary under profiles/
  token_roofline : algorithmic bytes per token (weights + scales + norms + KV rows at the measured positions) x tokens/s vs 8 TB/s
  parity         : ids decoded in this run vs the reference CPU path's (golden fixture), and whether a replay gave the same ids
  p50/p90        : per-token times of a second pass with one event per token
  cpu_baseline   : the reference's own CPU path (oracle/_ref/main, built from /root/reference by oracle/Makefile) timed on this
                   host's cores on a bounded sample (N = 1 only)
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


def upload_synthetic(ctx, cfg, log_every=8, threads=None):
    """stream the PORTABLE synthetic checkpoint (fast_llama_amd/synth.py: every value a splitmix64 function of tensor id and
    element index, norm weights 1.0 -- SURVEY.md 8d; the same tensors tests/golden/make_golden_r2.py fed to the reference)
    tensor by tensor: a few generator threads run ahead of the upload, peak host RAM = a few tensors."""
    from concurrent.futures import ThreadPoolExecutor
    from fast_llama_amd import flmfile as ff, synth
    t0 = time.time()
    jobs = [((ff.T_TOKEN_EMBD, 0), lambda: synth.portable_embedding(cfg.vocab_size, cfg.dim))]
    ones = np.ones(cfg.dim, np.float32)
    for l in range(cfg.n_layers):
        jobs.append(((ff.T_INPUT_NORM, l), lambda: ones)); jobs.append(((ff.T_POST_NORM, l), lambda: ones))
        for kind, (r, k) in synth.linear_shapes(cfg).items():
            jobs.append(((kind, l), lambda kind=kind, l=l, r=r, k=k: synth.portable_qweights(kind, l, r, k, cfg.quant_type)))
    jobs.append(((ff.T_OUTPUT_NORM, 0), lambda: ones))
    jobs.append(((ff.T_CLASSIFIER, 0), lambda: synth.portable_qweights(ff.T_CLASSIFIER, 0, cfg.vocab_size, cfg.dim, cfg.quant_type)))
    nthreads = threads or max(1, min(16, (os.cpu_count() or 2) - 1))
    with ThreadPoolExecutor(nthreads) as ex:
        window, it = [], iter(jobs)
        def push():
            j = next(it, None)
            if j is not None:
                window.append((j[0], ex.submit(j[1])))
        for _ in range(nthreads + 2):
            push()
        while window:
            (kind, layer), fut = window.pop(0)
            ctx.upload(kind, layer, fut.result())
            push()
            if kind == ff.T_MLP_DOWN and layer % log_every == 0:
                log(f"  uploaded layer {layer}/{cfg.n_layers} ({time.time() - t0:.1f}s)")
    log(f"  synthetic checkpoint resident in HBM after {time.time() - t0:.1f}s")


def golden_ids(cfg, qt, prompt_len):
    """the reference's greedy ids for this exact model and prompt (fixtures under tests/golden/, produced by make_golden_r2.py / make_golden_r4.py from
    oracle/_ref/libflref.so), or None when there is no fixture for the configuration.  prompt_len 9: bench.py's short prompt; 512: the long one."""
    from fast_llama_amd import flmfile as ff
    gd = os.path.join(ROOT, "tests", "golden")
    full7b = cfg.name.endswith("7B") and cfg.n_layers == 32
    if full7b and qt == ff.QT_INT8 and prompt_len == 9: path, key = "model_7B_int8_L32.npz", "ids"
    elif full7b and qt == ff.QT_INT8 and prompt_len == 512: path, key = "model_7B_int8_L32_p512.npz", "ids"
    elif full7b and qt == ff.QT_INT16 and prompt_len == 9: path, key = "model_7B_int16_L32.npz", "p9_ids"
    elif full7b and qt == ff.QT_INT16 and prompt_len == 512: path, key = "model_7B_int16_L32.npz", "p512_ids"
    elif cfg.name.endswith("1.3B") and cfg.n_layers == 4 and qt == ff.QT_INT8 and prompt_len == 9: path, key = "model_1p3B_int8.npz", "ids"
    else: return None
    if not os.path.exists(os.path.join(gd, path)):
        return None
    golden_ids.last = f"tests/golden/{path}"
    return [int(x) for x in np.load(os.path.join(gd, path))[key]]


def host_cores():
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True).stdout
        sockets = int(re.search(r"Socket\(s\):\s+(\d+)", out).group(1))
        cps = int(re.search(r"Core\(s\) per socket:\s+(\d+)", out).group(1))
        model = re.search(r"Model name:\s+(.*)", out).group(1).strip()
        host_cores.sockets = sockets
        return sockets * cps, model
    except Exception:
        host_cores.sockets = None
        return os.cpu_count() or 1, "unknown"


def _mangled_fragment(kernel_regex):
    """'k_ffn<2,' -> b'5k_ffnILi2E', 'k_gemv<2, 2, 2,' -> b'6k_gemvILi2ELi2ELi2E': the Itanium-mangled head of a kernel template instantiation
    whose first arguments are small integers (all that the PMC summaries' names are matched by)"""
    m = re.match(r"(\w+)<([\d, ]*)", kernel_regex.replace("\\", ""))
    if not m:
        return None
    name, args = m.group(1), [a for a in m.group(2).replace(" ", "").split(",") if a]
    return (f"{len(name)}{name}I" + "".join(f"Li{a}E" for a in args)).encode()


def pmc_traffic(kernel_regex, lib_path=None):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary under profiles/
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE x2 is the gfx950 correction of
    MI355X_MICROARCH.md; counters are in KiB).  None when no summary is there -- or when the kernel the summary names is no longer in the
    library this run loaded (a stale summary must not be reported as current): returns (bytes, file, note)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")))
    frag = _mangled_fragment(kernel_regex)
    if lib_path and frag:
        try:
            with open(lib_path, "rb") as f:
                if frag not in f.read():
                    return None, None, f"no kernel matching {kernel_regex!r} ({frag.decode()}) in {os.path.basename(lib_path)}: PMC summaries under profiles/ are stale"
        except OSError as e:
            return None, None, f"cannot read {lib_path}: {e}"
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        for name, c in d.items():
            if re.search(kernel_regex, name) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return int((2 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024), os.path.basename(f), "kernel symbol found in the loaded library"
    return None, None, "no PMC summary for this kernel under profiles/"


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
    res = {"unit": "tokens/s", "cores": threads, "host": model, "host_sockets": getattr(host_cores, "sockets", None), "host_physical_cores_total": cores,
           "host_logical_cpus": os.cpu_count()}
    # what the number is: the reference's own binary, timed on a checkpoint written for TIMING -- the same tensor shapes, synthetic int8 weights, but every layer holds
    # identical tensors (7 GB generated once instead of 32 times), so its ids are not the GPU checkpoint's; parity is pinned elsewhere (tests/golden, tests/test_gpu_configs.py)
    kind_detail = "reference binary, timing-only checkpoint (identical layers)"

    def run_ref(L, n_tokens, nthreads=None, keep=False, reuse=False):
        nthreads = nthreads or threads
        c = copy.copy(cfg); c.n_layers = L; c.name = f"synthetic-7Bwidth-L{L}"
        path = f"/tmp/flm-bench-L{L}.flm"
        if not (reuse and os.path.exists(path)):
            tensors = synth.make_tensors(c, seed=7, share_layers=True)
            ff.write_flm(path, c, synth.make_tokenizer(c.vocab_size), tensors)
            del tensors
        cmd = ["timeout", "300", ref_main, "-c", path, "-j", str(nthreads), "-q", "int8", "-n", str(n_tokens), "-t", "0",
               "--mode", "bm", "--rounds", "1", "--uma", "-i", "the shape of it"]
        t0 = time.time()
        out = subprocess.run(cmd, capture_output=True, text=True)
        if not keep:
            os.remove(path)
        m = re.search(r"output_token_latancy:(?:\x1b\[[0-9;]*m)?\s*([0-9.]+)", out.stdout)
        if out.returncode != 0 or not m:
            if os.path.exists(path):
                os.remove(path)
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
                t_tok, wall = run_ref(cfg.n_layers, ntok, keep=True)
                try:   # the reference's own README quotes -j 8 (/root/reference README.md:96-100): the same file once more on 8 threads, fewer tokens
                    t8, wall8 = run_ref(cfg.n_layers, 24, nthreads=8, reuse=True)
                    res["j8"] = {"value": round(1000.0 / t8, 2), "unit": "tokens/s", "cores": 8, "sample": f"the same file, -j 8, 24 decode tokens: {t8:.1f} ms per output token (wall {wall8:.0f}s)"}
                except Exception as e:  # noqa: BLE001
                    res["j8"] = {"value": None, "sample": f"failed: {e}"}
                    if os.path.exists(f"/tmp/flm-bench-L{cfg.n_layers}.flm"):
                        os.remove(f"/tmp/flm-bench-L{cfg.n_layers}.flm")
                res.update(value=1000.0 / t_tok, kind="reference", kind_detail=kind_detail,
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
            res.update(value=1000.0 / t_tok, kind="reference", kind_detail=kind_detail + ", 4- and 12-layer models extrapolated to 32",
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


def time_decode(ctx, cfg, args, prompt, barrier, gold):
    """prompt (untimed) -> greedy decode up to the start position (untimed; at least W steps, which also capture the
    hipGraphs) -> EXACTLY K timed steps between barriers -> the ids for the parity field, and a second pass with an event
    per token for the median.  Returns a dict of measurements."""
    import torch
    # Every rank runs BOTH barriers whatever happens on it (a rank that raised before its barriers would leave the others in theirs while
    # it goes on to the caller's all-reduce: mismatched collectives, a hang instead of the fall-back to replicas); the error is re-raised
    # behind the second barrier.
    err = None
    first, ids, pos, ms_dev, wall = 0, [], len(prompt), 0.0, 0.0
    try:
        first = ctx.forward_argmax(prompt, 0)
        ids = [int(first)]
        n_pre = max(args.warmup, (args.pos - pos) if args.pos is not None else 0)
        if n_pre > 0:
            pre = ctx.decode_greedy(first, pos, n_pre)
            ids += [int(x) for x in pre]; first = int(pre[-1]); pos += n_pre
        if pos + args.steps > 1024:
            raise RuntimeError(f"bench.py: positions {pos}..{pos + args.steps - 1} exceed max_seq_len 1024")
    except Exception as e:  # noqa: BLE001
        err = e
    barrier()
    t0 = time.perf_counter()
    if err is None:
        try:
            ms_dev = ctx.decode_timed(first, pos, args.steps)      # enqueues EXACTLY K tokens and waits for the last one
        except Exception as e:  # noqa: BLE001
            err = e
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    barrier()
    if err is not None:
        raise err
    ids += [int(x) for x in ctx.last_tokens(args.steps)]
    each = ctx.decode_timed_each(first, pos, args.steps)    # same tokens again (same cache rows), one event per token
    again = [int(x) for x in ctx.last_tokens(args.steps)]
    parity = {"against": None, "ids_checked": 0, "match": None, "replay_identical": again == ids[-args.steps:]}
    if gold is not None:
        n = min(len(ids), len(gold))
        mism = next((i for i in range(n) if ids[i] != gold[i]), None)
        parity.update(against="greedy ids of the reference CPU path (oracle/_ref/libflref.so, ParallelTransformer::forward) on this model and prompt: "
                              f"{getattr(golden_ids, 'last', 'tests/golden')}; the same fixture's logits digests are checked bit for bit by tests/test_gpu_configs.py",
                      ids_checked=n, match=mism is None, first_mismatch=mism)
    return {"wall_s": wall, "ms_dev": ms_dev, "pos": pos, "ids": ids, "parity": parity,
            "p50_ms": float(np.median(each)), "p90_ms": float(np.percentile(each, 90)), "each_mean_ms": float(np.mean(each))}


def prefill_main(args):
    """--config prefill<N>-<quant>: BASELINE.json's config 5 (LLaMA2-7B int16 + a 512-token prompt): the batched prompt path --
    int8 / int16 GEMM tiles on the matrix cores, QK^T and softmax x V on fp32 MFMA -- timed as whole forwards of the prompt (K
    repetitions on a cleared cache), reported as prompt tokens/s with the linear layers' MAC rate beside it; the next token is
    checked against the token-by-token decode path over the same prompt (the path the golden-logit tests pin to the reference)."""
    import re as _re
    import torch
    graft.load_package()
    from fast_llama_amd import capi, flmfile as ff, synth
    m = _re.fullmatch(r"prefill(\d+)-(int8|int16)", args.config)
    if not m:
        sys.exit("bench.py --config: expected prefill<N>-int8|int16, e.g. prefill512-int16")
    n, qt = int(m.group(1)), (ff.QT_INT8 if m.group(2) == "int8" else ff.QT_INT16)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    cfg = synth.make_config(args.shape, qt)
    ctx = capi.Ctx(capi.desc_from_config(cfg), device=0)
    upload_synthetic(ctx, cfg)
    V = cfg.vocab_size
    prompt = np.array([1] + [int(x) for x in (np.arange(1, n) * 7919) % V], dtype=np.int32)
    for _ in range(max(1, args.warmup)):
        ctx.reset_kv(); tok = ctx.forward_argmax(prompt, 0)
    times = []
    for _ in range(max(1, args.steps)):
        ctx.reset_kv(); ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter(); tok = ctx.forward_argmax(prompt, 0); times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    ctx.set_option("use_prefill", 0); ctx.reset_kv()
    tok_ref = ctx.forward_argmax(prompt, 0)                       # the same prompt, one token at a time through the decode kernels
    ctx.set_option("use_prefill", 1)
    L, dim, hid = cfg.n_layers, cfg.dim, cfg.hidden_dim
    macs = (n - 1) * ((L - 1) * (4 * dim * dim + 3 * dim * hid) + 3 * dim * dim)      # the batch: every layer but the last in full, the last one's q/k/v only
    flops_qk = 2.0 * (L - 1) * cfg.n_heads * cfg.head_size * sum(range(1, n))              # causal QK^T (the fp32-MFMA kernel)
    pgold = golden_ids(cfg, qt, n)
    line = {"metric": f"prefill tokens/s LLaMA2-{args.shape} {m.group(2)}, {n}-token prompt", "value": round(n / dt, 1), "unit": "tokens/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": m.group(2), "data": "synthetic",
            "config": {"workload": f"LLaMA2-{args.shape} {m.group(2)}, {n}-token prompt through the batched path (GEMM tiles on v_mfma_i32_32x32x32_i8, int16 as hi/lo byte planes; "
                                   f"QK^T and softmax x V on v_mfma_f32_16x16x4_f32; last token through the decode kernels), next token {int(tok)}"},
            "parity": {"against": "next token of the token-by-token decode path over the same prompt in the same session (that path's logits are pinned to the "
                                  "reference by tests/golden/model_7B_int8_L32.npz; batched vs token-by-token cache rows and logits bit for bit: tests/test_gpu_model.py)",
                       "match": bool(int(tok) == int(tok_ref)),
                       "reference_next_token": (None if pgold is None else {"fixture": golden_ids.last, "match": int(tok) == pgold[0]})},
            "linear_layers": {"int_macs": int(macs), "TMAC_per_s_over_whole_forward": round(macs / dt / 1e12, 1), "note": "lower bound: the forward's whole wall time is charged to the GEMMs"},
            "qk_flops": int(flops_qk)}
    print(json.dumps(line), flush=True)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None, help="prefill<N>-int8|int16: time the batched prompt path instead of decode (BASELINE config 5: prefill512-int16)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--shape", default="7B")
    ap.add_argument("--quant", default="int8", choices=["int8", "int16"])
    ap.add_argument("--no-config5", action="store_true", help="skip the int16 512-token prefill object (a second 13.5 GB checkpoint) of the default line")
    ap.add_argument("--pos", type=int, default=None, help="start the timed steps at this position (the prompt's greedy continuation is decoded, untimed, up to it); "
                                                          "default: prompt length + warmup")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode128", action="store_true", help="skip the decode_128 object (rocprofv3 passes: per-kernel averages / counters then cover the driver's positions only, as roofline.avg_launch_us does)")
    ap.add_argument("--wg-per-cu", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="k=v: flm_set_option on the decode context (launch-structure A/B runs; reported in config.options; none by default)")
    ap.add_argument("--prompt-len", type=int, default=9)
    ap.add_argument("--parallel", default="tp", choices=["replicas", "tp"],
                    help="N > 1: 'tp' (default) = ONE sequence, every matmul split by output rows over the GPUs (strong scaling, the headline value; "
                         "the replicas figure is reported beside it); 'replicas' = one independent sequence per GPU only")
    args = ap.parse_args()
    if args.config:
        return prefill_main(args)

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
    # FLM_BENCH_FORCE_DEVICE=0: every rank on the same GPU (a 1-GPU box exercising the multi-process path: gloo bootstrap, IPC-mapped
    # peers); the normal case is one GPU per rank and the nccl backend
    force_dev = os.environ.get("FLM_BENCH_FORCE_DEVICE")
    device = int(force_dev) if force_dev is not None else local_rank
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dev is not None:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    qt = ff.QT_INT8 if args.quant == "int8" else ff.QT_INT16
    esz = 1 if qt == ff.QT_INT8 else 2
    cfg = synth.make_config(args.shape, qt)
    V = cfg.vocab_size
    gold = golden_ids(cfg, qt, args.prompt_len)

    def prompt_for(seq):
        return np.array([1] + [int(x) for x in ((np.arange(1, args.prompt_len) + 131 * seq) * 7919) % V], dtype=np.int32)

    if rank == 0:
        log(f"bench: {args.shape} {args.quant}, world={world}, steps={args.steps}, warmup={args.warmup}")

    # ---- the headline run: single GPU, or ONE sequence tensor-parallel over all ranks -----------------------------------
    mode = "single" if world == 1 else args.parallel
    tp_note = None
    m = None
    tp_structures = None
    if mode == "tp":
        try:
            ctx = open_tp_ctx(capi, cfg, rank, world, device, dist, torch)
            upload_synthetic(ctx, cfg)
            m, tp_structures = run_tp_structures(capi, ctx, cfg, args, prompt_for(0), barrier, gold, rank, world, device, dist, torch)
            if m is None:      # a sharded run that decodes other ids than the reference (or gives up) under EVERY structure is not a result: fall back, say so
                raise RuntimeError("no launch structure of the sharded token was verified: " + "; ".join(f"{r['name']}: {r.get('error')}" for r in tp_structures))
            ok = 1
        except Exception as e:  # noqa: BLE001
            log(f"rank {rank}: tensor-parallel run failed: {e}")
            tp_note = f"tensor-parallel run failed on rank {rank}: {e}"
            ok = 0
        okt = torch.tensor([ok], device="cuda" if dist.get_backend() == "nccl" else "cpu"); dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 0:
            mode, m = "replicas", None
            tp_note = tp_note or "tensor-parallel run failed on another rank"
    if mode != "tp":
        ctx = capi.Ctx(capi.desc_from_config(cfg), device=device)
        if args.wg_per_cu:
            ctx.set_option("wg_per_cu", args.wg_per_cu)
        for kv in args.opt:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        upload_synthetic(ctx, cfg)
        m = time_decode(ctx, cfg, args, prompt_for(rank if mode == "replicas" else 0), barrier, gold if (mode == "single" or rank == 0) else None)
    tok_s, elapsed = job_throughput(m["wall_s"], args.steps, world, mode)
    pos = m["pos"]
    mid_pos = pos + args.steps // 2
    tp = world if mode == "tp" else 1

    # a second operating point, outside the timed region of `value`: the same decode from position 512 ("also report at
    # pos ~ 512"): a 512-token prompt through the batched kernels, 4 untimed steps (graph capture), then K timed steps
    long_ctx = None
    if mode == "single" and args.pos is None and rank == 0:
        try:
            lp = np.array([1] + [int(x) for x in (np.arange(1, 512) * 7919) % V], dtype=np.int32)
            lf = ctx.forward_argmax(lp, 0)
            lw = ctx.decode_greedy(lf, 512, 4)
            lms = ctx.decode_timed(int(lw[-1]), 516, args.steps)
            lids = [int(lf)] + [int(x) for x in lw] + [int(x) for x in ctx.last_tokens(args.steps)]
            lgold = golden_ids(cfg, qt, 512)
            lpar = None
            if lgold is not None:
                nchk = min(len(lids), len(lgold))
                lpar = {"against": f"the reference's greedy ids behind the same 512-token prompt ({golden_ids.last})", "ids_checked": nchk, "match": lids[:nchk] == lgold[:nchk]}
            lb = token_bytes(cfg, 516 + args.steps // 2, esz)
            long_ctx = {"positions": f"516..{515 + args.steps}", "ms_per_step": round(lms / args.steps, 4), "tokens_per_s": round(args.steps / (lms / 1e3), 2),
                        "bytes_per_token": int(lb), "token_roofline_frac": round(lb * (args.steps / (lms / 1e3)) / 1e9 / HBM_PEAK_GBS, 4),
                        "parity": lpar, "note": "device time of K steps between HIP events on the ctx stream; not part of `value`"}
        except Exception as e:  # noqa: BLE001
            long_ctx = {"error": str(e)}
    # SURVEY.md 8d's decode protocol beside the driver's K timed steps (20 steps at positions 14..33 flatter the token by ~3 %: fewer cache rows): prompt = BOS + 8 tokens, 128 greedy
    # steps with one event per token, the first 8 discarded, mean and p50 over the other 120 (positions 17..136), with that span's own roofline fraction.  Outside the timed region of `value`.
    decode_128 = None
    if mode == "single" and args.pos is None and rank == 0 and not args.no_decode128:
        try:
            p9 = prompt_for(0)
            ctx.reset_kv(); f9 = ctx.forward_argmax(p9, 0)
            d8 = ctx.decode_greedy(f9, len(p9), 8)                                   # the 8 discarded steps
            p17 = len(p9) + 8
            ctx.decode_greedy(int(d8[-1]), p17, 120)                                 # (untimed pass over the span: every chunk graph it replays has been launched once)
            ms120 = ctx.decode_timed(int(d8[-1]), p17, 120)                          # the mean: device time of the 120 tokens between two HIP events (chunk graphs, as the headline)
            i128 = [int(f9)] + [int(x) for x in d8] + [int(x) for x in ctx.last_tokens(120)]
            e120 = np.asarray(ctx.decode_timed_each(int(d8[-1]), p17, 120), dtype=np.float64)    # the p50: one event per token (single-token graphs + ~3 us of event per token)
            midp = p17 + 60
            b128 = token_bytes(cfg, midp, esz)
            par128 = None
            if gold is not None:
                nchk = min(len(i128), len(gold))
                par128 = {"ids_checked": nchk, "match": i128[:nchk] == gold[:nchk]}
            decode_128 = {"protocol": "SURVEY.md 8d: BOS + 8 prompt tokens, 128 greedy steps, first 8 discarded", "positions": f"{p17}..{p17 + 119}",
                          "tokens_per_s_mean": round(120.0 / (ms120 / 1e3), 2), "tokens_per_s_p50": round(1e3 / float(np.median(e120)), 2),
                          "ms_per_step_mean": round(ms120 / 120.0, 4), "ms_per_step_p50": round(float(np.median(e120)), 4),
                          "bytes_per_token": int(b128), "token_roofline_frac": round(b128 * (120.0 / (ms120 / 1e3)) / 1e9 / HBM_PEAK_GBS, 4),
                          "parity": par128, "note": "mean: device time of the 120 kept tokens between two HIP events on the ctx stream; p50: one event per token (adds a few us to each); not part of `value`"}
        except Exception as e:  # noqa: BLE001
            decode_128 = {"error": str(e)}
    # a third operating point, outside the timed region of `value`: BASELINE config 5's prompt path at this run's quant type -- a 512-token prompt through
    # the batched kernels (int8 GEMM tiles on the matrix cores, fp32-MFMA QK^T / PV), median of 3 forwards on a cleared cache
    prefill = None
    if mode == "single" and args.pos is None and rank == 0:
        try:
            lp = np.array([1] + [int(x) for x in (np.arange(1, 512) * 7919) % V], dtype=np.int32)
            ctx.reset_kv(); ptok = ctx.forward_argmax(lp, 0)               # warm-up (group-major scale copies, first launches)
            pgold = golden_ids(cfg, qt, 512)
            pts = []
            for _ in range(3):
                ctx.reset_kv(); ctx.sync(); torch.cuda.synchronize()
                t0 = time.perf_counter(); ctx.forward_argmax(lp, 0); pts.append(time.perf_counter() - t0)
            pdt = float(np.median(pts))
            L_, dim_, hid_ = cfg.n_layers, cfg.dim, cfg.hidden_dim
            macs = (len(lp) - 1) * ((L_ - 1) * (4 * dim_ * dim_ + 3 * dim_ * hid_) + 3 * dim_ * dim_) + L_ * (4 * dim_ * dim_ + 3 * dim_ * hid_) + V * dim_
            prefill = {"prompt_tokens": int(len(lp)), "ms": round(pdt * 1e3, 3), "prompt_tokens_per_s": round(len(lp) / pdt, 1), "linear_TMACs_per_s": round(macs / pdt / 1e12, 1),
                       "i8_mfma_peak_TMACs_per_s": 1972.0, "frac_of_i8_mfma_peak": round(macs / pdt / 1e12 / 1972.0, 4),
                       "valu_bound_TMACs_per_s": 840.0, "frac_of_valu_bound": round(macs / pdt / 1e12 / 840.0, 4),
                       "bound_note": "the reference's per-group fp32 chain (cvt + mul + fma per result and group: 48 VALU instructions = 192 cycles per 32 x 32 x 64 fragment against 128 "
                                     "MFMA cycles) bounds the int8 GEMM tiles at ~0.42 of the matrix-core peak (~840 TMAC/s): DESIGN.md section 8b",
                       "parity": (None if pgold is None else {"against": f"the reference's next token behind this prompt ({golden_ids.last})", "match": int(ptok) == pgold[0]}),
                       "note": "wall time of flm_forward_argmax (one host call, prompt ids in, next id out); MACs of the linear layers as the kernels run them (the batch skips "
                               "the last layer's attention and FFN); peak = 3944 TOPS int8 MFMA (MI355X_MICROARCH.md) / 2; not part of `value`"}
        except Exception as e:  # noqa: BLE001
            prefill = {"error": str(e)}
    # per-kernel times, live, HIP events on the ctx stream (eager launches, same kernels as the graph)
    kt = ctx.kernel_times(mid_pos, iters=3)
    # the dominant launch of the token: the fused FFN13 + FFN2 kernel where the token path runs it (single GPU), else FFN13
    dom = next((k for k in ("token", "layers", "layer", "back", "ffn") if kt.get(k, (0.0, 0))[1] > 0), "ffn13")
    dom_us, dom_cnt = kt[dom]
    dom_bytes = ctx.kernel_bytes(dom, mid_pos)
    achieved = dom_bytes / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
    try:
        tpath = ctx.query("token_path")
        token_path = {"resident": bool(ctx.query("resident")), "fell_back_after_timeout": bool(ctx.query("fallback")),
                      "attn_wo_fused": bool(tpath & 1), "ffn_fused": bool(tpath & 2), "qkv_joins_at_long_contexts": bool(tpath & 4),
                      "heads_split_at_long_contexts": bool(tpath & 64), "attention_to_ffn2_in_one_launch": bool(tpath & 128), "whole_layer_in_one_launch": bool(tpath & 256), "all_layers_in_one_launch": bool(tpath & 512) and kt.get("layers", (0.0, 0))[1] > 0,
                      "whole_greedy_token_in_one_launch": bool(tpath & 1024) and kt.get("token", (0.0, 0))[1] > 0,
                      "launches_per_token_short_context": (1 if (tpath & 1024 and kt.get("token", (0.0, 0))[1] > 0) else (1 if tpath & 512 and kt.get("layers", (0.0, 0))[1] > 0 else cfg.n_layers * (1 if tpath & 256 else 2 if tpath & 128 else 3)) + 3)}
    except Exception as e:  # noqa: BLE001
        token_path = {"error": str(e)}
    try:
        ao_active = ctx.query("ao_active")
        token_path["wo_and_ffn2_consumed_in_arrival_order"] = ao_active == 3
    except Exception:  # noqa: BLE001
        ao_active = 0
    ctx_tp_info = getattr(ctx, "tp_info", None) or {}
    kernels = {k: {"us": round(v[0], 2), "per_token": v[1], "GBps": round(ctx.kernel_bytes(k, mid_pos) / (v[0] * 1e-6) / 1e9, 1) if v[0] > 0 else 0.0}
               for k, v in kt.items() if v[1] > 0}
    ctx.close()

    # BASELINE config 5 as stated (LLaMA2-7B int16 + a 512-token prompt), outside the timed region of `value`: a second context with the int16 checkpoint, the prompt
    # through the batched path, the next token checked against the reference's (tests/golden/model_7B_int16_L32.npz)
    config5 = None
    if mode == "single" and args.pos is None and rank == 0 and args.shape == "7B" and qt == ff.QT_INT8 and not args.no_config5:
        try:
            cfg16 = synth.make_config("7B", ff.QT_INT16)
            c16 = capi.Ctx(capi.desc_from_config(cfg16), device=device)
            upload_synthetic(c16, cfg16)
            lp = np.array([1] + [int(x) for x in (np.arange(1, 512) * 7919) % V], dtype=np.int32)
            c16.reset_kv(); t16 = c16.forward_argmax(lp, 0)
            pts = []
            for _ in range(3):
                c16.reset_kv(); c16.sync(); torch.cuda.synchronize()
                t0 = time.perf_counter(); c16.forward_argmax(lp, 0); pts.append(time.perf_counter() - t0)
            g16 = golden_ids(cfg16, ff.QT_INT16, 512)
            config5 = {"workload": "LLaMA2-7B int16, 512-token prompt through the batched path (int16 as hi / lo byte planes on the int8 matrix cores; QK^T and softmax x V on fp32 MFMA)",
                       "ms": round(float(np.median(pts)) * 1e3, 3), "prompt_tokens_per_s": round(512 / float(np.median(pts)), 1),
                       "parity": (None if g16 is None else {"against": f"the reference's next token behind this prompt ({golden_ids.last})", "match": int(t16) == g16[0]})}
            c16.close()
        except Exception as e:  # noqa: BLE001
            config5 = {"error": str(e)}

    # ---- N > 1: the replicas figure beside the tensor-parallel headline (one independent sequence per GPU, no collective) ----
    replicas = None
    if mode == "tp":
        try:
            rctx = capi.Ctx(capi.desc_from_config(cfg), device=device)
            upload_synthetic(rctx, cfg)
            rm = time_decode(rctx, cfg, args, prompt_for(rank), barrier, None)
            r_tok_s, r_elapsed = job_throughput(rm["wall_s"], args.steps, world, "replicas")
            replicas = {"value": round(r_tok_s, 2), "unit": "tokens/s", "scaling": "weak", "ms_per_step": round(1000.0 * r_elapsed / args.steps, 4),
                        "note": f"{world} independent sequences, one whole model per GPU, no data-path collective; whole-job tokens/s = N*K / max over ranks"}
            rctx.close()
        except Exception as e:  # noqa: BLE001
            replicas = {"value": None, "note": f"failed: {e}"}

    qn = 2 if qt == ff.QT_INT8 else 1
    split_now = bool(token_path.get("heads_split_at_long_contexts")) and mid_pos + 1 >= 128      # (the launch's SPLIT instantiation: a head spread over hs / 32 workgroups)
    # (round 5: k_layers<QT, XR2, SPLIT, R5>, R5 = 3 where the launch consumes Wo's / FFN2's activation in arrival order -- the instantiation this run launched, not just any in the library)
    r5_now = 3 if (dom in ("layers", "token") and ao_active > 0) else 0
    dom_regex = {"token": rf"k_layers<{qn}, \d+, {'true' if split_now else 'false'}, {r5_now}, true(?:, \w+)*>", "layers": rf"k_layers<{qn}, \d+, {'true' if split_now else 'false'}, {r5_now}, false(?:, \w+)*>", "layer": rf"k_attn_ffn<{qn}, \d+, true, false>", "back": rf"k_attn_ffn<{qn}, \d+, false, false>", "ffn": rf"k_ffn<{qn},"}.get(dom, rf"k_gemv<{qn}, 2, 2,")
    if args.shape == "7B":
        traffic, traffic_src, traffic_note = pmc_traffic(dom_regex, capi.LIB_PATH)
    else:   # (the committed PMC summaries were collected on the 7B-shaped model: a launch of the same kernel on another shape moves other bytes)
        traffic, traffic_src, traffic_note = None, None, "PMC summaries under profiles/ are of the LLaMA2-7B shape"

    dom_name = {"token": f"k_layers<{args.quant}, TAIL> (the WHOLE greedy token in one launch: the embedding row read by the first layer, ALL {cfg.n_layers} decoder layers -- per layer QKV + RoPE, attention, "
                         f"Wo + residual, FFN13 + SwiGLU, FFN2 + residual --, final norm + classifier, argmax + state advance; every edge a flag round)",
                "layers": f"k_layers<{args.quant}> (ALL {cfg.n_layers} decoder layers of the token in one launch: per layer QKV + RoPE, attention, Wo + residual, FFN13 + SwiGLU, FFN2 + residual; "
                          f"the edges between phases and between layers are flag rounds)",
                "layer": f"k_attn_ffn<{args.quant}, QKV> (the whole decoder layer in one launch: QKV + RoPE, attention, Wo + residual, FFN13 + SwiGLU, FFN2 + residual)",
                "back": f"k_attn_ffn<{args.quant}> (attention, Wo + residual, FFN13 + SwiGLU, FFN2 + residual in one launch)",
                "ffn": f"k_ffn<{args.quant}> (ffn13 + SwiGLU and ffn2 + residual in one launch)"}.get(dom, f"k_gemv<{args.quant},rmsnorm+quantize,swiglu> (ffn13)")
    if rank == 0:
        tb = token_bytes(cfg, mid_pos, esz)
        line = {
            "metric": "decode tokens/s LLaMA2-7B int8" if args.shape == "7B" and qt == ff.QT_INT8 else f"decode tokens/s {args.shape} {args.quant}",
            "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 4), "higher_is_better": True,
            # (N = 1 carries the label of the series it starts: --parallel tp shards ONE sequence = total work fixed as N grows; replicas = per-GPU work fixed)
            "scaling": "strong" if (mode == "tp" or (mode == "single" and args.parallel == "tp")) else "weak", "vs_baseline": None, "dtype": "int8" if qt == ff.QT_INT8 else "int16", "data": "synthetic",
            "config": {"workload": f"LLaMA2-{args.shape} {args.quant} .flm-layout synthetic weights (portable splitmix64 checkpoint, norm weights 1.0), single-stream greedy decode, "
                                   f"prompt {args.prompt_len} tokens, positions {pos}..{pos + args.steps - 1}, fp32 KV cache, max_seq 1024",
                       "parallelism": {"single": "single-gpu", "tp": f"tp{world}: ONE sequence, every matmul split by output rows over {world} GPUs, activation slices exchanged by " + (getattr(ctx, "exchange", "") if mode == "tp" else ""),
                                       "replicas": f"{world} replicas: one independent sequence per GPU, no data-path collective"}[mode],
                       "device_ms_per_step": round(m["ms_dev"] / args.steps, 4), **({"options": args.opt} if args.opt else {})},
            "p50_ms_per_step": round(m["p50_ms"], 4), "p90_ms_per_step": round(m["p90_ms"], 4),
            "parity": m["parity"],
            # per GPU: bytes each GPU streams per token it works on, at the rate it produces them
            "token_roofline": {"bytes_per_token": int(tb / tp), "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU", "pos": mid_pos,
                               "achieved": round(tb / tp * (tok_s / (world / tp)) / 1e9, 1),
                               "frac": round(tb / tp * (tok_s / (world / tp)) / 1e9 / HBM_PEAK_GBS, 4)},
            "roofline": {"kernel": dom_name,
                         "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note,
                         "bytes_per_launch": int(dom_bytes), "avg_launch_us": round(dom_us, 2), "launches_per_token": dom_cnt,
                         **({"layers_per_launch": cfg.n_layers, "us_per_layer": round(dom_us / cfg.n_layers, 2)} if dom == "layers" else {}),
                         **({"layers_per_launch": cfg.n_layers, "layers_alone_us": round(kt.get("layers", (0.0, 0))[0], 2), "us_per_layer": round(kt.get("layers", (0.0, 0))[0] / cfg.n_layers, 2),
                             "note": "the whole greedy token is this one launch: bytes = the layers' + the classifier's + the embedding row; `layers_alone_us` = the same layers as a launch of their own (k_layers<.., TAIL = false>)"} if dom == "token" else {})},
            "token_path": token_path,
            "kernels": kernels,
            "kernels_note": "us per launch, back-to-back launches of one class between one pair of HIP events; on a single GPU a greedy token runs `token` (k_layers<.., TAIL>: embedding row, all L decoder "
                            "layers, classifier, argmax in ONE launch) where it is listed; else `layers` (k_layers: all L decoder layers in one launch): per token = embed + layers + cls + argmax; else `layer` (k_attn_ffn: one launch per layer; timed beside it); else attn_wo (k_attn_o) instead of attn + attn_o and ffn "
                            "(k_ffn) instead of ffn13 + ffn2, and where qkv_attn_wo (k_qkv_attn_o: contexts from 128 positions on) is listed, that instead of qkv + attn_wo; the "
                            "per-phase classes (qkv .. ffn2) are timed beside them for reference",
        }
        if decode_128 is not None:
            line["decode_128"] = decode_128
        if long_ctx is not None:
            line["long_context"] = long_ctx
        if prefill is not None:
            line["prefill"] = prefill
        if config5 is not None:
            line["config5_prefill512_int16"] = config5
        if replicas is not None:
            line["replicas"] = replicas
        if mode == "tp" and ctx_tp_info:
            # (c) of the review's multi-GPU item: who ran, over what, and what an exchange costs (HIP events around the exchange launches / collectives of a timed token)
            info = dict(ctx_tp_info)
            ar = kt.get("allreduce", (0.0, 0))
            info["exchange_us"] = round(ar[0], 2); info["exchanges_per_token"] = ar[1]
            info["scaling_measured"] = info["distinct_devices"] == world
            if tp_structures is not None:
                best = pick_tp_structure(tp_structures)
                info["structures"] = tp_structures
                info["structure"] = best["name"] if best else None
                info["structure_note"] = "`value` = the fastest launch structure whose ids are the reference's on every rank; structures that failed are listed, not fatal"
            line["tp"] = info
        if tp_note:
            line["tp_note"] = tp_note
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def open_tp_ctx(capi, cfg, rank, world, device, dist, torch):
    """one tensor-parallel context per rank.  The ranks are connected peer to peer (every rank's exchange buffer mapped into
    every other rank: flm_p2p_export -> all_gather of the 128-byte blobs -> flm_p2p_import); an RCCL communicator is created as
    well when the backend is nccl, as the fallback exchange should the peer mapping fail."""
    on_gpu = dist.get_backend() == "nccl"
    dev = "cuda" if on_gpu else "cpu"
    comm_id = None
    if on_gpu:
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        comm_id = bytes(idt.cpu().numpy().tobytes())
    ctx = capi.Ctx(capi.desc_from_config(cfg), device=device, rank=rank, world=world, comm_id=comm_id)
    tp_connect(capi, ctx, rank, world, device, dist, torch, comm_id, first=True)
    return ctx


def tp_connect(capi, ctx, rank, world, device, dist, torch, comm_id, first=False):
    """(re)agree on the group's launch structure: flm_p2p_export -> all_gather of the blobs -> flm_p2p_import on every rank (the peers' buffers are mapped the first
    time; options that shape the structure -- tp_trust_fused, tp_fuse_attn, tp_fuse_ffn, fold_xchg -- take effect at this round).  Fills ctx.exchange / ctx.tp_info."""
    on_gpu = dist.get_backend() == "nccl"
    dev = "cuda" if on_gpu else "cpu"
    rehearsal = os.environ.get("FLM_BENCH_FORCE_DEVICE") is not None and world in (2, 4, 8)
    if rehearsal and first:
        # the rehearsal on ONE GPU: give every rank a CU partition of its own, so that the latency path (flag rounds folded into the consuming launches,
        # attention + Wo as one launch across the ranks) runs between PROCESSES here as it does between GPUs there.  (Set BEFORE the blobs are exchanged:
        # the group agrees on its launch structure at flm_p2p_import.)
        ctx.set_option("cu_parts", world)
    mine = torch.frombuffer(bytearray(ctx.p2p_export()), dtype=torch.uint8).to(dev)
    allb = [torch.zeros(128, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(allb, mine)
    ok = 1
    try:
        if os.environ.get("FLM_BENCH_NO_P2P"):
            raise RuntimeError("FLM_BENCH_NO_P2P: peer mapping skipped (the RCCL all-gather branch with a world-size communicator)")
        ctx.p2p_import([bytes(b.cpu().numpy().tobytes()) for b in allb])
    except Exception as e:  # noqa: BLE001
        log(f"rank {rank}: peer-to-peer mapping failed ({e}); " + ("falling back to RCCL all-gathers" if comm_id else "no fallback"))
        ok = 0
    okt = torch.tensor([ok], device=dev); dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    if int(okt.item()) == 0:
        # all ranks must agree on the exchange: everybody stays on the RCCL all-gathers
        if not comm_id:
            ctx.close()
            raise RuntimeError("peer-to-peer mapping failed and there is no RCCL communicator")
        if ok:
            ctx.set_option("use_p2p", 0)
        ctx.exchange = "rccl all-gather"
        ctx.tp_info = {"transport": "rccl", "launches_per_sharded_layer": 9}
    else:
        ctx.exchange = "peer-to-peer stores over xGMI + flag round"
        if rehearsal:
            ctx.exchange += f" (rehearsal: {world} ranks on one GPU, 1/{world} of the CUs each)"
        # which launch structure the sharded token runs (read back, not assumed): flag rounds folded into the consuming launches or k_xchg launches; QKV + attention + Wo /
        # FFN13 + FFN2 as launches that span the ranks
        fold, fa, fn = ctx.query("fold_active"), ctx.query("grp_tp_fuse_attn"), ctx.query("grp_tp_fuse_ffn")      # (what the GROUP agreed on at import)
        per_layer = 9 if not fold else (5 - (2 if fa >= 2 else 1 if fa == 1 else 0) - (1 if fn else 0))
        tpl = ctx.query("grp_tp_fuse_layers")
        if tpl:
            per_layer = 0
            gra = 1 if (ctx.query("gr_edges") and ctx.query("grp_gr")) else 0
            ctx.exchange += (f"; ALL layers of the sharded token in one launch per rank that spans the ranks (k_layers<TP>; " +
                             ("every cross-rank vector as data-tagged 8-byte granules: no flag line, no fence)" if gra else f"flag rounds between the ranks' workgroups, tp_fence {ctx.query('tp_fence_active')})"))
        else:
            ctx.exchange += f"; {per_layer} launches per sharded layer (fold_active {fold}, tp_fuse_attn {fa}, tp_fuse_ffn {fn})"
        ctx.tp_info = {"transport": "p2p", "launches_per_sharded_layer": per_layer, "fold_active": int(fold), "tp_fuse_attn": int(fa), "tp_fuse_ffn": int(fn),
                       "tp_fuse_layers": int(tpl), "tp_fence": int(ctx.query("tp_fence_active")) if tpl else None, "granules": (gra if tpl else None)}
    # every rank's device ordinal, as the ranks themselves see it (ranks_seen = how many distinct devices the group really runs on)
    devs = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(devs, torch.tensor([int(device)], dtype=torch.int64, device=dev))
    ctx.tp_info["rank_devices"] = [int(d.item()) for d in devs]
    ctx.tp_info["ranks_seen"] = world
    ctx.tp_info["distinct_devices"] = len(set(ctx.tp_info["rank_devices"]))
    ctx.comm_id = comm_id
    return ctx


# The launch structures a sharded token can take, slowest and most conservative first.  Between DISTINCT devices the folded exchanges / rank-spanning launches rely on
# system-scope store / flag ordering over xGMI that the build box (one GPU) can only rehearse between CU partitions: the library runs them there only when every rank says
# "tp_trust_fused".  bench.py trusts nothing: it times the conservative structure, then each faster one, verifies EVERY one against the reference's golden ids on every rank
# (all-reduce MIN), and reports the fastest verified one as `value`; a structure that mismatches or gives up is reported under tp.structures, not fatal.
TP_STRUCTURES = (("xchg-launches", {"tp_trust_fused": 0, "tp_fuse_ffn": 0, "tp_fuse_layers": 0, "tp_fence": -1, "gr_edges": 0}),
                 ("folded, QKV + attention + Wo across ranks", {"tp_trust_fused": 1, "tp_fuse_ffn": 0, "tp_fuse_layers": 0, "tp_fence": -1, "gr_edges": 0}),
                 ("all layers in one rank-spanning launch, fenced flags", {"tp_trust_fused": 1, "tp_fuse_ffn": 0, "tp_fuse_layers": 1, "tp_fence": 3, "gr_edges": 0}),
                 ("all layers in one rank-spanning launch, flags", {"tp_trust_fused": 1, "tp_fuse_ffn": 0, "tp_fuse_layers": 1, "tp_fence": 0, "gr_edges": 0}),
                 ("all layers in one rank-spanning launch, data-tagged granules (no flag, no fence)", {"tp_trust_fused": 1, "tp_fuse_ffn": 0, "tp_fuse_layers": 1, "tp_fence": -1, "gr_edges": 1}),
                 ("folded, + FFN13 + FFN2 across ranks", {"tp_trust_fused": 1, "tp_fuse_ffn": 1, "tp_fuse_layers": 0, "tp_fence": -1, "gr_edges": 0}))      # (last: slower than its predecessor at 2-4 ranks on one GPU)


def pick_tp_structure(results):
    """fastest verified structure of [{'name', 'verified', 'ms_per_step', ...}]; None when none verified.  (CPU-testable: tests/test_bench_gloo.py)"""
    ok = [r for r in results if r.get("verified") and r.get("ms_per_step")]
    return min(ok, key=lambda r: r["ms_per_step"]) if ok else None


def run_tp_structures(capi, ctx, cfg, args, prompt, barrier, gold, rank, world, device, dist, torch):
    """time every launch structure of the sharded token; returns (best measurement or None, list of per-structure records).  Leaves ctx on the best structure."""
    devt = "cuda" if dist.get_backend() == "nccl" else "cpu"
    results, measured, seen = [], {}, {}
    forced = os.environ.get("FLM_TP_TRUST_FUSED")
    todo = TP_STRUCTURES if forced is None else tuple(x for x in TP_STRUCTURES if x[1]["tp_trust_fused"] == int(forced))[:1] or TP_STRUCTURES[:1]
    base_ids = None      # (no golden ids for the shape: the first -- conservative -- structure's ids)
    for name, opts in todo:
        rec = {"name": name, "options": dict(opts)}
        ok, m = 1, None
        try:
            for k, v in opts.items():
                ctx.set_option(k, v)
            tp_connect(capi, ctx, rank, world, device, dist, torch, getattr(ctx, "comm_id", None))
            info = dict(ctx.tp_info)
            sig = (info.get("transport"), info.get("launches_per_sharded_layer"), info.get("fold_active"), info.get("tp_fuse_attn"), info.get("tp_fuse_ffn"), info.get("tp_fuse_layers"), info.get("tp_fence"), info.get("granules"))
            rec.update(transport=info.get("transport"), launches_per_sharded_layer=info.get("launches_per_sharded_layer"),
                       ran=(f"ONE launch for all layers of the sharded token, spanning the ranks (k_layers<TP>, {'data-tagged granules' if info.get('granules') else 'flag rounds, tp_fence ' + str(info.get('tp_fence'))}) + embedding row, classifier, logits exchange, argmax" if info.get("tp_fuse_layers") else
                            f"{info.get('launches_per_sharded_layer')} launches per sharded layer (fold_active {info.get('fold_active')}, tp_fuse_attn {info.get('tp_fuse_attn')}, tp_fuse_ffn {info.get('tp_fuse_ffn')})")
                           + ("; every rank on ONE device: the group folds its exchanges without being asked to trust anything" if len(set(info.get("rank_devices", [0]))) == 1 and world > 1 else ""))
        except Exception as e:  # noqa: BLE001
            ok, sig = 0, None
            rec["error"] = f"rank {rank}: {e}"
        okt = torch.tensor([ok], device=devt); dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 0:
            rec.update(verified=False); rec.setdefault("error", "another rank could not set this structure up")
            results.append(rec); continue
        if sig in seen:      # (e.g. every rank on ONE device: the group folds without being asked to trust anything -- the same launches as a structure already timed)
            rec.update(verified=seen[sig].get("verified"), same_launches_as=seen[sig]["name"], ms_per_step=seen[sig].get("ms_per_step"))
            results.append(rec); continue
        try:
            m = time_decode(ctx, cfg, args, prompt, barrier, gold)
            if gold is not None and m["parity"]["match"] is not True:
                ok = 0; rec["error"] = f"rank {rank}: ids differ from the reference's at generated token {m['parity'].get('first_mismatch')}"
            elif gold is None and base_ids is not None and list(m["ids"]) != list(base_ids):      # no golden ids for this shape: the conservative structure's ids are what the others must reproduce
                ok = 0; rec["error"] = f"rank {rank}: ids differ from the conservative structure's"
            elif ctx.query("fallback"):
                ok = 0; rec["error"] = f"rank {rank}: a cross-workgroup wait timed out (the call was re-run on one kernel per phase)"
        except Exception as e:  # noqa: BLE001
            ok = 0; rec["error"] = f"rank {rank}: {e}"
        okt = torch.tensor([ok], device=devt); dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        verified = int(okt.item()) == 1
        wall = torch.tensor([m["wall_s"] if m is not None else 0.0], dtype=torch.float64, device=devt); dist.all_reduce(wall, op=dist.ReduceOp.MAX)   # (every rank, whatever happened on it)
        if float(wall.item()) > 0.0:
            rec.update(ms_per_step=round(1000.0 * float(wall.item()) / args.steps, 4))
        if m is not None:
            rec.update(p50_ms_per_step=round(m["p50_ms"], 4))
        if verified:     # what an exchange costs under this structure: HIP events around the exchange launches / collectives of whole timed tokens (every rank, in step)
            try:
                ar = ctx.kernel_times(m["pos"] + args.steps // 2, iters=2).get("allreduce", (0.0, 0))
                rec.update(exchange_us=round(ar[0], 2), exchange_launches_per_token=ar[1])
            except Exception as e:  # noqa: BLE001
                rec["exchange_us_error"] = str(e)
        if gold is None and base_ids is None:      # nothing was checked: the conservative structure is the baseline the others are compared with -- say so instead of "verified"
            rec["verified"] = "baseline-unverified" if verified else False
            rec["verified_against"] = "nothing: no golden ids for this shape; this structure's ids are the baseline of the others"
            if verified and m is not None: base_ids = list(m["ids"])
        else:
            rec["verified"] = verified
            rec["verified_against"] = ("the reference's golden ids on every rank (all-reduce MIN)" if gold is not None else "the conservative structure's ids on every rank (all-reduce MIN); no golden ids for this shape")
        if not verified:
            rec.setdefault("error", "another rank's ids differed, or it gave up")
        results.append(rec); seen[sig] = rec
        if verified:
            measured[name] = m
        else:
            break        # a structure that failed may have left flags / fallback state behind: do not build faster ones on it
    best = pick_tp_structure(results)
    if best is None:
        return None, results
    if results[-1]["name"] != best["name"] or not results[-1].get("verified"):
        for k, v in dict(next(o for n, o in TP_STRUCTURES if n == best["name"])).items():
            ctx.set_option(k, v)
        tp_connect(capi, ctx, rank, world, device, dist, torch, getattr(ctx, "comm_id", None))
    name = best.get("same_launches_as") or best["name"]
    return measured.get(name), results


if __name__ == "__main__":
    main()
