#!/usr/bin/env python3
"""Round-4 golden vectors, generated FROM THE REFERENCE ITSELF in the build container (needs /root/reference and oracle/_ref/libflref.so: `make -C oracle ref`).
Fixtures hold inputs and expected outputs only (ids, sha256 digests of logits, a few logits, a checksum of the weights).

  model_7B_int16_L32.npz   BASELINE config 5 at FULL depth: the 32-layer LLaMA2-7B-shaped int16 model (portable splitmix64 checkpoint of fast_llama_amd/synth.py)
                           through the reference's ParallelTransformer::forward with max_batch_size = 512 (transformer.h:78-79, transformer.cpp:92-94):
                           (a) the 512-token prompt of bench.py's prefill / long-context modes as ONE batched forward, then 4 greedy steps -- sha256 of every step's logits;
                           (b) bench.py's 9-token prompt and 26 greedy steps -- ids (bench.py --quant int16).
  model_7B_int8_L32_p512.npz  the int8 model of model_7B_int8_L32.npz with the 512-token prompt: the batched forward's logits digest, then 26 greedy steps
                           (ids + digests): bench.py's `long_context` / `--pos 512` / `--config prefill512-int8` lines.
  model_1p3B_int8.npz      bench.py --shape 1.3B (4 layers, vocabulary 55296): the 9-token prompt and 26 greedy steps (ids + digests).
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
import oracle_py as O  # noqa: E402
from fast_llama_amd import flmfile as ff, synth  # noqa: E402
from make_golden_r2 import bench_prompt, weights_checksum  # noqa: E402


def long_prompt(V, n=512):
    return np.array([1] + [int(x) for x in (np.arange(1, n) * 7919) % V], dtype=np.int32)


def sha(l):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(l).tobytes()).digest(), dtype=np.uint8)


def run(m, prompt, ndec, t0, tag):
    ids, digests, head, margin = [], [], [], []
    pos, cur = 0, prompt
    for step in range(ndec + 1):
        l = m.forward(cur, pos)
        srt = np.sort(l)
        ids.append(int(np.argmax(l))); digests.append(sha(l)); head.append(l[:16].copy()); margin.append(float((srt[-1] - srt[-2]) / abs(srt[-1])))
        pos += len(cur); cur = np.array([ids[-1]], np.int32)
        print(f"  {tag} step {step}: id {ids[-1]} margin {margin[-1]:.3e} ({time.time() - t0:.0f}s)", flush=True)
    return dict(ids=np.array(ids, np.int32), sha256=np.stack(digests), head=np.stack(head), margin=np.array(margin, np.float32))


def write_model(shape, qt, path):
    cfg = synth.make_config(shape, qt)
    tensors = synth.make_tensors_portable(cfg)
    chk = weights_checksum(sorted(tensors.items()))
    ff.write_flm(path, cfg, synth.make_tokenizer(cfg.vocab_size), tensors)
    del tensors
    return cfg, chk


def g_int16(threads):
    t0 = time.time(); path = "/tmp/golden-7B-int16-L32.flm"
    cfg, chk = write_model("7B", ff.QT_INT16, path)
    print(f"7B int16 checkpoint written in {time.time() - t0:.0f}s, checksum {chk}", flush=True)
    m = O.RefModel(path, O.QT_INT16, threads=threads, max_batch=512)
    a = run(m, long_prompt(cfg.vocab_size), 4, t0, "int16 p512")
    del m
    m = O.RefModel(path, O.QT_INT16, threads=threads, max_batch=64)
    b = run(m, bench_prompt(cfg.vocab_size), 26, t0, "int16 p9")
    del m
    np.savez_compressed(os.path.join(HERE, "model_7B_int16_L32.npz"), weights_checksum=np.uint64(chk), p512_prompt=long_prompt(cfg.vocab_size), p9_prompt=bench_prompt(cfg.vocab_size),
                        **{"p512_" + k: v for k, v in a.items()}, **{"p9_" + k: v for k, v in b.items()})
    os.remove(path)


def g_int8_p512(threads):
    t0 = time.time(); path = "/tmp/golden-7B-int8-L32.flm"
    cfg, chk = write_model("7B", ff.QT_INT8, path)
    print(f"7B int8 checkpoint written in {time.time() - t0:.0f}s, checksum {chk}", flush=True)
    m = O.RefModel(path, O.QT_INT8, threads=threads, max_batch=512)
    a = run(m, long_prompt(cfg.vocab_size), 26, t0, "int8 p512")
    del m
    np.savez_compressed(os.path.join(HERE, "model_7B_int8_L32_p512.npz"), weights_checksum=np.uint64(chk), prompt=long_prompt(cfg.vocab_size), **a)
    os.remove(path)


def g_1p3b(threads):
    t0 = time.time(); path = "/tmp/golden-1p3B-int8.flm"
    cfg, chk = write_model("1.3B", ff.QT_INT8, path)
    m = O.RefModel(path, O.QT_INT8, threads=threads, max_batch=64)
    a = run(m, bench_prompt(cfg.vocab_size), 26, t0, "1.3B p9")
    del m
    np.savez_compressed(os.path.join(HERE, "model_1p3B_int8.npz"), weights_checksum=np.uint64(chk), prompt=bench_prompt(cfg.vocab_size), **a)
    os.remove(path)


if __name__ == "__main__":
    assert O.have_ref(), "build oracle/_ref first: make -C oracle ref"
    what = sys.argv[1:] or ["1p3b", "int8", "int16"]
    th = int(os.environ.get("GOLDEN_THREADS", "8"))
    if "1p3b" in what:
        g_1p3b(th)
    if "int8" in what:
        g_int8_p512(th)
    if "int16" in what:
        g_int16(th)
