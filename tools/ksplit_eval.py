#!/usr/bin/env python3
"""What north_star's tensor-parallel partitioning (Megatron style: Wo and W2 split along K at 64-group boundaries, partials summed in
rank order 0..N-1 -- an all-reduce) does to the golden model's logits and greedy ids, against the reference's own arithmetic.

TEST INFRASTRUCTURE (oracle side; the product never sees this).  CPU only.  The full 32-layer LLaMA2-7B-shaped int8 model bench.py
times (fast_llama_amd/synth.py's portable checkpoint) runs through oracle/liboracle.so -- the pinned restatement of the reference
(`orc_model_forward`, N = 1: its ids must be the reference's, tests/golden/model_7B_int8_L32.npz) -- and through the same code with
`orc_model_set_ksplit(N)`, N = 2, 4, 8 (`orc_matmul_q_ksplit`: every rank runs quant_operators.cpp:252-284's chain over ITS groups,
the partials are added in rank order).  Two runs per N:
  * teacher-forced: the reference's ids are fed whatever the K-split model would pick, so that every step's logits compare like for like:
    max and mean relative logit error (|a - b| / max|b| per step, north_star's "1e-3 relative"), argmax agreement per step;
  * free-running greedy: the first step whose id differs from the reference's.
Writes profiles/r06_ksplit_eval.json.   usage: tools/ksplit_eval.py [steps=24] [layers=32]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
import oracle_py as O  # noqa: E402
from fast_llama_amd import flmfile as ff, synth  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    gold = np.load(os.path.join(ROOT, "tests", "golden", "model_7B_int8_L32.npz"))
    prompt, gids = gold["prompt"], gold["ids"]
    steps = min(steps, len(gids) - 1)
    cfg = synth.make_config("7B", ff.QT_INT8)
    cfg.n_layers = layers
    t0 = time.time()
    m = O.OracleModel(cfg, {})
    for k, v in synth.iter_portable(cfg):      # one tensor at a time: the model keeps its own copy
        m.set_tensor(k, v)
    print(f"model up in {time.time() - t0:.0f}s", flush=True)
    lib = O.orc()

    def run(parts, forced):
        lib.orc_model_set_ksplit(m.h, parts)
        m.reset()
        out, ids = [], []
        pos, cur = 0, prompt
        for s in range(steps + 1):
            l = m.forward(cur, pos)
            out.append(l); ids.append(int(np.argmax(l)))
            pos += len(cur)
            cur = np.array([int(gids[s]) if forced else ids[-1]], np.int32)
        return np.stack(out), ids

    base, bids = run(1, True)
    _, gids1 = run(1, False)            # the reference arithmetic, free running (32 layers: the golden ids themselves)
    full = layers == 32
    res = {"model": f"LLaMA2-7B shape int8, {layers} layers, portable synthetic checkpoint", "steps": steps + 1, "positions": f"prompt of {len(prompt)} then {steps} decode steps",
           "oracle_ids_equal_reference": bool(full and bids == [int(x) for x in gids[:steps + 1]]) if full else None,
           "golden_top2_margin_min": float(gold["margin"][:steps + 1].min()) if full else None, "splits": {}}
    print("N=1 ids == reference:", res["oracle_ids_equal_reference"], f"({time.time() - t0:.0f}s)", flush=True)
    for parts in (2, 4, 8):
        lo, ids_f = run(parts, True)
        den = np.abs(base).max(axis=1)
        rel = np.abs(lo - base).max(axis=1) / den
        mean_rel = (np.abs(lo - base).mean(axis=1) / den)
        agree = [int(a == b) for a, b in zip(ids_f, bids)]
        _, ids_g = run(parts, False)
        first = next((i for i, (a, b) in enumerate(zip(ids_g, gids1)) if a != b), None)
        res["splits"][str(parts)] = {
            "max_rel_logit_err": float(rel.max()), "max_rel_logit_err_first_step": float(rel[0]), "mean_rel_logit_err": float(mean_rel.mean()),
            "bitwise_equal_steps": int(sum(np.array_equal(a, b) for a, b in zip(lo, base))),
            "teacher_forced_argmax_agree": int(sum(agree)), "teacher_forced_first_disagree": (agree.index(0) if 0 in agree else None),
            "greedy_first_id_mismatch": first, "within_1e-3": bool(rel.max() <= 1e-3)}
        print(parts, json.dumps(res["splits"][str(parts)]), f"({time.time() - t0:.0f}s)", flush=True)
    lib.orc_model_set_ksplit(m.h, 1)
    name = "r06_ksplit_eval.json" if full else f"r06_ksplit_eval_L{layers}.json"
    with open(os.path.join(ROOT, "profiles", name), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
