"""replay of tests/test_gpu_model.py::test_7b_width_layer_every_code_path_vs_oracle in a loop: a 5-token prompt (batched prefill of 4 +
one decode token) and three single-token forwards, logits compared bit for bit with the first pass.  python tools/stress2.py [reps]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for qt, name in ((ff.QT_INT16, "int16"), (ff.QT_INT8, "int8")):
    cfg = synth.make_config("7B", qt); cfg.n_layers = 1
    tensors = synth.make_tensors(cfg, seed=31)
    prompt = np.array([1] + [int(x) for x in (np.arange(1, 5) * 7919) % cfg.vocab_size], dtype=np.int32)
    ref = None
    for opts in ({"fuse_attn_o": 0}, {}, {"use_prefill": 0}):
        ctx = capi.Ctx(capi.desc_from_config(cfg)); ctx.upload_all(tensors)
        for k, v in opts.items(): ctx.set_option(k, v)
        nbad = 0
        for r in range(reps):
            ctx.reset_kv()
            out = [ctx.forward(prompt, 0).copy()]
            cur, pos = int(np.argmax(out[0])), len(prompt)
            for i in range(3):
                out.append(ctx.forward(np.array([cur], np.int32), pos).copy()); cur = int(np.argmax(out[-1])); pos += 1
            if ref is None: ref = out
            for i, (a, b) in enumerate(zip(out, ref)):
                if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                    nbad += 1; print(f"{name} {opts} rep {r}: forward #{i} differs in {(a.view(np.uint32) != b.view(np.uint32)).sum()} logits", flush=True); break
        print(f"{name} {opts}: {nbad} bad of {reps}", flush=True)
        bad += nbad
        ctx.close()
print("stress2:", "FAILED" if bad else "ok")
