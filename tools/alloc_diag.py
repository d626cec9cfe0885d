"""where does the device's free memory go during a context's first calls?  python tools/alloc_diag.py   (run on the GPU box; tests/test_gpu_configs.py::test_nothing_is_allocated...)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as graft
graft.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
hip = ctypes.CDLL("libamdhip64.so")
def free_bytes():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value
for shape, qt, layers, nprompt in (("7B", ff.QT_INT8, 2, 9), ("small", ff.QT_INT16, None, 140), ("tiny", ff.QT_INT8, None, 3)):
    cfg = synth.make_config(shape, qt)
    if layers: cfg.n_layers = layers
    tensors = synth.make_tensors(cfg, seed=3)
    ctx = capi.Ctx(capi.desc_from_config(cfg)); ctx.upload_all(tensors)
    prompt = np.array([1] + [int(x) for x in (np.arange(1, nprompt) * 7919) % cfg.vocab_size], np.int32)
    f = [free_bytes()]
    def mark(name):
        f.append(free_bytes()); print(f"{shape:6s} {name:28s} free delta {f[-2] - f[-1]:10d}", flush=True)
    ctx.sync(); mark("sync")
    lg = ctx.forward(prompt, 0); mark("forward(prompt)")
    first = ctx.forward_argmax(prompt, 0); mark("forward_argmax(prompt)")
    ids = ctx.decode_greedy(first, len(prompt), 1); mark("decode_greedy 1")
    ids = ctx.decode_greedy(first, len(prompt), 2); mark("decode_greedy 2")
    ids = ctx.decode_greedy(first, len(prompt), 40); mark("decode_greedy 40")
    one = ctx.forward(np.array([int(ids[-1])], np.int32), len(prompt) + 40); mark("forward(1 token)")
    ctx.close()
