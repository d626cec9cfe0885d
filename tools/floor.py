import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
cfg = synth.make_config("7B", ff.QT_INT8); cfg.n_layers = 4
ctx = capi.Ctx(capi.desc_from_config(cfg)); ctx.upload_all(synth.make_tensors(cfg, seed=1))
first = ctx.forward_argmax(np.arange(1, 9, dtype=np.int32), 0)
for abl in (0, 16, 2, 34):
    ctx.set_option("ablate", abl)
    for graph in (1, 0):
        ctx.set_option("use_graph", graph)
        ctx.decode_timed(first, 8, 8)
        ms = ctx.decode_timed(first, 16, 64)
        print(f"ablate={abl} graph={graph}: {ms/64*1000:.1f} us/token over {5*4+3} launches -> {ms/64*1000/(5*4+3):.2f} us per launch")
