import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi, synth, flmfile as ff
import oracle_py as O
def rel(a,b): return float(np.max(np.abs(a-b))/max(np.max(np.abs(b)),1e-30))
def nbad(a,b): return int(np.sum(a.view(np.uint32)!=b.view(np.uint32)))
O.orc().orc_model_kcache.restype = C.POINTER(C.c_float); O.orc().orc_model_vcache.restype = C.POINTER(C.c_float); O.orc().orc_model_tap_x.restype = C.POINTER(C.c_float)
# rmsnorm 11008
rng=np.random.default_rng(11008); x=rng.standard_normal(11008).astype(np.float32)*3; w=rng.uniform(0.5,1.5,11008).astype(np.float32)
o=capi.op_rmsnorm(x,w); r=O.rmsnorm(x,w); print("rmsnorm11008 nbad",nbad(o,r),"rel",rel(o,r), "ratio", (o/r)[:3])
for n in (5120, 8192, 12288):
    x=rng.standard_normal(n).astype(np.float32)*3; w=rng.uniform(0.5,1.5,n).astype(np.float32)
    o=capi.op_rmsnorm(x,w); r=O.rmsnorm(x,w); print("rmsnorm",n,"nbad",nbad(o,r),"rel",rel(o,r))
# attention hs=96
hs,heads,ms=96,2,1024
kc_o=np.zeros((heads,ms,hs),np.float32); vc_o=np.zeros_like(kc_o); kc_g=np.zeros_like(kc_o); vc_g=np.zeros_like(kc_o)
for pos in range(3):
    q=rng.standard_normal((heads,hs)).astype(np.float32); k=rng.standard_normal((heads,hs)).astype(np.float32); v=rng.standard_normal((heads,hs)).astype(np.float32)
    ref=np.stack([O.attention_head(kc_o[h],vc_o[h],q[h:h+1],k[h:h+1],v[h:h+1],pos)[0] for h in range(heads)])
    out=capi.op_attention(kc_g,vc_g,q.reshape(-1),k.reshape(-1),v.reshape(-1),heads,hs,ms,pos).reshape(heads,hs)
    print("attn96 pos",pos,"nbad",nbad(out,ref),"rel",rel(out,ref), "kc bad", nbad(kc_g,kc_o))
for shape in ("tiny","small"):
    cfg = synth.make_config(shape, ff.QT_INT8)
    t = synth.make_tensors(cfg, seed=1234)
    ctx = capi.Ctx(capi.desc_from_config(cfg)); ctx.upload_all(t)
    om = O.OracleModel(cfg, t)
    for n in (1,2,8):
        ctx.reset_kv(); om.reset()
        p = np.array([1,5,9,200,33,7,8,100][:n], np.int32)
        lg=ctx.forward(p,0); lo=om.forward(p,0)
        print(shape,"n",n,"nbad",nbad(lg,lo),"rel",rel(lg,lo))
        H, hs, ms = cfg.n_heads, cfg.head_size, 1024
        for l in range(cfg.n_layers):
            go = ctx.debug_read("kcache", l, H*ms*hs).reshape(H, ms, hs); oo = np.ctypeslib.as_array(O.orc().orc_model_kcache(om.h, l), shape=(H*ms*hs,)).reshape(H, ms, hs)
            gv = ctx.debug_read("vcache", l, H*ms*hs).reshape(H, ms, hs); ov = np.ctypeslib.as_array(O.orc().orc_model_vcache(om.h, l), shape=(H*ms*hs,)).reshape(H, ms, hs)
            print("   layer",l,"K nbad per pos",[nbad(go[:,p_],oo[:,p_]) for p_ in range(n)],"V",[nbad(gv[:,p_],ov[:,p_]) for p_ in range(n)])
    ctx.close()
