import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from fractions import Fraction
import __graft_entry__ as g; g.load_package()
from fast_llama_amd import capi
import oracle_py as O
def rn32(fr):
    # correctly round a Fraction to float32
    d = float(fr)  # correctly rounded double
    f = np.float32(d)
    # fix double rounding: check neighbours
    cands = [f, np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf))]
    best = min(cands, key=lambda c: (abs(Fraction(float(c)) - fr), int(np.float32(c).view(np.uint32)) & 1))
    return np.float32(best)
def chain_ss(x):
    n=len(x); l=[Fraction(0)]*4; lf=[np.float32(0)]*4
    for i in range(0,n,4):
        for k in range(4):
            v=Fraction(float(x[i+k])); lf[k]=rn32(v*v+Fraction(float(lf[k])))
    r=np.float32(0)
    for k in range(4): r=np.float32(r+lf[k])
    return r
rng=np.random.default_rng(3)
for n in (4096, 8192, 8256, 9216, 11008, 12288):
    x=(rng.standard_normal(n)*3).astype(np.float32); w=np.ones(n,np.float32)
    o=capi.op_rmsnorm(x,w); r=O.rmsnorm(x,w)
    ss_exact=chain_ss(x)
    ss_orc=np.float32(O.orc().orc_square_sum(O._p(x), C.c_size_t(n)))
    rr=np.float32(1.0/np.float64(np.sqrt(np.float32(ss_exact/np.float32(n)+np.float32(1e-5)))))
    exp=(x*w)*rr
    print(n,"gpu==orc",np.array_equal(o.view(np.uint32),r.view(np.uint32)),"orc ss==exact chain",ss_orc==ss_exact, ss_orc, ss_exact, "gpu==exact-model", np.array_equal(o.view(np.uint32),exp.astype(np.float32).view(np.uint32)))
