// flm_ops.hip -- op-level exports (flm_op_*): 1:1 mirrors of the reference operator seam, host pointers in / out, running the same device code as the token
// path.  Used by the parity tests.
#include "flm_host.h"

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, n ? n : 4) == hipSuccess ? 0 : 1; }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};
#define OPC(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_last_error = std::string(#expr " failed: ") + hipGetErrorString(e_); return FLM_ERR_HIP; } } while (0)
}

extern "C" {

// ---------------------------------------------------------------------------------------------
// op-level exports
// ---------------------------------------------------------------------------------------------

int flm_op_quantize(int qt, void* qx, float* qs, const float* x, size_t n, int gs) {
    if (!qx || !qs || !x || gs != kGroup || n % kGroup) return FLM_ERR_INVALID;
    if (qt != FLM_QT_INT8 && qt != FLM_QT_INT16) return FLM_ERR_UNSUPPORTED;
    const int e = esz_of(qt);
    DevBuf dx, dq, ds;
    if (dx.alloc(n * 4) || dq.alloc(n * e) || ds.alloc(n / kGroup * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(dx.p, x, n * 4, hipMemcpyHostToDevice));
    if (n <= 16384) {
        // through the fused path's prologue (PRO_QUANT), tapped
        GemvArgs a{}; a.n = (int)n; a.items = 0; a.x = dx.as<float>(); a.dbg_xq = dq.p; a.dbg_xs = ds.as<float>();
        int r = launch_gemv<PRO_QUANT, EPI_STORE>(nullptr, 0, qt, a, 1); if (r) return r;
    } else {
        int r = quantize_flat(nullptr, 0, qt, dq.p, ds.as<float>(), dx.as<float>(), n); if (r) return r;
    }
    OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(qx, dq.p, n * e, hipMemcpyDeviceToHost));
    OPC(hipMemcpy(qs, ds.p, n / kGroup * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

// square_sum (x86_simd.cpp:942-960) of x[n], n a multiple of 16: out6 = { speculative wave evaluation (sq_chain_spec), sequential total, the 4 strided lanes }
int flm_op_square_sum(const float* x, size_t n, float* out6) {
    if (!x || !out6 || n % 16 || n == 0 || n > 16384) return FLM_ERR_INVALID;
    DevBuf dx, dout;
    if (dx.alloc(n * 4) || dout.alloc(16 * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(dx.p, x, n * 4, hipMemcpyHostToDevice));
    const size_t lds = ((size_t)4 * chain_strip_floats((int)n) + 4 * (n / 4 + 8)) * 4;
    hipLaunchKernelGGL(k_op_square_sum, dim3(1), dim3(256), lds, 0, dout.as<float>(), dx.as<float>(), (int)n);
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(out6, dout.p, 6 * 4, hipMemcpyDeviceToHost));
    if (getenv("FLM_SQ_ITERS")) { float it[6]; hipMemcpy(it, (char*)dout.p + 24, 24, hipMemcpyDeviceToHost); fprintf(stderr, "sq_chain_spec rounds per chain (-1: plain chain): %g %g %g %g; shader-clock ticks: speculative %g, plain %g\n", it[0], it[1], it[2], it[3], it[4], it[5]); }
    return FLM_OK;
}

// the fused launches' publish / poll / coherent-read sequence, `rounds` times on one 256-thread workgroup per CU (k_handoff_litmus, flm_misc.h)
int flm_op_handoff_litmus(int rounds, int* wrong_values, int* timed_out) {
    if (rounds < 1 || !wrong_values || !timed_out) return FLM_ERR_INVALID;
    int dev = 0; OPC(hipGetDevice(&dev));
    hipDeviceProp_t pr; OPC(hipGetDeviceProperties(&pr, dev));
    const int G = pr.multiProcessorCount < 256 ? pr.multiProcessorCount : 256;
    DevBuf pay, fl, er;
    if (pay.alloc((size_t)2 * G * 64 * 4) || fl.alloc((size_t)G * 64) || er.alloc(8)) return FLM_ERR_OOM;
    OPC(hipMemset(pay.p, 0, (size_t)2 * G * 64 * 4)); OPC(hipMemset(fl.p, 0, (size_t)G * 64)); OPC(hipMemset(er.p, 0, 8));
    hipLaunchKernelGGL(k_handoff_litmus, dim3(G), dim3(256), 0, 0, pay.as<float>(), fl.as<unsigned>(), rounds, er.as<int>(), er.as<int>() + 1);
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    int h[2] = {0, 0};
    OPC(hipMemcpy(h, er.p, 8, hipMemcpyDeviceToHost));
    *wrong_values = h[0]; *timed_out = h[1];
    return FLM_OK;
}

int flm_op_rmsnorm(float* o, const float* x, const float* w, size_t n) {
    if (!o || !x || !w || n % kGroup || n > 16384 || n == 0) return FLM_ERR_INVALID;
    DevBuf dx, dw, dn, dq, ds;
    if (dx.alloc(n * 4) || dw.alloc(n * 4) || dn.alloc(n * 4) || dq.alloc(n) || ds.alloc(n / kGroup * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(dx.p, x, n * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(dw.p, w, n * 4, hipMemcpyHostToDevice));
    GemvArgs a{}; a.n = (int)n; a.items = 0; a.x = dx.as<float>(); a.norm_w = dw.as<float>();
    a.dbg_xn = dn.as<float>(); a.dbg_xq = dq.p; a.dbg_xs = ds.as<float>();
    int r = launch_gemv<PRO_RMSNORM_QUANT, EPI_STORE>(nullptr, 0, FLM_QT_INT8, a, 1); if (r) return r;
    OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(o, dn.p, n * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

int flm_op_matmul_q(int qt, float* out, const void* W, const float* sW, const void* X, const float* sX, int m, int n, int w, int gs) {
    if (!out || !W || !sW || !X || !sX || m < 1 || n < 1 || w < 1 || gs != kGroup || n % kGroup) return FLM_ERR_INVALID;
    if (qt != FLM_QT_INT8 && qt != FLM_QT_INT16) return FLM_ERR_UNSUPPORTED;
    const size_t e = esz_of(qt), sn = n / kGroup;
    DevBuf dW, dsW, dX, dsX, dsXT, dsWT, dO;
    if (dW.alloc((size_t)m * n * e) || dsW.alloc((size_t)m * sn * 4) || dX.alloc((size_t)w * n * e) || dsX.alloc((size_t)w * sn * 4) || dsXT.alloc((size_t)w * sn * 4 + 64) || dsWT.alloc((size_t)m * sn * 4) || dO.alloc((size_t)w * m * 4)) return FLM_ERR_OOM;
    {   // the activation scales once more, group-major (k_rows_prologue writes both layouts on the prompt path)
        std::vector<float> t((size_t)w * sn);
        for (int b = 0; b < w; ++b) for (size_t g = 0; g < sn; ++g) t[g * w + b] = sX[(size_t)b * sn + g];
        OPC(hipMemcpy(dsXT.p, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> tw((size_t)m * sn);
        for (int r = 0; r < m; ++r) for (size_t g = 0; g < sn; ++g) tw[g * m + r] = sW[(size_t)r * sn + g];
        OPC(hipMemcpy(dsWT.p, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
    }
    OPC(hipMemcpy(dW.p, W, (size_t)m * n * e, hipMemcpyHostToDevice)); OPC(hipMemcpy(dsW.p, sW, (size_t)m * sn * 4, hipMemcpyHostToDevice));
    OPC(hipMemcpy(dX.p, X, (size_t)w * n * e, hipMemcpyHostToDevice)); OPC(hipMemcpy(dsX.p, sX, (size_t)w * sn * 4, hipMemcpyHostToDevice));
    int dev = 0, cus = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const char* gv = getenv("FLM_OP_GEMM");                     // tests: 0 .. 3 = launch_gemm's use_mfma, "gemv" = a GEMV per batch row
    if (w >= 16 && !(gv && !strcmp(gv, "gemv"))) {
        // the batched path the prompt takes (quant::matmul with w > 1, quant_operators.cpp:252-284): one tile kernel
        GemmArgs g{dW.p, dsW.as<float>(), dX.p, dsX.as<float>(), dO.as<float>(), m, n, m, w, dsXT.as<float>(), dsWT.as<float>()};
        const int um = gv ? atoi(gv) : 1;
        int r = launch_gemm_store(nullptr, 0, qt, g, um);
        if (r) return r;
    } else {
        for (int b = 0; b < w; ++b) {
            GemvArgs a{}; a.W = dW.p; a.sW = dsW.as<float>(); a.n = n; a.items = m;
            a.xq = (const char*)dX.p + (size_t)b * n * e; a.xs = dsX.as<float>() + (size_t)b * sn; a.out = dO.as<float>() + (size_t)b * m;
            int r = launch_gemv<PRO_NONE, EPI_STORE>(nullptr, 0, qt, a, gemv_grid(cus, 1, m, 1)); if (r) return r;
        }
    }
    OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(out, dO.p, (size_t)w * m * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

/* sample_argmax (sampler.cpp:36-47) as k_argmax_advance evaluates it: first maximum wins */
int flm_op_argmax(const float* logits, int n, int32_t* idx) {
    if (!logits || !idx || n < 1) return FLM_ERR_INVALID;
    DevBuf dl, dst, dout;
    if (dl.alloc((size_t)n * 4) || dst.alloc(sizeof(DecodeState)) || dout.alloc(16)) return FLM_ERR_OOM;
    OPC(hipMemcpy(dl.p, logits, (size_t)n * 4, hipMemcpyHostToDevice));
    OPC(hipMemset(dst.p, 0, sizeof(DecodeState)));
    hipLaunchKernelGGL(k_argmax_advance, dim3(1), dim3(1024), 0, 0, (const float*)dl.as<float>(), n, dst.as<DecodeState>(), dout.as<int>(), 0, 4);
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(idx, dout.p, 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

int flm_op_swiglu(float* xo, const float* xr, size_t n) {
    if (!xo || !xr || n == 0) return FLM_ERR_INVALID;
    DevBuf a, b; if (a.alloc(n * 4) || b.alloc(n * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(a.p, xo, n * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(b.p, xr, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_swiglu, dim3(256), dim3(256), 0, 0, a.as<float>(), (const float*)b.as<float>(), n);
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(xo, a.p, n * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

int flm_op_rope(float* o, const float* x, int n_dims, int pos) {
    if (!o || !x || n_dims < 2 || n_dims % 2 || pos < 0) return FLM_ERR_INVALID;
    std::vector<float> cs, sn; build_rope_table(n_dims, pos + 1, cs, sn);
    DevBuf dx, dout, dc, dsn; const size_t h = n_dims / 2;
    if (dx.alloc(n_dims * 4) || dout.alloc(n_dims * 4) || dc.alloc(h * 4) || dsn.alloc(h * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(dx.p, x, n_dims * 4, hipMemcpyHostToDevice));
    OPC(hipMemcpy(dc.p, cs.data() + (size_t)pos * h, h * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(dsn.p, sn.data() + (size_t)pos * h, h * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_rope, dim3((unsigned)((h + 63) / 64)), dim3(64), 0, 0, dout.as<float>(), (const float*)dx.as<float>(), n_dims, (const float*)dc.as<float>(), (const float*)dsn.as<float>());
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(o, dout.p, n_dims * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

int flm_op_softmax(float* x, int n) {
    if (!x || n < 1) return FLM_ERR_INVALID;
    DevBuf d; if (d.alloc((size_t)n * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(d.p, x, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_softmax, dim3(1), dim3(kBlock), 0, 0, d.as<float>(), n);
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(x, d.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

int flm_op_attention(float* out, float* kc, float* vc, const float* q, const float* k, const float* v,
                     int n_heads, int hs, int max_seq, int pos) {
    if (!out || !kc || !vc || !q || !k || !v || n_heads < 1 || hs < 32 || hs > 256 || hs % 8 || pos < 0 || pos >= max_seq) return FLM_ERR_INVALID;
    const size_t nd = (size_t)n_heads * hs, nc = (size_t)n_heads * max_seq * hs, h2 = hs / 2;
    std::vector<float> cs, sn; build_rope_table(hs, pos + 1, cs, sn);
    DevBuf dq, dk, dv, dkc, dvc, dout, dc, dsn, dpos;
    if (dq.alloc(nd * 4) || dk.alloc(nd * 4) || dv.alloc(nd * 4) || dkc.alloc(nc * 4) || dvc.alloc(nc * 4) || dout.alloc(nd * 4) ||
        dc.alloc(h2 * 4) || dsn.alloc(h2 * 4) || dpos.alloc(4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(dq.p, q, nd * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(dk.p, k, nd * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(dv.p, v, nd * 4, hipMemcpyHostToDevice));
    OPC(hipMemcpy(dkc.p, kc, nc * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(dvc.p, vc, nc * 4, hipMemcpyHostToDevice));
    OPC(hipMemcpy(dc.p, cs.data() + (size_t)pos * h2, h2 * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(dsn.p, sn.data() + (size_t)pos * h2, h2 * 4, hipMemcpyHostToDevice));
    OPC(hipMemcpy(dpos.p, &pos, 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_kv_append, dim3((unsigned)((nd / 2 + 255) / 256)), dim3(256), 0, 0, dq.as<float>(), (const float*)dk.as<float>(), (const float*)dv.as<float>(),
                       dkc.as<float>(), dvc.as<float>(), (const float*)dc.as<float>(), (const float*)dsn.as<float>(), n_heads, hs, max_seq, pos);
    OPC(hipGetLastError());
    AttnArgs a{}; a.q = dq.as<float>(); a.kcache = dkc.as<float>(); a.vcache = dvc.as<float>(); a.pos_ptr = dpos.as<int>(); a.hs = hs; a.max_seq = max_seq;
    a.out = dout.as<float>();
    // tests: FLM_OP_ATTN_PARTS = G spreads every head over G workgroups (the long-context path of the decode loop)
    int G = (getenv("FLM_OP_ATTN_PARTS") && atoi(getenv("FLM_OP_ATTN_PARTS")) > 1) ? hs / kSplitDims : 1;
    if (G < 2 || hs % kSplitDims || hs > 128 || n_heads * G > 256 || max_seq > kSplitMaxSeq) G = 1;
    DevBuf dsc, dfl, derr;
    if (dsc.alloc((size_t)n_heads * max_seq * 4) || dfl.alloc(256 * 64) || derr.alloc(64)) return FLM_ERR_OOM;
    OPC(hipMemset(dfl.p, 0, 256 * 64)); OPC(hipMemset(derr.p, 0, 64));
    a.G = G; a.sc_global = dsc.as<float>(); a.flag_sc = dfl.as<unsigned>(); a.epoch = 1; a.err = derr.as<int>();
    if (G > 1) hipLaunchKernelGGL(k_attn_decode<true>, dim3(n_heads * G), dim3(kAttnBlock), attn_lds_bytes(max_seq, hs, true), 0, a);
    else       hipLaunchKernelGGL(k_attn_decode<false>, dim3(n_heads), dim3(kAttnBlock), attn_lds_bytes(max_seq, hs, false), 0, a);
    OPC(hipGetLastError());
    OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(out, dout.p, nd * 4, hipMemcpyDeviceToHost));
    OPC(hipMemcpy(kc, dkc.p, nc * 4, hipMemcpyDeviceToHost)); OPC(hipMemcpy(vc, dvc.p, nc * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

/* elementary functions exactly as the kernels evaluate them (tests pin them against the host's IEEE results) */
int flm_op_math(int fn, float* x, const float* y, size_t n) {
    if (!x || n == 0 || fn < 0 || fn > 5 || (fn >= 2 && !y)) return FLM_ERR_INVALID;
    DevBuf d, e; if (d.alloc(n * 4) || e.alloc(n * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(d.p, x, n * 4, hipMemcpyHostToDevice));
    if (y) OPC(hipMemcpy(e.p, y, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_math, dim3(1024), dim3(256), 0, 0, fn, d.as<float>(), (const float*)e.as<float>(), n);
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(x, d.p, n * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}
int flm_op_expf(float* x, size_t n) { return flm_op_math(0, x, nullptr, n); }

} // extern "C"
