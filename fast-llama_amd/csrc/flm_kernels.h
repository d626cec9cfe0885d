// flm_kernels.h -- hand-written gfx950 (CDNA4) kernels for the fast-llama per-token hot path.
//
// Everything here is written for wave64 / MI355X only.  Reference citations are paths inside
// CoderLSF/fast-llama (the CPU engine whose arithmetic these kernels reproduce).
//
// DESIGN RULE: every fp32 value is produced with the SAME operations in the SAME order as the
// reference's x86 build (-O3 -mfma, FMA-contracted), so results are BIT-IDENTICAL to the CPU path.
// This is not pedantry: the reference quantizer q = trunc(x / (max|x|/127)) puts the largest
// element of every 64-group exactly on a truncation boundary (x_max/scale = 127 +- 1 ulp), so a
// 1-ulp difference anywhere upstream flips int8 values 126 <-> 127 chaotically and logits drift by
// 1e-2 -- far outside the 1e-3 parity bound.  Integer work (the int8/int16 dots, max, argmax) is
// order-free and fully parallel; every fp32 accumulation is a chain in reference order:
//   * GEMV      : exact int32 group dots in parallel, then acc = fma(sW*sX, float(dot_g), acc), g ascending
//   * rmsnorm   : sum of squares as the reference's 4 strided SSE lanes, each a sequential FMA chain
//   * attention : q.k as 8 strided lanes + sequential lane sum; glibc-exact expf; sequential softmax
//                 sum; weighted V sum sequential over positions
//
// Kernel inventory (one decode token = embed + L x {qkv, attn, attn_o, ffn13, ffn2} + cls + argmax):
//   k_gemv<QT,PRO,EPI,XR> group-quantized GEMV, HBM-bound; fused prologue (rmsnorm+quantize | quantize)
//                         and epilogue (store | residual add | SwiGLU | RoPE + KV-cache append)
//   k_attn_decode         fp32 single-query attention over the fp32 KV cache
//   k_embed, k_argmax_advance
// plus small op-level kernels that expose the same __device__ functions to the parity tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Bit-exactness hygiene (see DESIGN RULE above):
//  * no implicit FMA contraction anywhere in this TU -- every fused multiply-add is written as fmaf()/fma().
//    (HIP's __fmul_rn/__fadd_rn are plain operators and WOULD be contracted under the default
//    -ffp-contract=fast; __graft_entry__.build() also passes -ffp-contract=off.)
//  * sqrt via __builtin_sqrtf / division via operator/ : IEEE-correct under hipcc's default
//    -fhip-fp32-correctly-rounded-divide-sqrt.  HIP's __fsqrt_rn maps to the 1-ulp native sqrt: never used.
#pragma clang fp contract(off)

namespace flm {

constexpr int kWave = 64;
constexpr int kBlock = 256;           // attention / small kernels: 4 waves per workgroup
constexpr int kGemvBlock = 1024;      // GEMV: 16 waves = ONE workgroup per CU (<= 128 VGPRs): one activation prologue (and one
                                      // sequential rmsnorm chain) per CU instead of two competing for a SIMD
constexpr int kWavesPerBlock = kGemvBlock / kWave;
constexpr int kGroup = 64;            // quantization group (QUANT_GROUP_SIZE, the only value the reference uses)

enum { QT_INT16 = 1, QT_INT8 = 2 };
enum Prologue { PRO_NONE = 0, PRO_QUANT = 1, PRO_RMSNORM_QUANT = 2 };
enum Epilogue { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2, EPI_ROPE_KV = 3 };

template <int QT> struct QTraits;
template <> struct QTraits<QT_INT8>  { using elem = int8_t;  static constexpr int kEsz = 1; static constexpr int kEPC = 16; static constexpr float kF = 127.0f; };
template <> struct QTraits<QT_INT16> { using elem = int16_t; static constexpr int kEsz = 2; static constexpr int kEPC = 8;  static constexpr float kF = 5792.0f; };
// kEPC = elements per 16-byte chunk; lanes per quant group = 64 / kEPC (4 for int8, 8 for int16)

// ------------------------------------------------------------------------------------------
// block reductions for ORDER-FREE quantities only (max): wave64 xor butterflies
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// max over each aligned group of 16 lanes (one DPP row), result in all 16 lanes; order-free, no LDS
__device__ __forceinline__ float row16_max(float v) {
    const int i0 = __float_as_int(v);
    float t = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(i0, i0, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true)));
    int i1 = __float_as_int(t);
    t = fmaxf(t, __int_as_float(__builtin_amdgcn_update_dpp(i1, i1, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true)));
    i1 = __float_as_int(t);
    t = fmaxf(t, __int_as_float(__builtin_amdgcn_update_dpp(i1, i1, 0x141 /* row_half_mirror */, 0xF, 0xF, true)));
    i1 = __float_as_int(t);
    t = fmaxf(t, __int_as_float(__builtin_amdgcn_update_dpp(i1, i1, 0x140 /* row_mirror */, 0xF, 0xF, true)));
    return t;
}
// exact integer sum over the 4 lanes of a quad (DPP quad_perm, no LDS)
__device__ __forceinline__ int quad_sum(int p) {
    p += __builtin_amdgcn_update_dpp(0, p, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
    p += __builtin_amdgcn_update_dpp(0, p, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true);
    return p;
}

// ------------------------------------------------------------------------------------------
// scalar pieces shared by the fused kernels and the op-level test kernels
// ------------------------------------------------------------------------------------------
// glibc 2.35 expf (sysdeps/ieee754/flt-32/e_expf.c, the Arm optimized-routines algorithm): the
// reference calls libm's expf in softmax_sisd (tf_operators.cpp:180) and swiglu (x86_simd.cpp:1768).
// Evaluated in double exactly as libm does: z = x*N/ln2, k = round(z), 2^(k/N) from a 32-entry table,
// cubic in r = z - k.  The table is tab[i] = bits(2^(i/32)) - (i << 47), recomputed at 60 digits;
// this routine was checked bit-for-bit against libm's expf on 6e7 inputs on the build host
// (tools/check_expf.c) and is checked again on the GPU by tests/test_gpu_ops.py::test_expf_bit_exact.
__device__ const unsigned long long kExp2fTab[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};

__device__ __forceinline__ float expf_ref(float x) {
    const uint32_t ix = __float_as_uint(x);
    const uint32_t abstop = (ix >> 20) & 0x7ff;
    if (abstop >= (0x42b00000u >> 20)) {                       // |x| >= 88 or NaN/inf
        if (ix == 0xff800000u) return 0.0f;                    // -inf
        if (abstop >= (0x7f800000u >> 20)) return x + x;       // +inf, NaN
        if (x > 0x1.62e42ep6f) return INFINITY;                // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;                   // underflow
        if (x < -0x1.9d1d9ep6f) return __fmul_rn(0x1.4p-75f, 0x1.4p-75f);   // __math_may_uflowf
    }
    constexpr double N = 32.0;
    constexpr double InvLn2N = 0x1.71547652b82fep+0 * N, SHIFT = 0x1.8p+52;
    constexpr double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
    double z = __dmul_rn(InvLn2N, (double)x);
    double kd = __dadd_rn(z, SHIFT);
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd = __dsub_rn(kd, SHIFT);
    const double r = __dsub_rn(z, kd);
    const unsigned long long t = kExp2fTab[ki % 32] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    z = __fma_rn(C0, r, C1);
    const double r2 = __dmul_rn(r, r);
    double y = __fma_rn(C2, r, 1.0);
    y = __fma_rn(z, r2, y);
    y = __dmul_rn(y, s);
    return (float)y;
}

// quant::quantize<T> element step (src/blas/quant_operators.cpp:26-47): q = (T)(x / r), C truncation.
// r == 0 (all-zero group): x/r is NaN; the x86 reference yields 0, stated explicitly here.
__device__ __forceinline__ int quant_elem(float x, float r) {
    float t = __fdiv_rn(x, r);          // IEEE-correct fp32 divide, never the fast reciprocal
    return (r == 0.0f) ? 0 : (int)t;    // v_cvt_i32_f32 truncates toward zero
}
// simd::rmsnorm scale (src/platforms/arch/x86_simd.cpp:1754-1756): r = float(1. / sqrtf(ss/n + 1e-5f))
__device__ __forceinline__ float rms_scale(float ss, int n) {
    float v = __fadd_rn(__fdiv_rn(ss, (float)n), 1e-5f);
    return (float)(1.0 / (double)__builtin_sqrtf(v));
}
// simd::swiglu (x86_simd.cpp:1766-1770): xo / (1. + expf(-xo)) * xr evaluated in double, rounded to float
__device__ __forceinline__ float swiglu_elem(float a, float b) {
    const double e = (double)expf_ref(-a);
    return (float)__dmul_rn(__ddiv_rn((double)a, __dadd_rn(1.0, e)), (double)b);
}
// rope_v2 pair (src/blas/tf_operators.cpp:398-401) with the reference build's FMA contraction
__device__ __forceinline__ void rope_pair(float x0, float x1, float c, float s, float& o0, float& o1) {
    o0 = __fmaf_rn(x0, c, -__fmul_rn(x1, s));
    o1 = __fmaf_rn(x0, s, __fmul_rn(x1, c));
}

typedef int v4i __attribute__((ext_vector_type(4)));      // native vector: usable with __builtin_nontemporal_load
__device__ __forceinline__ int dot16_i8(const v4i& w, const v4i& a, int acc) {
    acc = __builtin_amdgcn_sdot4(w.x, a.x, acc, false);
    acc = __builtin_amdgcn_sdot4(w.y, a.y, acc, false);
    acc = __builtin_amdgcn_sdot4(w.z, a.z, acc, false);
    acc = __builtin_amdgcn_sdot4(w.w, a.w, acc, false);
    return acc;
}
typedef short short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ short2_t as_short2(int v) { return __builtin_bit_cast(short2_t, v); }   // by value: bit_cast of a vector-element lvalue miscompiles
__device__ __forceinline__ int dot8_i16(const v4i& w, const v4i& a, int acc) {
    const int w0 = w.x, w1 = w.y, w2 = w.z, w3 = w.w, a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w;
    acc = __builtin_amdgcn_sdot2(as_short2(w0), as_short2(a0), acc, false);
    acc = __builtin_amdgcn_sdot2(as_short2(w1), as_short2(a1), acc, false);
    acc = __builtin_amdgcn_sdot2(as_short2(w2), as_short2(a2), acc, false);
    acc = __builtin_amdgcn_sdot2(as_short2(w3), as_short2(a3), acc, false);
    return acc;
}
template <int QT> __device__ __forceinline__ int dot_chunk(const v4i& w, const v4i& a) {
    if constexpr (QT == QT_INT8) return dot16_i8(w, a, 0); else return dot8_i16(w, a, 0);
}

// ------------------------------------------------------------------------------------------
// GEMV argument block
// ------------------------------------------------------------------------------------------
struct GemvArgs {
    // weights: row-major [rows][n] quantized values + natural-layout scales [rows][n/64]
    const void*  W;   const float* sW;          // EPI_SWIGLU: W = W1 (gate), W2nd = W3 (up)
    const void*  W2nd; const float* sW2nd;
    int n;                                      // K (columns), multiple of 64
    int items;                                  // rows (STORE/RESIDUAL), hidden (SWIGLU), row pairs (ROPE_KV)
    int rows_per_pass;                          // R: rows one workgroup reduces per pass (multiple of 64 >> cb_shift and of 2)
    int cb_shift;                               // log2(CB): a 1 KiB wave load covers (64 >> cb_shift) rows x CB 16-byte chunks
    // prologue inputs
    const float* x;                             // fp32 activation [n]          (QUANT / RMSNORM_QUANT)
    const float* norm_w;                        // rmsnorm weight [n]           (RMSNORM_QUANT)
    const void*  xq; const float* xs;           // pre-quantized activation     (NONE)
    // epilogue outputs
    float* out;                                 // STORE: out[row]; RESIDUAL: out[row] += ; SWIGLU: hd[i]; ROPE_KV: q[row]
    float* kcache; float* vcache;               // ROPE_KV: this layer's caches [heads][max_seq][hs]
    const float* rope_cos; const float* rope_sin; // [max_seq][hs/2]
    const int* pos_ptr;                         // device-resident position
    int dim; int kv_dim; int max_seq; int hs;   // ROPE_KV geometry
    // debugging taps used by the op-level exports (may be null)
    void* dbg_xq; float* dbg_xs; float* dbg_xn;
    int ablate;                                 // perf exploration only (results invalid when != 0): 1 no group chain, 2 no rmsnorm chain, 4 no weight loads, 8 no dots, 16 return immediately, 32 return after prologue
};

constexpr int kMaxBlk = 8;             // 1 KiB wave loads in flight per wave and pass (8 KiB/wave, 64 KiB/workgroup, 128 KiB/CU)
constexpr int kChainPad = 4;           // LDS row padding (dwords) of the per-wave chain scratch: no bank conflicts, keeps 16-B alignment

// LDS layout: [xq : n*esz] [xs : n/64 floats, padded to 16 B] [red : 16 floats] [scratch]
// scratch = max( rmsnorm transpose staging 4n bytes ,
//                2 buffers x { dF[R][gstride] float(group dot), sP[R][gstride] sW*sX } )
struct GemvLds {
    int off_xs, off_red, off_scr;     // byte offsets
    int gstride;                      // dwords per row strip; multiple of 4 (16-B aligned strips), +4 pad against bank conflicts
    int buf_bytes;                    // one {dF, sP} buffer
    int total;                        // bytes
};
__host__ __device__ inline GemvLds gemv_lds_layout(int n, int esz, bool norm, int R) {
    GemvLds L;
    const int sn = n / kGroup;
    L.off_xs = n * esz;
    L.off_red = L.off_xs + ((sn * 4 + 15) & ~15);
    L.off_scr = L.off_red + 64;
    L.gstride = ((sn + 3) & ~3) + kChainPad;
    L.buf_bytes = 2 * (R + 1) * L.gstride * 4;                                // +1: dummy strip that absorbs the writes of blocks past the pass
    int scratch = 2 * L.buf_bytes;
    if (norm && n * 4 + 256 > scratch) scratch = n * 4 + 256;                   // +256: the chain ring reads up to 32 floats past the last strip
    L.total = L.off_scr + scratch;
    return L;
}

// ------------------------------------------------------------------------------------------
// Prologue: produce the quantized activation vector in LDS.  Every workgroup recomputes it
// (n <= 16K floats out of L2) so that no separate norm/quantize kernel sits on the critical path.
//   RMSNORM_QUANT == x2.rmsnorm(x1, w) ; qx.quantize(x2)   (transformer.cpp:132-134, 144-146, 155-156)
//   QUANT         == qx.quantize(x2) / qh.quantize(hd)     (transformer.cpp:138, 149)
// Thread t owns elements 4t..4t+3 (+1024 per round): 16 consecutive lanes own one 64-group, so the
// group max (order-free) is a 16-lane xor-butterfly.
// The first XR rounds of x (and of the norm weight) are handed in as registers that the caller
// loaded BEFORE issuing its first batch of weight loads: loads return in issue order, so an x load
// issued behind 32 HBM weight loads would make the whole prologue wait for them.
// ------------------------------------------------------------------------------------------
template <int QT, int PRO, int XR>
__device__ __forceinline__ void gemv_preload(const GemvArgs& a, float4 (&xv)[XR > 0 ? XR : 1], float4 (&wv)[XR > 0 ? XR : 1]) {
    if constexpr (PRO == PRO_QUANT || PRO == PRO_RMSNORM_QUANT) {
        // branch-free: raw buffer loads, elements past n read as zero
        typedef float v4f __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.n * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRO == PRO_RMSNORM_QUANT ? a.norm_w : a.x), 0, a.n * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int off = (threadIdx.x * 4 + i * kGemvBlock * 4) * 4;
            const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
            xv[i] = make_float4(v.x, v.y, v.z, v.w);
            if constexpr (PRO == PRO_RMSNORM_QUANT) {
                const v4f u = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rn, off, 0, 0));
                wv[i] = make_float4(u.x, u.y, u.z, u.w);
            }
        }
    }
}

template <int QT, int PRO, int XR>
__device__ __forceinline__ void gemv_prologue(const GemvArgs& a, char* lds, float4 (&xv)[XR > 0 ? XR : 1], float4 (&wv)[XR > 0 ? XR : 1]) {
    using T = QTraits<QT>;
    const int n = a.n;
    const int tid = threadIdx.x;
    const GemvLds L = gemv_lds_layout(n, T::kEsz, PRO == PRO_RMSNORM_QUANT, a.rows_per_pass);
    char*  xq = lds;
    float* xs = reinterpret_cast<float*>(lds + L.off_xs);
    float* red = reinterpret_cast<float*>(lds + L.off_red);
    float* scratch = reinterpret_cast<float*>(lds + L.off_scr);

    if constexpr (PRO == PRO_NONE) {
        // copy pre-quantized activation (op-level matmul and generic callers)
        const int nb16 = n * T::kEsz / 16;
        for (int c = tid; c < nb16; c += kGemvBlock)
            reinterpret_cast<int4*>(xq)[c] = reinterpret_cast<const int4*>(a.xq)[c];
        for (int g = tid; g < n / kGroup; g += kGemvBlock) xs[g] = a.xs[g];
        __syncthreads();
        return;
    } else {
        const int rounds = (n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float r = 1.0f;
        if constexpr (PRO == PRO_RMSNORM_QUANT) {
            // simd::square_sum -> square_sum_avx128 (x86_simd.cpp:942-960; the AVX2 branch is dead, :1093):
            // lane c of 4 accumulates x[c], x[c+4], x[c+8]... by FMA, then res = ((0+l0)+l1)+l2)+l3.
            // Stage x transposed ([4][n/4]) so that 4 threads can each walk one strided lane sequentially.
            const int n4 = n / 4;
            auto stage = [&](int i, const float4& v) {
                const int k = tid + i * kGemvBlock;
                if (k < n4) { scratch[k] = v.x; scratch[n4 + k] = v.y; scratch[2 * n4 + k] = v.z; scratch[3 * n4 + k] = v.w; }
            };
#pragma unroll
            for (int i = 0; i < XR; ++i) { if (i < rounds) stage(i, xv[i]); }
            for (int i = XR; i < rounds; ++i) {
                const int e = tid * 4 + i * kGemvBlock * 4;
                if (e < n) stage(i, *reinterpret_cast<const float4*>(a.x + e));
            }
            __syncthreads();
            if (tid < 4 && !(a.ablate & 2)) {
                // 4 strided lanes, each a strictly sequential FMA chain; LDS reads are issued 32 values
                // ahead of their use so the chain runs at FMA latency, not LDS latency.
                const float* p = scratch + tid * n4;
                float l = 0.f;
                int k = 0;
#define FLM_SQ4(v) l = __fmaf_rn(v.x, v.x, l); l = __fmaf_rn(v.y, v.y, l); l = __fmaf_rn(v.z, v.z, l); l = __fmaf_rn(v.w, v.w, l);
#define FLM_STEP(q, off) FLM_SQ4(q) q = *reinterpret_cast<const float4*>(pp + (off)); __builtin_amdgcn_sched_barrier(0);
                if (n4 >= 32) {
                    // A lone wave issues roughly one instruction every ~5 cycles, whatever its kind, so the
                    // loop body is nothing but the 32 dependent FMAs and 8 LDS reads with immediate offsets:
                    // a ring of 8 float4 registers, every read issued 28 FMAs before its first use (the
                    // sched_barriers pin that order).  Reads run up to 32 floats past a lane's strip: the
                    // staging area is sized for that (gemv_lds_layout) and those values are never consumed.
                    const float* pp = p;
                    float4 q0 = *reinterpret_cast<const float4*>(pp), q1 = *reinterpret_cast<const float4*>(pp + 4), q2 = *reinterpret_cast<const float4*>(pp + 8), q3 = *reinterpret_cast<const float4*>(pp + 12);
                    float4 q4 = *reinterpret_cast<const float4*>(pp + 16), q5 = *reinterpret_cast<const float4*>(pp + 20), q6 = *reinterpret_cast<const float4*>(pp + 24), q7 = *reinterpret_cast<const float4*>(pp + 28);
                    __builtin_amdgcn_sched_barrier(0);
                    for (; k + 32 <= n4; k += 32, pp += 32) {
                        FLM_STEP(q0, 32) FLM_STEP(q1, 36) FLM_STEP(q2, 40) FLM_STEP(q3, 44)
                        FLM_STEP(q4, 48) FLM_STEP(q5, 52) FLM_STEP(q6, 56) FLM_STEP(q7, 60)
                    }
                    // the ring now holds p[k .. k+31]
                    if (k + 4 <= n4) { FLM_SQ4(q0) k += 4; } if (k + 4 <= n4) { FLM_SQ4(q1) k += 4; } if (k + 4 <= n4) { FLM_SQ4(q2) k += 4; } if (k + 4 <= n4) { FLM_SQ4(q3) k += 4; }
                    if (k + 4 <= n4) { FLM_SQ4(q4) k += 4; } if (k + 4 <= n4) { FLM_SQ4(q5) k += 4; } if (k + 4 <= n4) { FLM_SQ4(q6) k += 4; }
                }
#undef FLM_STEP
#undef FLM_SQ4
                for (; k < n4; ++k) l = __fmaf_rn(p[k], p[k], l);
                red[8 + tid] = l;
            }
            __syncthreads();
            const float ss = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, red[8]), red[9]), red[10]), red[11]);
            r = rms_scale(ss, n);
            __syncthreads();                                   // scratch is reused by the GEMV waves below
        }
        // one round: (normalise,) group max over 16 lanes, quantize, pack into LDS
        auto round = [&](int i, float4 v, float4 w) {
            const int e = tid * 4 + i * kGemvBlock * 4;
            const bool act = e < n;
            if constexpr (PRO == PRO_RMSNORM_QUANT) {
                // multiply_avx256 (x86_simd.cpp:1360-1372): (x*w)*r
                v.x = __fmul_rn(__fmul_rn(v.x, w.x), r); v.y = __fmul_rn(__fmul_rn(v.y, w.y), r);
                v.z = __fmul_rn(__fmul_rn(v.z, w.z), r); v.w = __fmul_rn(__fmul_rn(v.w, w.w), r);
            }
            if (act && a.dbg_xn && blockIdx.x == 0) *reinterpret_cast<float4*>(a.dbg_xn + e) = v;
            // group max over the 16 lanes that share this 64-element group (order-free, exact)
            const float mx = row16_max(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            const float sc = __fdiv_rn(mx, T::kF);           // scale = max|x| / F
            if (act) {
                const int q0 = quant_elem(v.x, sc), q1 = quant_elem(v.y, sc), q2 = quant_elem(v.z, sc), q3 = quant_elem(v.w, sc);
                if constexpr (QT == QT_INT8) {
                    const uint32_t pk = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
                    *reinterpret_cast<uint32_t*>(xq + e) = pk;
                } else {
                    uint2 pk;
                    pk.x = (uint32_t)(q0 & 0xffff) | ((uint32_t)(q1 & 0xffff) << 16);
                    pk.y = (uint32_t)(q2 & 0xffff) | ((uint32_t)(q3 & 0xffff) << 16);
                    *reinterpret_cast<uint2*>(xq + (size_t)e * 2) = pk;
                }
                if ((tid & 15) == 0) xs[e / kGroup] = sc;
            }
        };
#pragma unroll
        for (int i = 0; i < XR; ++i) { if (i < rounds) round(i, xv[i], wv[i]); }
        for (int i = XR; i < rounds; ++i) {
            const int e = tid * 4 + i * kGemvBlock * 4;
            float4 v = z4, w = z4;
            if (e < n) {
                v = *reinterpret_cast<const float4*>(a.x + e);
                if constexpr (PRO == PRO_RMSNORM_QUANT) w = *reinterpret_cast<const float4*>(a.norm_w + e);
            }
            round(i, v, w);
        }
        __syncthreads();
        if (a.dbg_xq && blockIdx.x == 0) {
            const int nb4 = n * T::kEsz / 4;
            for (int c = tid; c < nb4; c += kGemvBlock) reinterpret_cast<uint32_t*>(a.dbg_xq)[c] = reinterpret_cast<uint32_t*>(xq)[c];
            for (int g = tid; g < n / kGroup; g += kGemvBlock) a.dbg_xs[g] = xs[g];
        }
    }
}

// ------------------------------------------------------------------------------------------
// The GEMV.  quant::matmul<T> at w == 1 (src/blas/quant_operators.cpp:252-284):
//     out[r] = sum_g (sW[r,g] * sX[g]) * float( sum_{k<64} W[r,64g+k] * X[64g+k] ),   g ASCENDING, FMA per group
//
// A workgroup (8 waves) reduces R rows per pass.  The R x K tile is cut into 1 KiB blocks of
// (RB rows x CB chunks of 16 B), RB*CB = 64, CB = the largest power of two dividing K/16 (so a block is
// one fully coalesced global_load_dwordx4 per wave and every lane is busy for any K); blocks are dealt
// round-robin to the 8 waves and ALL of a wave's blocks (<= kMaxBlk) are in flight before the
// prologue runs.  Then
//   1. int32 dot per 16-byte chunk (v_dot4 / v_dot2), exact;
//   2. DPP sum over the 4 (int8) / 8 (int16) lanes of a quant group -> the group's int32 dot, exact;
//   3. group leaders park float(dot) and s = sW*sX in LDS strips dF[row][g], sP[row][g];
//   4. after ONE workgroup barrier, one wave walks the strips, lane r = row r:
//        acc = fma(sP[g], dF[g], acc), g ascending -- the reference's summation order, bit-identical --
//      amortising the sequential fp32 chain over R rows instead of paying it per row;
//   5. the same lanes run the epilogue (coalesced stores).
// The strips are double buffered, so the other waves are already in the next pass's dots.
// ------------------------------------------------------------------------------------------
template <int QT, int PRO, int EPI, int XR>
__global__ void __launch_bounds__(kGemvBlock, 4) k_gemv(const GemvArgs a) {
    using T = QTraits<QT>;
    typedef unsigned int u32;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const u32 n = a.n;
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr u32 LPGS = (T::kEPC == 16) ? 2 : 3;                             // log2(lanes per quant group): 4 | 8 lanes
    constexpr u32 LPG = 1u << LPGS;
    const u32 rowbytes = n * T::kEsz, nchunks = rowbytes / 16, sn = n / kGroup;
    const u32 cbs = a.cb_shift, CB = 1u << cbs, RB = 64u >> cbs;
    const u32 nbc = nchunks >> cbs;                                           // chunk blocks per row
    const u32 R = a.rows_per_pass;
    // SWIGLU passes hold R/2 rows of W1 followed by the R/2 matching rows of W3 (blocks never mix matrices)
    constexpr bool TWO = EPI == EPI_SWIGLU;
    const u32 TRm = TWO ? (u32)a.items : (u32)a.items * (EPI == EPI_ROPE_KV ? 2u : 1u);    // rows per matrix
    const u32 Rm = TWO ? R / 2 : R;                                           // rows of one matrix per pass
    const u32 npass = (TRm + Rm - 1) / Rm;
    const u32 NB = (R / RB) * nbc;                                            // blocks per pass
    const u32 nblk = (NB + kWavesPerBlock - 1 - wave) / kWavesPerBlock;       // blocks of this wave: b = wave + 8k
    const u32 rb = lane >> cbs, cb = lane & (CB - 1);
    // lane-constant parts of every address (the per-block parts are wave-uniform scalars)
    const u32 lane_woff = rb * rowbytes + cb * 16;                            // weights, bytes from the block base
    const u32 lane_soff = (rb * sn + (cb >> LPGS)) * 4;                       // scales
    const u32 lane_xoff = cb * 16;                                            // activation chunk in LDS
    const bool leader = (cb & (LPG - 1)) == 0;
    // block sequence of this wave: (row block, chunk block) = divmod(wave + 8k, nbc), advanced incrementally
    const u32 q8 = kWavesPerBlock / nbc, r8 = kWavesPerBlock - q8 * nbc;
    const u32 rbk0 = wave / nbc, cbk0 = wave - rbk0 * nbc;

    // 1. the activation (L2-resident) first, 2. then every weight block of the first pass, both
    // BEFORE the prologue: weights do not depend on the activation, so HBM latency (and the
    // sequential rmsnorm chain) overlap the weight stream.
    if (a.ablate & 16) return;
    float4 xv[XR > 0 ? XR : 1], nv[XR > 0 ? XR : 1];
    gemv_preload<QT, PRO, XR>(a, xv, nv);

    // Each wave owns blocks b = wave + 8k, k < nblk, of every pass and streams them in steps of H
    // blocks through two register sets (A, B): while one set is being reduced the other one and the
    // refill of the first are in flight, so >= H KiB per wave (64 KiB per CU at 16 waves) are always
    // outstanding and R is not limited by registers.
    constexpr int H = kMaxBlk / 2;                                             // blocks per step
    const u32 spp = ((NB + kWavesPerBlock - 1) / kWavesPerBlock + H - 1) / H;  // steps per pass (same for all waves)
    struct Set { v4i w[H]; float sw[H]; };
    struct Cursor { u32 pass, s, k, rbk, cbk; };                               // pass, step in pass, block counter, divmod(wave + 8k, nbc)
    auto cursor_init = [&](Cursor& c, u32 pass) { c.pass = pass; c.s = 0; c.k = 0; c.rbk = rbk0; c.cbk = cbk0; };
    auto cursor_next_block = [&](Cursor& c) { ++c.k; c.rbk += q8; c.cbk += r8; if (c.cbk >= nbc) { c.cbk -= nbc; ++c.rbk; } };
    auto cursor_end_step = [&](Cursor& c) { if (++c.s == spp) cursor_init(c, c.pass + gridDim.x); };

    // Weight and scale blocks are fetched with raw buffer loads: address = descriptor base + wave-uniform
    // scalar offset (the block) + lane-constant 32-bit offset: no per-load vector address arithmetic, no
    // branches.  A block past the pass (k >= nblk), past the last pass, or rows past the end of the
    // matrix get an offset outside the descriptor -> the load returns zero without touching memory.
    // "nt": each weight byte is read once per token.
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr int kRsrcFlags = 0x00020000;                                     // raw buffer, 32-bit data format (gfx9 family)
    const __amdgpu_buffer_rsrc_t rW  = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, (int)(TRm * rowbytes), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rS  = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.sW), 0, (int)(TRm * sn * 4), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(TWO ? a.W2nd : a.W), 0, (int)(TRm * rowbytes), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rS2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(TWO ? a.sW2nd : a.sW), 0, (int)(TRm * sn * 4), kRsrcFlags);
    auto load_step = [&](Set& S, Cursor& c) {
#pragma unroll
        for (int j = 0; j < H; ++j) {
            u32 vr = c.rbk * RB;                                             // first row of the block inside the pass
            bool second = false;
            if constexpr (TWO) { second = vr >= Rm; vr = second ? vr - Rm : vr; }
            const u32 row = c.pass * Rm + vr;
            const bool live = c.k < nblk && c.pass < npass && !(a.ablate & 4);
            const u32 wo = live ? row * rowbytes + ((c.cbk << cbs) * 16) : 0x80000000u;
            const u32 so = live ? (row * sn + ((c.cbk << cbs) >> LPGS)) * 4 : 0x80000000u;
            v4u wv; unsigned sv;
            if constexpr (TWO) {
                const __amdgpu_buffer_rsrc_t rw = second ? rW2 : rW, rs_ = second ? rS2 : rS;
                wv = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)lane_woff, (int)wo, 2); sv = __builtin_amdgcn_raw_buffer_load_b32(rs_, (int)lane_soff, (int)so, 2);
            } else {
                wv = __builtin_amdgcn_raw_buffer_load_b128(rW, (int)lane_woff, (int)wo, 2); sv = __builtin_amdgcn_raw_buffer_load_b32(rS, (int)lane_soff, (int)so, 2);
            }
            S.w[j] = __builtin_bit_cast(v4i, wv);
            S.sw[j] = __uint_as_float(sv);
            cursor_next_block(c);
        }
        cursor_end_step(c);
    };
    Set setA, setB;
    Cursor lc;                                                                 // load cursor, runs two steps ahead of the reduce cursor
    cursor_init(lc, blockIdx.x);
    load_step(setA, lc);
    load_step(setB, lc);

    gemv_prologue<QT, PRO, XR>(a, lds, xv, nv);
    if (a.ablate & 32) return;

    const GemvLds L = gemv_lds_layout(n, T::kEsz, PRO == PRO_RMSNORM_QUANT, R);
    const char*  xq = lds;
    const char*  xs = lds + L.off_xs;
    const u32 gstride = L.gstride;
    const u32 lane_goff = (rb * gstride + (cb >> LPGS)) * 4;                   // strip position of this lane's group, bytes
    const bool vec_ok = (sn % 4) == 0;                                         // 16-B LDS reads need whole float4s per strip
    int pos = 0;
    if constexpr (EPI == EPI_ROPE_KV) pos = *a.pos_ptr;

    // reduce one step: dots for all its blocks first (registers), then ONE leader-only region parks them.
    // Blocks past the pass hold zeros; their strip writes are redirected to the dummy strip (row R).
    auto reduce_step = [&](Set& S, Cursor c, char* dF, char* sP) {
        int d[H];
        {
            Cursor q = c;
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const v4i av = *reinterpret_cast<const v4i*>(xq + ((q.cbk << cbs) * 16) + lane_xoff);
                int t = (a.ablate & 8) ? 0 : quad_sum(dot_chunk<QT>(S.w[j], av));
                if constexpr (LPG == 8) t += __shfl_xor(t, 4, kWave);
                d[j] = t;
                cursor_next_block(q);
            }
        }
        if (leader) {
            Cursor q = c;
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const u32 g0 = (q.cbk << cbs) >> LPGS;                       // first group of the block (scalar)
                const bool live = q.k < nblk;
                const u32 so = ((live ? q.rbk * RB : R) * gstride + g0) * 4 + (live ? lane_goff : (cb >> LPGS) * 4);
                const float sx = *reinterpret_cast<const float*>(xs + g0 * 4 + (cb >> LPGS) * 4);
                *reinterpret_cast<float*>(dF + so) = (float)d[j];                          // exact int32 -> fp32, as "s * dot" does
                *reinterpret_cast<float*>(sP + so) = __fmul_rn(S.sw[j], sx);               // s = sW * sX (quant_operators.cpp:274)
                cursor_next_block(q);
            }
        }
    };

    // the end of a pass: one barrier, then ONE wave runs the fp32 chains of all R rows and the epilogue
    auto finish_pass = [&](u32 pass, u32 it, char* dF, char* sP) {
        const bool chain_wave = wave == (it & (kWavesPerBlock - 1));
        // epilogue operands of the chain wave, fetched before the barrier (lane r = row r of the pass)
        float resid = 0.f, rc = 0.f, rs = 0.f;
        const u32 row = pass * Rm + (TWO ? (lane < Rm ? lane : lane - Rm) : lane);   // row inside its matrix
        const bool rv = chain_wave && lane < R && row < TRm;
        if constexpr (EPI == EPI_RESIDUAL) { if (rv) resid = a.out[row]; }
        if constexpr (EPI == EPI_ROPE_KV) {
            if (rv && row < (u32)(a.dim + a.kv_dim)) {
                const u32 r2 = (row < (u32)a.dim ? row : row - a.dim) & ~1u;
                const u32 dd = r2 % (u32)a.hs;
                rc = a.rope_cos[(size_t)pos * (a.hs / 2) + dd / 2];
                rs = a.rope_sin[(size_t)pos * (a.hs / 2) + dd / 2];
            }
        }
        __syncthreads();
        if (!chain_wave) return;
        // ---- the reference's fp32 chain, lane r = row r: o[j] += s * dot (FMA), groups ascending.
        //      LDS reads run 8 groups ahead of the FMAs so the chain advances at FMA latency.
        float acc = 0.f;
        if (lane < R && !(a.ablate & 1)) {
            const float* dp = reinterpret_cast<const float*>(dF) + lane * gstride;
            const float* sp = reinterpret_cast<const float*>(sP) + lane * gstride;
            u32 g = 0;
            if (vec_ok) {
                float4 sa = {0.f, 0.f, 0.f, 0.f}, sb = sa, sc = sa, sd = sa, da = sa, db = sa, dc = sa, dd = sa;
                if (sn >= 8) { sa = *reinterpret_cast<const float4*>(sp); da = *reinterpret_cast<const float4*>(dp);
                               sb = *reinterpret_cast<const float4*>(sp + 4); db = *reinterpret_cast<const float4*>(dp + 4); }
                for (; g + 16 <= sn; g += 16) {
                    sc = *reinterpret_cast<const float4*>(sp + g + 8);  dc = *reinterpret_cast<const float4*>(dp + g + 8);
                    sd = *reinterpret_cast<const float4*>(sp + g + 12); dd = *reinterpret_cast<const float4*>(dp + g + 12);
                    acc = __fmaf_rn(sa.x, da.x, acc); acc = __fmaf_rn(sa.y, da.y, acc); acc = __fmaf_rn(sa.z, da.z, acc); acc = __fmaf_rn(sa.w, da.w, acc);
                    acc = __fmaf_rn(sb.x, db.x, acc); acc = __fmaf_rn(sb.y, db.y, acc); acc = __fmaf_rn(sb.z, db.z, acc); acc = __fmaf_rn(sb.w, db.w, acc);
                    if (g + 24 <= sn) { sa = *reinterpret_cast<const float4*>(sp + g + 16); da = *reinterpret_cast<const float4*>(dp + g + 16);
                                        sb = *reinterpret_cast<const float4*>(sp + g + 20); db = *reinterpret_cast<const float4*>(dp + g + 20); }
                    acc = __fmaf_rn(sc.x, dc.x, acc); acc = __fmaf_rn(sc.y, dc.y, acc); acc = __fmaf_rn(sc.z, dc.z, acc); acc = __fmaf_rn(sc.w, dc.w, acc);
                    acc = __fmaf_rn(sd.x, dd.x, acc); acc = __fmaf_rn(sd.y, dd.y, acc); acc = __fmaf_rn(sd.z, dd.z, acc); acc = __fmaf_rn(sd.w, dd.w, acc);
                }
                if (g + 8 <= sn) {          // sa/sb hold groups g..g+7
                    acc = __fmaf_rn(sa.x, da.x, acc); acc = __fmaf_rn(sa.y, da.y, acc); acc = __fmaf_rn(sa.z, da.z, acc); acc = __fmaf_rn(sa.w, da.w, acc);
                    acc = __fmaf_rn(sb.x, db.x, acc); acc = __fmaf_rn(sb.y, db.y, acc); acc = __fmaf_rn(sb.z, db.z, acc); acc = __fmaf_rn(sb.w, db.w, acc);
                    g += 8;
                }
                for (; g + 4 <= sn; g += 4) {
                    const float4 s4 = *reinterpret_cast<const float4*>(sp + g), d4 = *reinterpret_cast<const float4*>(dp + g);
                    acc = __fmaf_rn(s4.x, d4.x, acc); acc = __fmaf_rn(s4.y, d4.y, acc); acc = __fmaf_rn(s4.z, d4.z, acc); acc = __fmaf_rn(s4.w, d4.w, acc);
                }
            }
            for (; g < sn; ++g) acc = __fmaf_rn(sp[g], dp[g], acc);
        }
        // ---------------- epilogues ----------------
        if constexpr (EPI == EPI_STORE || EPI == EPI_RESIDUAL) {
            if (rv) {
                if constexpr (EPI == EPI_STORE) a.out[row] = acc;
                else a.out[row] = __fadd_rn(resid, acc);             // o.add(tmp, offset) transformer.cpp:465,493
            }
        } else if constexpr (EPI == EPI_SWIGLU) {
            // lane r < R/2 holds W1[row].x, lane r + R/2 holds W3[row].x
            const float up = __shfl(acc, (int)(lane + Rm), kWave);
            if (rv && lane < Rm) a.out[row] = swiglu_elem(acc, up);  // o1.swiglu(o3) transformer.cpp:481
        } else {   // EPI_ROPE_KV: rows (2i, 2i+1) of [Wq;Wk;Wv]; RoPE on q and k, append k,v to the cache
            const float other = __shfl_xor(acc, 1, kWave);
            if (rv && (lane & 1) == 0) {
                const float x0 = acc, x1 = other;
                const u32 hs = a.hs;
                if (row < (u32)(a.dim + a.kv_dim)) {
                    const u32 rr = row < (u32)a.dim ? row : row - a.dim;
                    const u32 h = rr / hs, d = rr - h * hs;
                    float o0, o1;
                    rope_pair(x0, x1, rc, rs, o0, o1);
                    if (row < (u32)a.dim) { a.out[row] = o0; a.out[row + 1] = o1; }
                    else { float* kp = a.kcache + ((size_t)h * a.max_seq + pos) * hs + d; kp[0] = o0; kp[1] = o1; }
                } else {
                    const u32 rr = row - a.dim - a.kv_dim;
                    const u32 h = rr / hs, d = rr - h * hs;
                    float* vp = a.vcache + ((size_t)h * a.max_seq + pos) * hs + d; vp[0] = x0; vp[1] = x1;
                }
            }
        }
    };

    Cursor rc_;                                                                // reduce cursor
    cursor_init(rc_, blockIdx.x);
    u32 it = 0;                                                                // pass counter of this workgroup
    auto do_step = [&](Set& S) {
        char* dF = lds + L.off_scr + (it & 1) * L.buf_bytes;                   // float [R+1][gstride], double buffered across passes
        char* sP = dF + (R + 1) * gstride * 4;
        reduce_step(S, rc_, dF, sP);
        const u32 pass = rc_.pass; const bool last = rc_.s + 1 == spp;
        for (int j = 0; j < H; ++j) cursor_next_block(rc_);
        cursor_end_step(rc_);
        load_step(S, lc);                                                      // refill this set: two steps ahead
        if (last) { finish_pass(pass, it, dF, sP); ++it; }
    };
    while (rc_.pass < npass) {
        do_step(setA);
        if (rc_.pass < npass) do_step(setB);
    }
}

// ------------------------------------------------------------------------------------------
// Decode attention (execute_attn at bs == 1, transformer.cpp:397-455), all fp32, one workgroup per
// head, bit-exact with the reference's order of operations:
//   att[t] = dot(K[t], q)           dot_product_avx256 (x86_simd.cpp:1447-1467): 8 strided FMA lanes, summed 0..7
//   att   *= 1/sqrt(hs)             quant::mul (quant_operators.cpp:425-428)
//   softmax                         softmax_sisd (tf_operators.cpp:176-186): max, expf, sequential sum, divide
//   o      = sum_t att[t] V[t]      batch weighted_sum (tf_operators.cpp:325-350): t ascending, FMA,
//                                   rows t >= 1 with |w| <= 1e-15 skipped
// Parallelism: phase 1 one position per thread (8 register accumulators), phase 4 one output
// dimension per thread, sequential over positions -- the chains the reference defines.
// ------------------------------------------------------------------------------------------
struct AttnArgs {
    const float* q;          // [heads*hs], RoPE already applied
    const float* kcache;     // [heads][max_seq][hs]
    const float* vcache;
    float* out;              // [heads*hs]
    const int* pos_ptr;
    int hs, max_seq;
};

constexpr int kAttnBlock = 1024;      // 16 waves: 128 positions are scored per sweep (8 lanes per position)
__global__ void __launch_bounds__(kAttnBlock) k_attn_decode(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int hs = a.hs, h = blockIdx.x;
    const int T = *a.pos_ptr + 1;
    float* qs   = reinterpret_cast<float*>(lds);                 // [hs]
    float* red  = qs + hs;                                       // 32
    float* part = red + 32;                                      // [16 waves][8 positions][8 lanes] partial dots
    float* sc   = part + 16 * 64;                                // [T] scores -> probabilities (+32 floats of slack)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* K = a.kcache + (size_t)h * a.max_seq * hs;
    const float* V = a.vcache + (size_t)h * a.max_seq * hs;
    const float scale = (float)(1.0 / (double)__builtin_sqrtf((float)hs));   // attn_scale, transformer.cpp:418

    for (int d = threadIdx.x; d < hs; d += kAttnBlock) qs[d] = a.q[(size_t)h * hs + d];
    __syncthreads();

    // ---- scores: lane = (position p = lane/8, strided accumulator k = lane%8) -- the 8 lanes of
    //      dot_product_avx256; each lane's chain is i ascending, then the 8 partials are added 0..7.
    const int p = lane >> 3, k = lane & 7;
    float* wpart = part + wave * 64;
    float lmax = -INFINITY;
    for (int tb = wave * 8; tb < T; tb += 16 * 8) {
        const int t = tb + p;
        const bool tv = t < T;
        const float* kr = K + (size_t)(tv ? t : 0) * hs + k;
        float l = 0.f;
#pragma unroll 16
        for (int i = 0; i < hs; i += 8) l = __fmaf_rn(kr[i], qs[i + k], l);
        wpart[lane] = l;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (k == 0 && tv) {
            const float4 u = *reinterpret_cast<const float4*>(wpart + lane), v = *reinterpret_cast<const float4*>(wpart + lane + 4);
            float tot = __fadd_rn(0.f, u.x);
            tot = __fadd_rn(tot, u.y); tot = __fadd_rn(tot, u.z); tot = __fadd_rn(tot, u.w);
            tot = __fadd_rn(tot, v.x); tot = __fadd_rn(tot, v.y); tot = __fadd_rn(tot, v.z); tot = __fadd_rn(tot, v.w);
            const float sv = __fmul_rn(tot, scale);             // att.multiply(attn_scale) :443
            sc[t] = sv;
            lmax = fmaxf(lmax, sv);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // block max over 16 waves (array_max is order-free)
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    for (int t = threadIdx.x; t < T; t += kAttnBlock) sc[t] = expf_ref(__fsub_rn(sc[t], m));
    __syncthreads();
    if (threadIdx.x == 0) {                                        // sum += x[i], i ascending (tf_operators.cpp:180-183)
        // a lone lane: the loop is the T dependent adds plus one LDS read per four of them, reads 28 adds ahead
        float sum = 0.f;
        int t = 0;
#define FLM_ADD4(q) sum = __fadd_rn(sum, q.x); sum = __fadd_rn(sum, q.y); sum = __fadd_rn(sum, q.z); sum = __fadd_rn(sum, q.w);
#define FLM_ASTEP(q, off) FLM_ADD4(q) q = *reinterpret_cast<const float4*>(pp + (off)); __builtin_amdgcn_sched_barrier(0);
        if (T >= 32) {
            const float* pp = sc;
            float4 q0 = *reinterpret_cast<const float4*>(pp), q1 = *reinterpret_cast<const float4*>(pp + 4), q2 = *reinterpret_cast<const float4*>(pp + 8), q3 = *reinterpret_cast<const float4*>(pp + 12);
            float4 q4 = *reinterpret_cast<const float4*>(pp + 16), q5 = *reinterpret_cast<const float4*>(pp + 20), q6 = *reinterpret_cast<const float4*>(pp + 24), q7 = *reinterpret_cast<const float4*>(pp + 28);
            __builtin_amdgcn_sched_barrier(0);
            for (; t + 32 <= T; t += 32, pp += 32) {
                FLM_ASTEP(q0, 32) FLM_ASTEP(q1, 36) FLM_ASTEP(q2, 40) FLM_ASTEP(q3, 44)
                FLM_ASTEP(q4, 48) FLM_ASTEP(q5, 52) FLM_ASTEP(q6, 56) FLM_ASTEP(q7, 60)
            }
            if (t + 4 <= T) { FLM_ADD4(q0) t += 4; } if (t + 4 <= T) { FLM_ADD4(q1) t += 4; } if (t + 4 <= T) { FLM_ADD4(q2) t += 4; } if (t + 4 <= T) { FLM_ADD4(q3) t += 4; }
            if (t + 4 <= T) { FLM_ADD4(q4) t += 4; } if (t + 4 <= T) { FLM_ADD4(q5) t += 4; } if (t + 4 <= T) { FLM_ADD4(q6) t += 4; }
        }
#undef FLM_ASTEP
#undef FLM_ADD4
        for (; t < T; ++t) sum = __fadd_rn(sum, sc[t]);
        red[16] = sum;
    }
    __syncthreads();
    const float sum = red[16];
    for (int t = threadIdx.x; t < T; t += kAttnBlock) sc[t] = __fdiv_rn(sc[t], sum);
    __syncthreads();
    // ---- o[d] = sum_t att[t] V[t][d]: one thread per output dimension, t ascending (the reference's
    //      chain); 8 V rows are loaded ahead of their 8 dependent FMAs, the weights come 4 per LDS read.
    for (int d = threadIdx.x; d < hs; d += kAttnBlock) {
        const float* vp = V + d;
        float o = __fmul_rn(vp[0], sc[0]);                         // row 0 always (tf_operators.cpp:331-336)
        int t = 1;
        for (; t < T && (t & 3); ++t) { const float w = sc[t]; o = fabsf(w) <= 1e-15f ? o : __fmaf_rn(vp[(size_t)t * hs], w, o); }
        for (; t + 8 <= T; t += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = vp[(size_t)(t + u) * hs];
            const float4 w0 = *reinterpret_cast<const float4*>(sc + t), w1 = *reinterpret_cast<const float4*>(sc + t + 4);
            const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int u = 0; u < 8; ++u) o = fabsf(w[u]) <= 1e-15f ? o : __fmaf_rn(v[u], w[u], o);   // threshold, transformer.cpp:449
        }
        for (; t < T; ++t) { const float w = sc[t]; o = fabsf(w) <= 1e-15f ? o : __fmaf_rn(vp[(size_t)t * hs], w, o); }
        a.out[(size_t)h * hs + d] = o;
    }
}
__host__ inline size_t attn_lds_bytes(int max_seq, int hs) { return (size_t)(hs + 32 + 16 * 64 + ((max_seq + 3) & ~3) + 64) * 4; }

// ------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------
// x1 = embedding[token] (copy or dequantize; transformer.cpp:115-122)
__global__ void k_embed(float* x, const void* emb, const float* emb_s, int emb_qt, int dim, const int* tok_ptr) {
    const int tok = *tok_ptr;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < dim; e += gridDim.x * blockDim.x) {
        float v;
        if (emb_qt == 0) v = reinterpret_cast<const float*>(emb)[(size_t)tok * dim + e];
        else {
            const float s = emb_s[((size_t)tok * dim + e) / kGroup];
            const int q = emb_qt == QT_INT8 ? (int)reinterpret_cast<const int8_t*>(emb)[(size_t)tok * dim + e]
                                            : (int)reinterpret_cast<const int16_t*>(emb)[(size_t)tok * dim + e];
            v = __fmul_rn((float)q, s);                              // dequantize_ quant_operators.cpp:49-65
        }
        x[e] = v;
    }
}

// sample_argmax (src/transformer/sampler.cpp:36-47): first maximum wins.  One workgroup.
// Also advances the device-resident decode state: tok <- argmax, pos <- pos+1, out[step++] <- argmax.
struct DecodeState { int pos; int tok; int step; int pad; };
__global__ void __launch_bounds__(1024) k_argmax_advance(const float* logits, int n, DecodeState* st, int* out_tokens, int advance) {
    __shared__ float bv[16]; __shared__ int bi[16];
    float best = -INFINITY; int idx = 0x7fffffff;
    // ascending index order within a thread and strict '>' keep the FIRST maximum
    const int n4 = n >> 2;
    for (int j0 = 0; j0 < n4; j0 += 8 * 1024) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = j0 + u * 1024 + threadIdx.x; v[u] = j < n4 ? reinterpret_cast<const float4*>(logits)[j] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY); }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = (j0 + u * 1024 + threadIdx.x) * 4;
            if (v[u].x > best) { best = v[u].x; idx = i; }
            if (v[u].y > best) { best = v[u].y; idx = i + 1; }
            if (v[u].z > best) { best = v[u].z; idx = i + 2; }
            if (v[u].w > best) { best = v[u].w; idx = i + 3; }
        }
    }
    for (int i = n4 * 4 + threadIdx.x; i < n; i += 1024) { const float v = logits[i]; if (v > best) { best = v; idx = i; } }
    // lower index wins ties across threads: thread-local indices are not globally ordered, so compare (value, index)
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, kWave); const int oi = __shfl_xor(idx, o, kWave);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        if (idx == 0x7fffffff) idx = 0;      // all -inf / NaN: reference returns index 0
        if (out_tokens) out_tokens[st->step] = idx;
        if (advance) { st->tok = idx; st->pos += 1; }
        st->step += 1;
    }
}
// prompt feeding: pos <- pos+1, tok <- prompt[++step]
__global__ void k_advance_prompt(DecodeState* st, const int* prompt) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { st->step += 1; st->pos += 1; st->tok = prompt[st->step]; }
}
__global__ void k_set_step(DecodeState* st, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) st->step = v; }
// x += y (tensor-parallel path: residual add after the all-reduce; Tensor::add, tensor.cpp:723-743)
__global__ void k_add_inplace(float* x, const float* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = __fadd_rn(x[i], y[i]);
}

// ---- op-level test kernels: thin launchers over the same __device__ functions ----
__global__ void k_op_swiglu(float* xo, const float* xr, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) xo[i] = swiglu_elem(xo[i], xr[i]);
}
// elementary functions as the kernels evaluate them: fn 0 expf_ref(x), 1 sqrtf(x), 2 x / y, 3 rms_scale(x, n = (int)y)
__global__ void k_op_math(int fn, float* x, const float* y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        x[i] = fn == 0 ? expf_ref(v) : fn == 1 ? __builtin_sqrtf(v) : fn == 2 ? __fdiv_rn(v, y[i]) : rms_scale(v, (int)y[i]);
    }
}
__global__ void k_op_rope(float* o, const float* x, int n_dims, const float* c, const float* s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n_dims) rope_pair(x[2 * i], x[2 * i + 1], c[i], s[i], o[2 * i], o[2 * i + 1]);
}
// softmax_sisd over n entries, one workgroup (same statements as k_attn_decode's softmax)
__global__ void __launch_bounds__(kBlock) k_op_softmax(float* x, int n) {
    __shared__ float red[16];
    float lm = -INFINITY;
    for (int i = threadIdx.x; i < n; i += kBlock) lm = fmaxf(lm, x[i]);
    const float m = block_max(lm, red);
    for (int i = threadIdx.x; i < n; i += kBlock) x[i] = expf_ref(__fsub_rn(x[i], m));
    __syncthreads();
    if (threadIdx.x == 0) { float s = 0.f; for (int i = 0; i < n; ++i) s = __fadd_rn(s, x[i]); red[8] = s; }
    __syncthreads();
    const float L = red[8];
    for (int i = threadIdx.x; i < n; i += kBlock) x[i] = __fdiv_rn(x[i], L);
}
// append one token's k (with RoPE), v to the caches and rotate q: what EPI_ROPE_KV does, for flm_op_attention
__global__ void k_op_kv_append(float* q, const float* k, const float* v, float* kc, float* vc, const float* c, const float* s,
                               int n_heads, int hs, int max_seq, int pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // pair index over heads*hs/2
    if (i >= n_heads * hs / 2) return;
    const int h = (2 * i) / hs, d = 2 * i - h * hs;
    float o0, o1;
    rope_pair(q[2 * i], q[2 * i + 1], c[d / 2], s[d / 2], o0, o1); q[2 * i] = o0; q[2 * i + 1] = o1;
    rope_pair(k[2 * i], k[2 * i + 1], c[d / 2], s[d / 2], o0, o1);
    float* kp = kc + ((size_t)h * max_seq + pos) * hs + d; kp[0] = o0; kp[1] = o1;
    float* vp = vc + ((size_t)h * max_seq + pos) * hs + d; vp[0] = v[2 * i]; vp[1] = v[2 * i + 1];
}

} // namespace flm
