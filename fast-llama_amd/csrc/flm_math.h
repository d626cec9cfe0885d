// flm_math.h -- constants and the exact scalar / wave-level building blocks shared by all kernels (part of flm_kernels.h).
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

#ifdef FLM_OPAQUE_TID
// k_layers (flm_layers.hip) runs layer_body in a LOOP over layers: the compiler then hoists every thread-index-derived value (lane offsets of five GemvCtx, the attention's
// piece offsets, poll indices ...) out of the loop and keeps them in registers across all phases -- 24 to 98 spilled VGPRs, and a kernel with scratch.  Behind this macro every
// read of threadIdx.x is an opaque value (an empty asm statement, one v_mov at most), so nothing derived from it is loop-invariant and the phases keep their own short live ranges.
struct FlmTid3 { unsigned x, y, z; };
__device__ __forceinline__ FlmTid3 flm_tid3() { unsigned t = __builtin_amdgcn_workitem_id_x(); asm volatile("" : "+v"(t)); return FlmTid3{t, 0u, 0u}; }
#define threadIdx (flm_tid3())
#endif

namespace flm {

constexpr int kWave = 64;
constexpr int kBlock = 256;           // attention / small kernels: 4 waves per workgroup
constexpr int kGemvBlock = 1024;      // GEMV: 16 waves = ONE workgroup per CU (<= 128 VGPRs): one activation prologue (and one
                                      // sequential rmsnorm chain) per CU instead of two competing for a SIMD
constexpr int kWavesPerBlock = kGemvBlock / kWave;
constexpr int kGroup = 64;            // quantization group (QUANT_GROUP_SIZE, the only value the reference uses)

// the device-resident decode state: a token is replayed from a hipGraph without host round trips
struct DecodeState { int pos; int tok; int step; int pad; };

enum { QT_INT16 = 1, QT_INT8 = 2 };
enum Prologue { PRO_NONE = 0, PRO_QUANT = 1, PRO_RMSNORM_QUANT = 2 };
enum Epilogue { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2, EPI_ROPE_KV = 3 };

template <int QT> struct QTraits;
template <> struct QTraits<QT_INT8>  { using elem = int8_t;  static constexpr int kEsz = 1; static constexpr int kEPC = 16; static constexpr float kF = 127.0f; };
template <> struct QTraits<QT_INT16> { using elem = int16_t; static constexpr int kEsz = 2; static constexpr int kEPC = 8;  static constexpr float kF = 5792.0f; };
// kEPC = elements per 16-byte chunk; lanes per quant group = 64 / kEPC (4 for int8, 8 for int16)

// ------------------------------------------------------------------------------------------
// block reductions for ORDER-FREE quantities only (max)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float row16_max(float v);
// max over the wave, in every lane: DPP only (rows, then row_bcast:15 / row_bcast:31, lane 63 read back as a scalar) -- six ds_bpermute round trips
// (__shfl_xor) cost a lone wave ~0.3 us, and the attention calls this twice on its critical path
__device__ __forceinline__ float wave_max(float v) {
    float t = row16_max(v);
    int i = __float_as_int(t);
    t = fmaxf(t, __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x142 /* row_bcast:15 */, 0xA, 0xF, false)));
    i = __float_as_int(t);
    t = fmaxf(t, __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x143 /* row_bcast:31 */, 0xC, 0xF, false)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 63));
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

// tab: kExp2fTab, or a copy of it in LDS (the lookup is on a lone wave's path in the attention: a global load there is most of a microsecond)
__device__ __forceinline__ float expf_ref(float x, const unsigned long long* tab = kExp2fTab) {
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
    const unsigned long long t = tab[ki % 32] + (ki << 47);
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
// The same element step for FOUR elements that share one scale, with the divisions' common part computed once.  x / r as the compiler lowers an IEEE fp32 division is
// div_scale (x2), v_rcp_f32, two FMAs that refine the reciprocal, a multiply, four FMAs, div_fmas, div_fixup: 11 instructions, one of them quarter-rate, per element --
// most of the quantizer round that every workgroup runs on every prologue's critical path.  When neither operand needs scaling (div_scale returns its input, div_fmas is a
// plain FMA, div_fixup passes the result through) the sequence is  y0 = rcp(r); y = fma(fma(-r, y0, 1), y0, y0);  q0 = x y; q1 = fma(fma(-r, q0, x), y, q0);
// q = fma(fma(-r, q1, x), y, q1)  -- and y depends on r only.  That holds for r in [2^-100, 2^100] (checked for the whole wave: otherwise everybody takes quant_elem) and
// |x| >= r (the quotient lies in [1, 128 F]: exponents 0 .. 13 apart); |x| < r gives a quotient below 1, i.e. 0 after the truncation, whatever its bits (and NaN -> 0 as
// v_cvt_i32_f32 does).  Precondition (the quantizer's: x belongs to the group whose maximum made r): |x| <= ~F r.  Same instructions on the same values = the same bits: tests/test_gpu_ops.py::test_quantize_shared_reciprocal_equals_ieee_division.
__device__ __forceinline__ void quant_elems4(const float4& v, float r, int (&q)[4]) {
    const unsigned re = __float_as_uint(r) & 0x7f800000u;
    const bool safe = re >= (27u << 23) && re <= (227u << 23);
    if (__ballot(!safe) != 0ull) { q[0] = quant_elem(v.x, r); q[1] = quant_elem(v.y, r); q[2] = quant_elem(v.z, r); q[3] = quant_elem(v.w, r); return; }
    const float y0 = __builtin_amdgcn_rcpf(r);
    const float y = __fmaf_rn(__fmaf_rn(-r, y0, 1.0f), y0, y0);
    const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = xs[i];
        const float q0 = __fmul_rn(x, y);
        const float q1 = __fmaf_rn(__fmaf_rn(-r, q0, x), y, q0);
        const float t = __fmaf_rn(__fmaf_rn(-r, q1, x), y, q1);
        q[i] = fabsf(x) >= r ? (int)t : 0;
    }
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

// Activations and KV-cache entries cross workgroups (and XCDs, whose L2s are not coherent with each other) INSIDE the
// persistent kernel.  Every such access is a relaxed agent-scope atomic: stores write through to memory (sc1), loads
// are served coherently (sc1) -- so a grid barrier needs no L2 write-back / invalidate, only 'my stores have completed'.
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// "My global stores have COMPLETED" (written through to where other workgroups read them): the vector-memory counter at zero.
// A workgroup-scope release fence is NOT that -- the compiler emits only s_waitcnt lgkmcnt(0) for it (one CU, one L1: nothing
// to wait for inside a workgroup), and a flag raised by another wave of the workgroup could then overtake the data.
__device__ __forceinline__ void wait_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
constexpr int kAuxCoherent = 17;      // raw buffer load cache policy: sc0 | sc1 (gfx940+ encoding of the aux operand)

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

} // namespace flm
