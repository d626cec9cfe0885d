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

// Activations and KV-cache entries cross workgroups (and XCDs, whose L2s are not coherent with each other) INSIDE the
// persistent kernel.  Every such access is a relaxed agent-scope atomic: stores write through to memory (sc1), loads
// are served coherently (sc1) -- so a grid barrier needs no L2 write-back / invalidate, only 'my stores have completed'.
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
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

// ------------------------------------------------------------------------------------------
// GEMV argument block
// ------------------------------------------------------------------------------------------
struct GemvArgs {
    // weights: row-major [rows][n] quantized values + natural-layout scales [rows][n/64].
    // EPI_SWIGLU: W = [W1 (gate) ; W3 (up)], both [items][n], stored back to back (values and scales alike)
    const void*  W;   const float* sW;
    int n;                                      // K (columns), multiple of 64
    int items;                                  // rows (STORE/RESIDUAL), hidden (SWIGLU), row pairs (ROPE_KV)
    int rows_per_pass;                          // Rm: rows (of each matrix) one workgroup reduces per pass; multiple of RB, <= 64
    int cb_shift;                               // log2(CB): a 1 KiB wave load covers RB = (64 >> cb_shift) rows x CB 16-byte chunks
    int ctr_off;                                // LDS byte offset of the two step counters; 0: the layout's own (k_token keeps them at a fixed place across phases)
    int nbuf;                                   // strip buffers: 2, or 1 when each workgroup has a single pass and LDS is short
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
    unsigned long long* trace;                  // FLM_ABLATE builds: per-workgroup timeline [grid][8] (s_memtime), else unused
    int ablate;                                 // perf exploration only (results invalid when != 0): 1 no group chain, 2 no rmsnorm chain, 4 no weight loads, 8 no dots, 16 return immediately, 32 return after prologue
};

#ifndef FLM_ABLATE
#define FLM_ABLATE 0          // build with -DFLM_ABLATE=1 to compile the perf-exploration switches of GemvArgs::ablate into the hot loop
#endif
constexpr bool kAblate = FLM_ABLATE != 0;
constexpr int kStepBlk = 4;            // H: 1 KiB wave loads per step; two steps (register sets) in flight: 8 KiB/wave, 128 KiB/CU

// LDS layout: [xq : n*esz] [xs : n/64 floats, padded to 16 B] [red : 16 floats] [scratch]
// scratch = max( rmsnorm transpose staging 4n bytes ,
//                2 buffers x (Rm + RB) strips; strip r = { float(group dot), sW*sX } pairs of row r, groups ascending
//                (SWIGLU: entries { d(W1), d(W3), s(W1), s(W3) }: the two chains are the halves of one packed FMA) )
struct GemvLds {
    int off_xs, off_red, off_ctr, off_scr;     // byte offsets (off_ctr: the two step counters of GemvCtx)
    int gstride;                      // BYTES per strip: 16 x odd, so that 16 lanes reading 16 B each from 16 strips hit all banks
    int buf_bytes;                    // one strip buffer
    int total;                        // bytes
};
__host__ __device__ inline GemvLds gemv_lds_layout(int n, int esz, bool norm, int Rm, int RB, bool two, int nbuf = 2) {
    GemvLds L;
    const int sn = n / kGroup, ng = two ? 2 * sn : sn;
    L.off_xs = n * esz;
    L.off_red = L.off_xs + ((sn * 4 + 15) & ~15);
    L.off_ctr = L.off_red + 64;
    L.off_scr = L.off_ctr + 16;
    int g16 = (ng * 8 + 15) / 16; if ((g16 & 1) == 0) ++g16;
    L.gstride = g16 * 16;
    L.buf_bytes = (Rm + RB) * L.gstride;                                       // + RB dummy strips that absorb the writes of padding blocks
    int scratch = nbuf * L.buf_bytes + 64;                                     // + 64: the chain's read-ahead past the last strip
    if (norm && n * 4 + 512 > scratch) scratch = n * 4 + 512;                   // 4 strips of n/4 + 8 floats (+8: the 4 chain lanes read different banks); + the chain's read-ahead past the last strip
    L.total = L.off_scr + scratch;
    return L;
}
// One of the 4 strided lanes of simd::square_sum -> square_sum_avx128 (x86_simd.cpp:942-960; the AVX2 branch is
// dead, :1093): p[0..n4) = x[c], x[c+4], x[c+8], ... walked as a strictly sequential FMA chain.
__device__ __forceinline__ float sq_chain(const float* p, int n4) {
    float l = 0.f;
    int k = 0;
#define FLM_SQ4(v) l = __fmaf_rn(v.x, v.x, l); l = __fmaf_rn(v.y, v.y, l); l = __fmaf_rn(v.z, v.z, l); l = __fmaf_rn(v.w, v.w, l);
#define FLM_RD4(a, b, c, d, base) a = *reinterpret_cast<const float4*>(pp + (base)); b = *reinterpret_cast<const float4*>(pp + (base) + 4); c = *reinterpret_cast<const float4*>(pp + (base) + 8); d = *reinterpret_cast<const float4*>(pp + (base) + 12);
    if (n4 >= 32) {
        // A lone wave issues roughly one instruction every ~5 cycles, whatever its kind, so the loop body must be
        // little more than the dependent FMAs: two rings of 4 float4 registers; while the 16 FMAs of one ring run,
        // the 4 LDS reads of the other are in flight, and ONE explicit s_waitcnt per 16 FMAs (instead of the
        // compiler's one per read) covers them.  Reads run up to 32 floats past a lane's strip: the staging area
        // is sized for that (gemv_lds_layout) and those values are never consumed.
        const float* pp = p;
        float4 a0, a1, a2, a3, b0, b1, b2, b3;
        FLM_RD4(a0, a1, a2, a3, 0)
#pragma unroll 2
        for (; k + 32 <= n4; k += 32, pp += 32) {
            FLM_RD4(b0, b1, b2, b3, 16)
            __builtin_amdgcn_s_waitcnt(0xC47F);        // lgkmcnt <= 4: ring A has landed (vmcnt / expcnt untouched)
            __builtin_amdgcn_sched_barrier(0);
            FLM_SQ4(a0) FLM_SQ4(a1) FLM_SQ4(a2) FLM_SQ4(a3)
            __builtin_amdgcn_sched_barrier(0);
            FLM_RD4(a0, a1, a2, a3, 32)
            __builtin_amdgcn_s_waitcnt(0xC47F);        // ring B has landed
            __builtin_amdgcn_sched_barrier(0);
            FLM_SQ4(b0) FLM_SQ4(b1) FLM_SQ4(b2) FLM_SQ4(b3)
            __builtin_amdgcn_sched_barrier(0);
        }
        // ring A holds p[k .. k+15]
        if (k + 4 <= n4) { FLM_SQ4(a0) k += 4; } if (k + 4 <= n4) { FLM_SQ4(a1) k += 4; } if (k + 4 <= n4) { FLM_SQ4(a2) k += 4; } if (k + 4 <= n4) { FLM_SQ4(a3) k += 4; }
    }
#undef FLM_RD4
#undef FLM_SQ4
    for (; k < n4; ++k) l = __fmaf_rn(p[k], p[k], l);
    return l;
}

// The same chain -- l <- fma(x_k, x_k, l), k ascending, bit for bit -- evaluated by a WHOLE WAVE in far fewer than n
// dependent steps.  The terms are non-negative, so l only grows, and while l stays inside one binade [2^E, 2^(E+1)) every
// step rounds l + x^2 to a multiple of u = ulp(l).  Within the binade the increment t_k = fl(l + x_k^2) - l does not depend
// on l (except for exact ties): it is what ONE fma against the bottom of the binade gives, t_k = fma(x_k, x_k, 2^E) - 2^E,
// a multiple of u, and sums of such multiples below 2^(E+1) are exact in fp32 in ANY order -- a prefix sum.  Lane L takes
// elements 4L..4L+3 of a 256-element block, a wave scan adds them up.  Two kinds of element stop the scan: one on which l
// leaves the binade (fma(x, x, l_before) >= 2^(E+1): the rounding unit changes) and one whose x^2 lies exactly halfway
// between two multiples of u (round-half-even then looks at the parity of l: detected as |fma(x, x, -t)| == u/2).  The scan
// commits everything before the first such element, that element takes one real fma, and the scan resumes behind it with the
// new binade.  l doubles only ~log2(n) times over a chain, mostly within the first elements, which are simply run in order.
// STATUS: exact (tests/test_gpu_ops.py::test_square_sum_wave_parallel_is_bit_exact, adversarial ties / overflow / denormals)
// but NOT used by the product path: a lone wave pays ~7 cycles per instruction whatever it does, this formulation runs
// ~150 instructions per scan round and needs 4 rounds + one per binade change (8-9 for n/4 = 1024), i.e. about as many
// instructions as the 1024 dependent FMAs and their LDS reads (measured 7.1 us against 4.3 us in the prologue).  It pays
// only below ~75 instructions per round; kept, tested, for the round that hand-schedules it.
__device__ __forceinline__ float wave_scan_incl(float v) {
#define FLM_SCAN_STEP(ctrl, rmask, bc) v = __fadd_rn(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, bc)));
    FLM_SCAN_STEP(0x111 /* row_shr:1 */, 0xF, true) FLM_SCAN_STEP(0x112 /* row_shr:2 */, 0xF, true)
    FLM_SCAN_STEP(0x114 /* row_shr:4 */, 0xF, true) FLM_SCAN_STEP(0x118 /* row_shr:8 */, 0xF, true)
    FLM_SCAN_STEP(0x142 /* row_bcast:15 */, 0xA, false) FLM_SCAN_STEP(0x143 /* row_bcast:31 */, 0xC, false)
#undef FLM_SCAN_STEP
    return v;
}
__device__ __forceinline__ float sq_chain_wave(const float* p, int n4, int* iters = nullptr) {
    const int lane = threadIdx.x & 63;
    int n_it = 0;
    constexpr int kHead = 64;                                      // elements run in order first (l crosses most binades here)
    float acc = 0.f;
    int k = 0;
    for (; k + 4 <= n4 && k < kHead; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + k);
        acc = __fmaf_rn(v.x, v.x, acc); acc = __fmaf_rn(v.y, v.y, acc); acc = __fmaf_rn(v.z, v.z, acc); acc = __fmaf_rn(v.w, v.w, acc);
    }
    for (int base = 0; base < n4; base += 256) {
        const int e0 = base + 4 * lane;
        float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
        if (e0 + 4 <= n4) { const float4 v = *reinterpret_cast<const float4*>(p + e0); x0 = v.x; x1 = v.y; x2 = v.z; x3 = v.w; }
        else { if (e0 < n4) x0 = p[e0]; if (e0 + 1 < n4) x1 = p[e0 + 1]; if (e0 + 2 < n4) x2 = p[e0 + 2]; }
        int done = k > base ? k - base : 0;                                                // elements of this block already consumed (uniform)
        const int limit = (n4 - base) < 256 ? (n4 - base) : 256;
        auto pick = [&](int i) { return i == 0 ? x0 : i == 1 ? x1 : i == 2 ? x2 : x3; };
        while (done < limit) {
            ++n_it;
            const unsigned ab = __builtin_amdgcn_readfirstlane(__float_as_uint(acc));
            const unsigned eb = ab & 0x7f800000u;
            if (eb < (32u << 23) || eb >= (254u << 23)) {
                // l is zero / tiny / not finite: no usable binade -- one plain step, then look again
                const float xs = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(pick(done & 3)), done >> 2));
                acc = __fmaf_rn(xs, xs, acc);
                ++done;
                continue;
            }
            const float aref = __uint_as_float(eb), top = __fadd_rn(aref, aref), half_u = __uint_as_float(eb - (24u << 23));
            const int el = 4 * lane;
            const bool v0 = el >= done, v1 = el + 1 >= done, v2 = el + 2 >= done, v3 = el + 3 >= done;
            // increments (multiples of u) and exact ties
            const float t0 = v0 ? __fsub_rn(__fmaf_rn(x0, x0, aref), aref) : 0.f, t1 = v1 ? __fsub_rn(__fmaf_rn(x1, x1, aref), aref) : 0.f;
            const float t2 = v2 ? __fsub_rn(__fmaf_rn(x2, x2, aref), aref) : 0.f, t3 = v3 ? __fsub_rn(__fmaf_rn(x3, x3, aref), aref) : 0.f;
            bool s0 = v0 && fabsf(__fmaf_rn(x0, x0, -t0)) == half_u, s1 = v1 && fabsf(__fmaf_rn(x1, x1, -t1)) == half_u;
            bool s2 = v2 && fabsf(__fmaf_rn(x2, x2, -t2)) == half_u, s3 = v3 && fabsf(__fmaf_rn(x3, x3, -t3)) == half_u;
            const float c0 = t0, c1 = __fadd_rn(c0, t1), c2 = __fadd_rn(c1, t2), c3 = __fadd_rn(c2, t3);
            const float incl = wave_scan_incl(c3);
            // l in front of this lane's first element: the inclusive sum of the lane BELOW (incl - c3 would not do: the lane of a
            // binade-leaving element holds a huge increment, incl is rounded there, and the difference is off by an ulp)
            const float lb = __fadd_rn(acc, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(incl), 0x138 /* wave_shr:1 */, 0xF, 0xF, true)));
            // does l leave the binade on this element?  (exact test: the real step from the value in front of it)
            s0 = s0 || (v0 && __fmaf_rn(x0, x0, lb) >= top);
            s1 = s1 || (v1 && __fmaf_rn(x1, x1, __fadd_rn(lb, c0)) >= top);
            s2 = s2 || (v2 && __fmaf_rn(x2, x2, __fadd_rn(lb, c1)) >= top);
            s3 = s3 || (v3 && __fmaf_rn(x3, x3, __fadd_rn(lb, c2)) >= top);
            const int fi = s0 ? 0 : s1 ? 1 : s2 ? 2 : s3 ? 3 : 4;
            const float before = __fadd_rn(lb, s0 ? 0.f : s1 ? c0 : s2 ? c1 : c2);          // l in front of the lane's first special element
            const unsigned long long sm = __ballot(fi < 4);
            if (sm) {
                const int Ls = __ffsll((long long)sm) - 1;
                const int fs = __builtin_amdgcn_readlane(fi, Ls);
                acc = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(before), Ls));
                done = 4 * Ls + fs;
                if (done < limit) {
                    const float xs = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(pick(fs)), Ls));
                    acc = __fmaf_rn(xs, xs, acc);                                           // the special element: one real step
                    ++done;
                }
            } else {
                acc = __fadd_rn(acc, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(incl), 63)));
                done = limit;
            }
        }
        k = base + 256;
    }
    if (iters) *iters = n_it;
    return acc;
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
// COH: the activation was written by other workgroups of the SAME kernel (k_attn_o) -> coherent sc0|sc1 loads
template <int QT, int PRO, int XR, bool COH = false>
__device__ __forceinline__ void gemv_preload(const GemvArgs& a, float4 (&xv)[XR > 0 ? XR : 1], float4 (&wv)[XR > 0 ? XR : 1]) {
    if constexpr (PRO == PRO_QUANT || PRO == PRO_RMSNORM_QUANT) {
        // branch-free: raw buffer loads, elements past n read as zero
        typedef float v4f __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.n * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRO == PRO_RMSNORM_QUANT ? a.norm_w : a.x), 0, a.n * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int off = (threadIdx.x * 4 + i * kGemvBlock * 4) * 4;
            const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, COH ? kAuxCoherent : 0));
            xv[i] = make_float4(v.x, v.y, v.z, v.w);
            if constexpr (PRO == PRO_RMSNORM_QUANT) {
                const v4f u = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rn, off, 0, 0));
                wv[i] = make_float4(u.x, u.y, u.z, u.w);
            }
        }
    }
}

#ifdef FLM_TRACE_PRO
#define FLM_PRO_STAMP(k) if (kAblate && a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define FLM_PRO_STAMP(k)
#endif
template <int QT, int PRO, int XR, class AfterStage>
__device__ __forceinline__ void gemv_prologue(const GemvArgs& a, char* lds, float4 (&xv)[XR > 0 ? XR : 1], float4 (&wv)[XR > 0 ? XR : 1], AfterStage&& after_stage) {
    using T = QTraits<QT>;
    const int n = a.n;
    const int tid = threadIdx.x;
    const GemvLds L = gemv_lds_layout(n, T::kEsz, PRO == PRO_RMSNORM_QUANT, a.rows_per_pass, 64 >> a.cb_shift, false);   // only the fixed offsets are used here
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
        if constexpr (PRO == PRO_QUANT) {
            // no staging here: the hook (the weight prefetch) runs as soon as this thread's activation registers have landed
            if constexpr (XR > 0) { asm volatile("" :: "v"(xv[XR - 1].w)); }
            FLM_PRO_STAMP(3)
            after_stage(0);
            FLM_PRO_STAMP(4)
        }
        if constexpr (PRO == PRO_RMSNORM_QUANT) {
            // simd::square_sum -> square_sum_avx128 (x86_simd.cpp:942-960; the AVX2 branch is dead, :1093):
            // lane c of 4 accumulates x[c], x[c+4], x[c+8]... by FMA, then res = ((0+l0)+l1)+l2)+l3.
            // Stage x transposed ([4][n/4]) so that 4 threads can each walk one strided lane sequentially.
            const int n4 = n / 4, ns = n4 + 8;                  // strip stride: +8 floats so that the 4 chain lanes' 16-byte reads hit different banks
            auto stage = [&](int i, const float4& v) {
                const int k = tid + i * kGemvBlock;
                if (k < n4) { scratch[k] = v.x; scratch[ns + k] = v.y; scratch[2 * ns + k] = v.z; scratch[3 * ns + k] = v.w; }
            };
#pragma unroll
            for (int i = 0; i < XR; ++i) { if (i < rounds) stage(i, xv[i]); }
            for (int i = XR; i < rounds; ++i) {
                const int e = tid * 4 + i * kGemvBlock * 4;
                if (e < n) stage(i, *reinterpret_cast<const float4*>(a.x + e));
            }
            __syncthreads();
            FLM_PRO_STAMP(3)
            // the hook issues the weight prefetch.  Wave 0 goes first (the others give it ~128 cycles): its 16 loads
            // enter an empty memory pipeline at once and it is free for the chain; queued behind the other 15 waves'
            // 240 loads it would stall for ~1 us before (or after) the chain.
            if (tid >= kWave) __builtin_amdgcn_s_sleep(2);
            after_stage(0);
            if (tid < 4 && !(a.ablate & 2)) red[8 + tid] = sq_chain(scratch + tid * ns, n4);
            FLM_PRO_STAMP(4)
            __syncthreads();
            const float ss = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, red[8]), red[9]), red[10]), red[11]);
            r = rms_scale(ss, n);
            __syncthreads();                                   // scratch is reused by the GEMV waves below
            FLM_PRO_STAMP(5)
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
        if constexpr (PRO == PRO_QUANT) {
            FLM_PRO_STAMP(5)
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
// One 16-wave workgroup per CU reduces Rm rows per pass.  The Rm x K tile is cut into 1 KiB blocks of
// (RB rows x CB chunks of 16 B), RB*CB = 64, CB = the largest power of two dividing K/16 (so a block is
// one fully coalesced buffer_load_dwordx4 per wave and every lane is busy for any K).  The waves form a
// WC x WR grid: wave (wc, wr) owns the column blocks wc, wc+WC, ... and, of those, the row blocks
// wr, wr+WR, ...; it walks them column block by column block, so that from one block to the next
// only three scalar offsets advance by constants (the instruction stream per KiB is what limits a
// GEMV whose operands arrive at several TB/s), and its activation chunk stays in registers.
// SWIGLU runs [W1 ; W3] as ONE matrix with twice the column blocks: both dot products of a row land in
// the same strip and the same chain lane.  Per block:
//   1. int32 dot per 16-byte chunk (v_dot4 / v_dot2), exact;
//   2. DPP sum over the 4 (int8) / 8 (int16) lanes of a quant group -> the group's int32 dot, exact;
//   3. group leaders park { float(dot), sW*sX } in the row's LDS strip (one ds_write_b64);
// and per pass, after ONE workgroup barrier, one wave walks the strips, lane r = row r:
//        acc = fma(s[g], d[g], acc), g ascending -- the reference's summation order, bit-identical --
// amortising the sequential fp32 chain over Rm rows, and runs the epilogue.  Strips are double
// buffered, so the other waves are already in the next pass.  Two register sets of H blocks each
// keep 8 KiB per wave (128 KiB per CU) of weight loads in flight at all times; the first 8 are
// issued before the prologue so HBM latency and the sequential rmsnorm chain overlap the stream.
//
// GemvCtx is the per-wave state of one GEMV: geometry, the load cursor and the two register sets.
// The standalone kernel k_gemv and the persistent whole-token kernel k_token both drive it:
//     init -> issue (weight loads of the first two steps) -> [activation prologue] -> run
// ------------------------------------------------------------------------------------------
typedef unsigned int u32;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int QT, int EPI>
struct GemvCtx {
    using T = QTraits<QT>;
    static constexpr u32 LPGS = (T::kEPC == 16) ? 2 : 3;                      // log2(lanes per quant group): 4 | 8 lanes
    static constexpr u32 LPG = 1u << LPGS;
    static constexpr bool TWO = EPI == EPI_SWIGLU;
    static constexpr int H = kStepBlk;
    static constexpr u32 ES = TWO ? 16 : 8;                                    // bytes per strip entry: {d, s}, SWIGLU {d1, d3, s1, s3}
    static constexpr u32 kOOB = 0x80000000u;

    // A STEP is H consecutive row blocks of one column block of one pass: H weight loads of 1 KiB per wave plus ONE scale
    // load (lane (j, g) fetches the scale of block j's g-th quant group -- 64 / LPG groups per block, so for int8 the
    // step's 64 scales fill the wave exactly; a separate scale load per block cost as much of the CU's address
    // pipeline as the weight load itself).  The steps of a workgroup -- its passes in order, inside a pass row chunk by
    // row chunk, a chunk's column blocks next to each other -- are numbered, and handed out through a counter in LDS: a
    // wave takes the next number whenever it refills a register set.  With a fixed wave grid the waves that the CU's
    // memory pipeline serves last (it is a FIFO: wave 15's requests queue behind everybody else's every round) ended
    // 3 us after the first ones, on a 12 us main loop; the wave that runs a pass's chain falls behind as well.
    // Numbered steps cost two scalar multiply-high's to decode, and the activation chunk is re-read from LDS when the
    // column block changes (one ds_read_b128 per step at most).
    struct Set { v4i w[H]; float sw; u32 itl, st, xo, nlive; };               // itl: workgroup-local pass index (np_wg: no work left)

    // geometry (wave-uniform unless noted)
    u32 n, lane, wave, rowbytes, sn, cbs, RB, nbc, NBCV, TRm, Rm, RBP, SP, NS, np_wg, npass, gstride, buf_bytes, off_xs, off_scr, ctr_off;
    u32 inv_SP, inv_NBCV;                                                      // ceil(2^32 / d): exact quotients for the step numbers that occur (< 2^16)
    u32 lane_woff, lane_xoff, lane_goff;                                       // per lane: weight chunk, activation chunk, strip entry (dot)
    u32 lane_j, lane_s2off, lane_sx2off, lane_poff;                            // per lane, scale role: block of the step, its scale, the activation scale, strip entry (s)
    bool leader;                                                               // per lane
    u32 dW, dS, dT, dummy_st, wg, nwg, nbuf;
    __amdgpu_buffer_rsrc_t rW, rS;
    Set setA, setB;
    bool stored;                                                               // this wave wrote results to global memory

    static __device__ __forceinline__ u32 inv_of(u32 d) { return d > 1 ? 0xFFFFFFFFu / d + 1u : 0u; }
    static __device__ __forceinline__ u32 udiv(u32 x, u32 d, u32 inv) { return d > 1 ? __umulhi(x, inv) : x; }

    // ctr_slot: which of the two step counters in LDS this GEMV uses (k_token alternates them from phase to phase: a
    // fast wave initialises the next phase while slow ones still draw from this phase's counter)
    __device__ __forceinline__ void init(const GemvArgs& a, u32 wg_, u32 nwg_, char* lds, u32 ctr_slot = 0) {
        n = a.n; wg = wg_; nwg = nwg_;
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        rowbytes = n * T::kEsz; sn = n / kGroup;
        const u32 nchunks = rowbytes / 16;
        cbs = a.cb_shift; RB = 64u >> cbs;
        const u32 CB = 1u << cbs;
        nbc = nchunks >> cbs;                                                  // column blocks per row
        NBCV = TWO ? 2 * nbc : nbc;                                            // ... of the (virtual) matrix this launch walks
        TRm = (u32)a.items * (EPI == EPI_ROPE_KV ? 2u : 1u);                   // rows per matrix
        Rm = a.rows_per_pass;
        RBP = Rm / RB;                                                         // row blocks per pass
        npass = (TRm + Rm - 1) / Rm;
        np_wg = wg < npass ? (npass - wg + nwg - 1) / nwg : 0;                 // passes of this workgroup
        SP = ((RBP + H - 1) / H) * NBCV;                                       // steps per pass
        NS = np_wg * SP;
        inv_SP = inv_of(SP); inv_NBCV = inv_of(NBCV);
        nbuf = a.nbuf > 0 ? a.nbuf : 2;
        const GemvLds L = gemv_lds_layout(n, T::kEsz, true, Rm, RB, TWO, nbuf);
        gstride = L.gstride; buf_bytes = L.buf_bytes; off_xs = L.off_xs; off_scr = L.off_scr; ctr_off = a.ctr_off ? (u32)a.ctr_off : (u32)L.off_ctr + 4 * (ctr_slot & 1);
        dW = RB * rowbytes; dS = RB * sn * 4; dT = RB * gstride;               // row block to row block
        dummy_st = Rm * gstride;
        // lane-constant parts of every address (the per-step parts are wave-uniform scalars)
        const u32 rb = lane >> cbs, cb = lane & (CB - 1);
        lane_woff = rb * rowbytes + cb * 16;                                   // weights, bytes from the block base
        lane_xoff = cb * 16;                                                   // activation chunk in LDS
        lane_goff = rb * gstride + (cb >> LPGS) * ES;                          // strip entry of this lane's group
        leader = (cb & (LPG - 1)) == 0;
        // scale role: lane -> (block j of the step, quant group g of the block); g's leader lane is g * LPG
        constexpr u32 GPB = 64u / LPG;
        lane_j = lane / GPB;                                                   // >= H: no scale role (int16: lanes 32..63)
        const u32 ll = (lane % GPB) * LPG, rb2 = ll >> cbs, cb2 = ll & (CB - 1);
        lane_s2off = lane_j * dS + (rb2 * sn + (cb2 >> LPGS)) * 4;
        lane_sx2off = (cb2 >> LPGS) * 4;
        lane_poff = lane_j * dT + rb2 * gstride + (cb2 >> LPGS) * ES + ES / 2;
        // Weight and scale blocks are fetched with raw buffer loads whose whole offset sits in the VGPR operand (lane
        // constant + the step's scalar): that operand is what the hardware bounds-checks, so padding blocks, rows past
        // the end of the matrix and steps past the end of the work (offset kOOB) return zero without touching memory.
        // "nt": each weight byte is read once per token.
        constexpr int kRsrcFlags = 0x00020000;                                 // raw buffer, 32-bit data format (gfx9 family)
        const u32 NM = TWO ? 2u : 1u;
        rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, (int)(NM * TRm * rowbytes), kRsrcFlags);
        rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.sW), 0, (int)(NM * TRm * sn * 4), kRsrcFlags);
        stored = false; primedA = primedB = false;
        // the first two steps of every wave are fixed (wave, wave + 16): they are requested before any barrier
        if (threadIdx.x == 0) *reinterpret_cast<u32*>(lds + ctr_off) = 2 * kWavesPerBlock;
    }

    // step number -> the set's bookkeeping and the scalar offsets of its first block
    __device__ __forceinline__ void decode(u32 s, Set& S, u32& wo, u32& so) const {
        if (s >= NS) { S.itl = np_wg; S.st = 0; S.xo = 0; S.nlive = 0; wo = kOOB; so = kOOB; return; }
        const u32 itl = udiv(s, SP, inv_SP), rem = s - itl * SP;
        const u32 q = udiv(rem, NBCV, inv_NBCV), cv = rem - q * NBCV;
        const bool second = TWO && cv >= nbc;
        const u32 cc = second ? cv - nbc : cv, g0 = (cc << cbs) >> LPGS;       // column block inside its matrix, its first quant group
        const u32 rb0 = q * H, row0 = (second ? TRm : 0u) + (wg + itl * nwg) * Rm + rb0 * RB;
        wo = row0 * rowbytes + ((cc << cbs) * 16);
        so = (row0 * sn + g0) * 4;
        S.itl = itl; S.xo = cc; S.nlive = RBP - rb0 < (u32)H ? RBP - rb0 : (u32)H;
        S.st = rb0 * RB * gstride + g0 * ES + (second ? 4u : 0u);
    }
    __device__ __forceinline__ void load_step(Set& S, u32 s, int ablate) const {
        u32 wo, so;
        decode(s, S, wo, so);
        u32 nl = S.nlive;
        if (kAblate && (ablate & 4)) { nl = 0; so = kOOB; }
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const u32 woj = (u32)j < nl ? wo + j * dW : kOOB;
            S.w[j] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rW, (int)(lane_woff + woj), 0, 2));
        }
        const u32 svo = lane_j < nl ? lane_s2off + so : kOOB;
        S.sw = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rS, (int)svo, 0, 2));
    }
    // the first two steps of weight loads: independent of the activation
    // (part 1 / 2: only the first / second set; a set that was never requested is drawn at the start of run())
    bool primedA, primedB;
    __device__ __forceinline__ void issue(int ablate, int part = 0) {
        if (part != 2) { load_step(setA, wave, ablate); primedA = true; }
        if (part != 1) { load_step(setB, wave + kWavesPerBlock, ablate); primedB = true; }
    }

    // reduce one step: the group dots of its H blocks (registers), then the leaders park them; the scale-role lanes park s = sW * sX
    v4i xa; float sx2; u32 cur_xo;                                             // activation chunk (dot role) / activation scale (scale role) of the current column block
    __device__ __forceinline__ void reduce_step(const Set& S, char* lds, char* strips, int ablate) {
        const char* xq = lds; const char* xs = lds + off_xs;
        if (S.xo != cur_xo) {                                                  // wave-uniform: a new column block
            cur_xo = S.xo;
            xa = *reinterpret_cast<const v4i*>(xq + ((cur_xo << cbs) * 16) + lane_xoff);
            sx2 = *reinterpret_cast<const float*>(xs + (((cur_xo << cbs) >> LPGS) * 4) + lane_sx2off);
        }
        float d[H];
#pragma unroll
        for (int j = 0; j < H; ++j) {
            int t = (kAblate && (ablate & 8)) ? 0 : quad_sum(dot_chunk<QT>(S.w[j], xa));
            if constexpr (LPG == 8) t += __builtin_amdgcn_update_dpp(0, t, 0x104 /* row_shl:4 */, 0xF, 0xF, true);
            d[j] = (float)t;                                                   // exact int32 -> fp32, as "s * dot" does
        }
        if (leader) {
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const u32 stj = (u32)j < S.nlive ? S.st + j * dT : dummy_st;   // padding blocks hold zeros: parked in the dummy strips
                *reinterpret_cast<float*>(strips + stj + lane_goff) = d[j];
            }
        }
        if (lane_j < S.nlive) *reinterpret_cast<float*>(strips + S.st + lane_poff) = __fmul_rn(S.sw, sx2);   // s = sW * sX (quant_operators.cpp:274)
    }

    // the end of a pass: one barrier, then ONE wave runs the fp32 chains of all Rm rows and the epilogue
    __device__ __forceinline__ void finish_pass(const GemvArgs& a, u32 pass, u32 it, const char* strips, int pos) {
        const bool chain_wave = wave == (it & (kWavesPerBlock - 1));
        // epilogue operands of the chain wave, fetched before the barrier (lane r = row r of the pass)
        float resid = 0.f, rc = 0.f, rs = 0.f;
        const u32 row = pass * Rm + lane;                                      // row inside its matrix
        const bool rv = chain_wave && lane < Rm && row < TRm;
        if constexpr (EPI == EPI_RESIDUAL) { if (rv) resid = ld_agent(a.out + row); }
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
#ifdef FLM_TRACE_BAR
        if (kAblate && a.trace && threadIdx.x == 0 && it == 0) a.trace[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memtime();
#endif
        stored = true;
        // ---- the reference's fp32 chain, lane r = row r: o[j] += s * dot (FMA), groups ascending.
        //      A lone wave issues an instruction every ~5-7 cycles whatever its kind, so the loop is little more than the
        //      dependent FMAs: strip entries are read 4 at a time into two register rings, one ring's reads fly while the
        //      other ring's FMAs run, ONE explicit s_waitcnt per ring.  SWIGLU: an entry is {d1, d3, s1, s3}, and the W1
        //      and W3 chains are the two halves of one v_pk_fma_f32 (each half an IEEE fma), operands in place.
        float acc = 0.f, acc2 = 0.f;
        if (lane < Rm && !(a.ablate & 1)) {
            const char* sp = strips + lane * gstride;
            u32 g = 0;
#define FLM_RD4(r0, r1, r2, r3, ptr) r0 = *reinterpret_cast<const float4*>(ptr); r1 = *reinterpret_cast<const float4*>((ptr) + 16); r2 = *reinterpret_cast<const float4*>((ptr) + 32); r3 = *reinterpret_cast<const float4*>((ptr) + 48);
            if constexpr (TWO) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 ac = {0.f, 0.f};
#define FLM_CH(q) ac = __builtin_elementwise_fma(f2{q.z, q.w}, f2{q.x, q.y}, ac);
                if (sn >= 8) {
                    float4 a0, a1, a2, a3, b0, b1, b2, b3;
                    FLM_RD4(a0, a1, a2, a3, sp)
                    for (; g + 8 <= sn; g += 8) {
                        const char* pn = sp + (g + 4) * 16;
                        FLM_RD4(b0, b1, b2, b3, pn)
                        __builtin_amdgcn_s_waitcnt(0xC47F);        // lgkmcnt <= 4: ring A has landed
                        __builtin_amdgcn_sched_barrier(0);
                        FLM_CH(a0) FLM_CH(a1) FLM_CH(a2) FLM_CH(a3)
                        __builtin_amdgcn_sched_barrier(0);
                        FLM_RD4(a0, a1, a2, a3, pn + 64)           // (past the end on the last round: inside the allocation, never consumed)
                        __builtin_amdgcn_s_waitcnt(0xC47F);        // ring B has landed
                        __builtin_amdgcn_sched_barrier(0);
                        FLM_CH(b0) FLM_CH(b1) FLM_CH(b2) FLM_CH(b3)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // ring A holds groups g .. g+3
                    if (g < sn) { FLM_CH(a0) ++g; } if (g < sn) { FLM_CH(a1) ++g; } if (g < sn) { FLM_CH(a2) ++g; } if (g < sn) { FLM_CH(a3) ++g; }
                }
                for (; g < sn; ++g) { const float4 e = *reinterpret_cast<const float4*>(sp + g * 16); FLM_CH(e) }
#undef FLM_CH
                acc = ac.x; acc2 = ac.y;
            } else {
#define FLM_CH(q) acc = __fmaf_rn(q.y, q.x, acc); acc = __fmaf_rn(q.w, q.z, acc);
                if (sn >= 16) {
                    float4 a0, a1, a2, a3, b0, b1, b2, b3;
                    FLM_RD4(a0, a1, a2, a3, sp)
                    for (; g + 16 <= sn; g += 16) {
                        const char* pn = sp + (g + 8) * 8;
                        FLM_RD4(b0, b1, b2, b3, pn)
                        __builtin_amdgcn_s_waitcnt(0xC47F);
                        __builtin_amdgcn_sched_barrier(0);
                        FLM_CH(a0) FLM_CH(a1) FLM_CH(a2) FLM_CH(a3)
                        __builtin_amdgcn_sched_barrier(0);
                        FLM_RD4(a0, a1, a2, a3, pn + 64)
                        __builtin_amdgcn_s_waitcnt(0xC47F);
                        __builtin_amdgcn_sched_barrier(0);
                        FLM_CH(b0) FLM_CH(b1) FLM_CH(b2) FLM_CH(b3)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // ring A holds groups g .. g+7; a pair is consumed only when both of its groups exist
                    if (g + 2 <= sn) { FLM_CH(a0) g += 2; } if (g + 2 <= sn) { FLM_CH(a1) g += 2; } if (g + 2 <= sn) { FLM_CH(a2) g += 2; } if (g + 2 <= sn) { FLM_CH(a3) g += 2; }
                }
#undef FLM_CH
                for (; g < sn; ++g) { const float2 e = *reinterpret_cast<const float2*>(sp + g * 8); acc = __fmaf_rn(e.y, e.x, acc); }
            }
#undef FLM_RD4
        }
        // ---------------- epilogues ----------------
        if constexpr (EPI == EPI_STORE || EPI == EPI_RESIDUAL) {
            if (rv) {
                if constexpr (EPI == EPI_STORE) st_agent(a.out + row, acc);
                else st_agent(a.out + row, __fadd_rn(resid, acc));   // o.add(tmp, offset) transformer.cpp:465,493
            }
        } else if constexpr (EPI == EPI_SWIGLU) {
            if (rv) st_agent(a.out + row, swiglu_elem(acc, acc2));   // o1.swiglu(o3) transformer.cpp:481
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
                    if (row < (u32)a.dim) { st_agent(a.out + row, o0); st_agent(a.out + row + 1, o1); }
                    else { float* kp = a.kcache + ((size_t)h * a.max_seq + pos) * hs + d; st_agent(kp, o0); st_agent(kp + 1, o1); }
                } else {
                    const u32 rr = row - a.dim - a.kv_dim;
                    const u32 h = rr / hs, d = rr - h * hs;
                    float* vp = a.vcache + ((size_t)h * a.max_seq + pos) * hs + d; st_agent(vp, x0); st_agent(vp + 1, x1);
                }
            }
        }
    }

    // the main loop: the quantized activation is in LDS (xq at 0, xs at off_xs); issue() has run
    template <class Stamp>
    __device__ __forceinline__ void run(const GemvArgs& a, char* lds, Stamp&& stamp) {
        int pos = 0;
        if constexpr (EPI == EPI_ROPE_KV) pos = *a.pos_ptr;
        xa = v4i{0, 0, 0, 0}; sx2 = 0.f; cur_xo = 0xffffffffu;
        u32* ctr = reinterpret_cast<u32*>(lds + ctr_off);
        u32 it = 0;                                                            // pass of this workgroup this wave is in
        bool tr3 = false;
        // a wave's step numbers only grow, so when a set belongs to a later pass every earlier pass is complete for this wave
        auto do_set = [&](Set& S) -> bool {
            while (it < S.itl) {
                char* strips = lds + off_scr + (nbuf > 1 ? (it & 1) * buf_bytes : 0u);   // double buffered across passes
#ifdef FLM_TRACE_WAVES
                if (kAblate && a.trace && it == 0 && lane == 0 && wave % 3 == 0) a.trace[blockIdx.x * 8 + 1 + wave / 3] = __builtin_amdgcn_s_memtime();
#elif !defined(FLM_TRACE_PRO)
                if (it == 0) stamp(4);
#endif
                finish_pass(a, wg + it * nwg, it, strips, pos);
#ifndef FLM_TRACE_PRO
                if (it == 0) stamp(5);
#endif
                ++it;
            }
            if (S.itl >= np_wg) return false;                                  // no work left (every later number is past the end too)
            reduce_step(S, lds, lds + off_scr + (nbuf > 1 ? (S.itl & 1) * buf_bytes : 0u), a.ablate);
            u32 s = 0;
            if (lane == 0) s = atomicAdd(ctr, 1u);
            load_step(S, __builtin_amdgcn_readfirstlane(s), a.ablate);         // refill this set: a full cycle ahead
#if !defined(FLM_TRACE_PRO) && !defined(FLM_TRACE_BAR)
            if (!tr3) { tr3 = true; stamp(3); }
#endif
            return true;
        };
        if (!primedA) load_step(setA, wave, a.ablate);                         // (a wave that was busy elsewhere during the prologue)
        if (!primedB) load_step(setB, wave + kWavesPerBlock, a.ablate);
        while (do_set(setA) && do_set(setB)) {}
    }
};

template <int QT, int PRO, int EPI, int XR>
__global__ void __launch_bounds__(kGemvBlock, 4) k_gemv(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    unsigned long long rt0 = 0;
#ifdef FLM_TRACE_WAVES
    auto stamp = [&](int k) { if (kAblate && a.trace && threadIdx.x == 0 && k == 0) a.trace[blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime(); };
#else
    auto stamp = [&](int k) { if (kAblate && a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime(); };
#endif
    if (kAblate && a.trace && threadIdx.x == 0) rt0 = __builtin_amdgcn_s_memrealtime();
    stamp(0);
    if (a.ablate & 16) return;
    // The activation first, and the weight prefetch only once it HAS ARRIVED (the hook runs after the staging barrier /
    // after the activation registers landed).  Weights do not depend on the activation and were once requested up
    // front -- but workgroups start ~1 us apart, and the activation loads of the late ones then queued in HBM behind
    // 32 MB of weight requests of the early ones: the activation came back 2.6 us later (measured), delaying the
    // whole rmsnorm chain.  Issued after the activation, the first 128 KiB per CU still arrive under the chain.
    float4 xv[XR > 0 ? XR : 1], nv[XR > 0 ? XR : 1];
    gemv_preload<QT, PRO, XR>(a, xv, nv);
    GemvCtx<QT, EPI> g;
    g.init(a, blockIdx.x, gridDim.x, lds);
    if constexpr (PRO == PRO_NONE) g.issue(a.ablate);
    stamp(1);
    gemv_prologue<QT, PRO, XR>(a, lds, xv, nv, [&](int part) { g.issue(a.ablate, part); });
    stamp(2);
    if (a.ablate & 32) return;
    g.run(a, lds, stamp);
    stamp(6);
    if (kAblate && a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime() - rt0;
}

// ------------------------------------------------------------------------------------------
// Decode attention (execute_attn at bs == 1, transformer.cpp:397-455), all fp32, one workgroup per
// head, bit-exact with the reference's order of operations:
//   att[t] = dot(K[t], q)           dot_product_avx256 (x86_simd.cpp:1447-1467): 8 strided FMA lanes, summed 0..7
//   att   *= 1/sqrt(hs)             quant::mul (quant_operators.cpp:425-428)
//   softmax                         softmax_sisd (tf_operators.cpp:176-186): max, expf, sequential sum, divide
//   o      = sum_t att[t] V[t]      batch weighted_sum (tf_operators.cpp:325-350): t ascending, FMA,
//                                   rows t >= 1 with |w| <= 1e-15 skipped
// The chains are the reference's; what is engineered is LATENCY (this kernel sits on the token's critical path):
// K and V stream through LDS in tiles of 64 positions -- every thread fetches coalesced 16-byte pieces of the NEXT
// tile while the current one is consumed, so a tile costs compute time, not a memory round trip -- and the first
// K tile, the first V tile and q are all requested at once when the kernel starts.  Scores: lane = (position,
// one of the 8 strided accumulators), operands from LDS (row stride hs+8 floats: conflict-free), the 8 partials
// are added in order with DPP shifts.  PV: one thread per output dimension walks the tile's positions in order.
// ------------------------------------------------------------------------------------------
struct AttnArgs {
    const float* q;          // [heads*hs], RoPE already applied
    const float* kcache;     // [heads][max_seq][hs]
    const float* vcache;
    float* out;              // [heads*hs]
    const int* pos_ptr;
    int hs, max_seq;
    unsigned long long* trace;   // FLM_ABLATE builds: [head][8] s_memtime stamps
};

constexpr int kAttnBlock = 1024;      // 16 waves
constexpr int kAttnTile = 64;         // positions per LDS tile
constexpr int kAttnDepth = 2;         // tiles in flight per stream (K, V), register rings: both streams start when the kernel does
// LDS row stride of a tile in floats: compile-time per instantiation (64*NF + 8), so that the chains' LDS reads use
// immediate offsets; +8: the 8x8 (position, accumulator) score lanes and the PV lanes hit distinct banks
__host__ __device__ inline int attn_row_stride(int hs) { const int nf = hs <= 64 ? 1 : hs <= 128 ? 2 : 4; return nf * 64 + 8; }
__host__ inline size_t attn_lds_bytes(int max_seq, int hs) { return (size_t)(hs + 32 + ((max_seq + 3) & ~3) + 64 + (2 * kAttnTile + 4) * attn_row_stride(hs)) * 4; }   // + 4 slack rows: the PV read-ahead

// NF = 16-byte pieces of a tile per thread = ceil(hs / 64)
// COH: K/V/q were (partly) written by other workgroups of the SAME kernel (k_token) -> coherent sc0|sc1 loads; the
// per-phase kernels read them after a kernel boundary and use ordinary cached loads
template <int NF, bool COH>
__device__ __forceinline__ void attn_head(const AttnArgs& a, const int h, char* lds, const int T, const float* qrow, float* orow) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    constexpr int D = kAttnDepth;
    const int hs = a.hs, tid = threadIdx.x;
    auto stamp = [&](int k) { if (kAblate && a.trace && threadIdx.x == 0) a.trace[h * 8 + k] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    constexpr int rs = NF * 64 + 8;
    const int f4r = hs >> 2, tile_f4 = kAttnTile * f4r;
    float* qs   = reinterpret_cast<float*>(lds);                 // [hs]
    float* red  = qs + hs;                                       // 32
    float* sc   = red + 32;                                      // [T] scores -> probabilities (+ slack for the sum ring's read-ahead)
    float* tile0 = sc + ((a.max_seq + 3) & ~3) + 64;
    float* tile1 = tile0 + kAttnTile * rs;
    const int lane = tid & 63, wave = tid >> 6;
    const float* K = a.kcache + (size_t)h * a.max_seq * hs;
    const float* V = a.vcache + (size_t)h * a.max_seq * hs;
    // K/V rows of this token may have been written by other workgroups of the same kernel (k_token): coherent loads
    // (sc0|sc1) through buffer descriptors; positions past T get an out-of-range offset and read as zero
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K), 0, a.max_seq * hs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V), 0, a.max_seq * hs * 4, 0x00020000);
    const float scale = (float)(1.0 / (double)__builtin_sqrtf((float)hs));   // attn_scale, transformer.cpp:418
    const int nt = (T + kAttnTile - 1) / kAttnTile;
    // Two streams of tiles (K for the scores, V for the weighted sum), each through a ring of D register sets; both are
    // requested when the kernel starts (the V tiles arrive under the softmax), tile i+D when tile i has been parked.
    v4f ringK[D][NF], ringV[D][NF];
    // this thread's pieces of a tile: row / byte offsets computed once (an integer division per piece per tile would
    // cost more than the tile's arithmetic)
    int prow[NF], goff[NF], loff[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int f = tid + j * kAttnBlock, row = f / f4r, c4 = f - row * f4r;
        prow[j] = f < tile_f4 ? row : (1 << 28);                    // pieces past the tile never pass the t < T test
        goff[j] = (row * hs + c4 * 4) * 4;
        loff[j] = row * rs + c4 * 4;
    }
    const int tile_bytes = kAttnTile * hs * 4;
    auto request = [&](const __amdgpu_buffer_rsrc_t& r, int tile, v4f (&reg)[NF]) {
        const int t0 = tile * kAttnTile;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const unsigned off = (tile < nt && t0 + prow[j] < T) ? (unsigned)(tile * tile_bytes + goff[j]) : 0x80000000u;
            reg[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, COH ? kAuxCoherent : 0));
        }
    };
    auto park = [&](float* buf, const v4f (&reg)[NF]) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
            if (prow[j] < kAttnTile) *reinterpret_cast<float4*>(buf + loff[j]) = make_float4(reg[j].x, reg[j].y, reg[j].z, reg[j].w);
    };
#pragma unroll
    for (int u = 0; u < D; ++u) request(rK, u, ringK[u]);
#pragma unroll
    for (int u = 0; u < D; ++u) request(rV, u, ringV[u]);
    for (int d = tid; d < hs; d += kAttnBlock) qs[d] = COH ? ld_agent(qrow + (size_t)h * hs + d) : qrow[(size_t)h * hs + d];

    // ---- scores: lane = (position p, strided accumulator k) -- the 8 lanes of dot_product_avx256; each lane's
    //      chain is i ascending, then the 8 partials are added 0..7 (lane k = 0 collects them with DPP row shifts).
    float lmax = -INFINITY;
    for (int base = 0; base < nt; base += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int s = base + u;
            if (s >= nt) break;                                     // (uniform)
            float* cur = (u & 1) ? tile1 : tile0;                   // D is even: tile parity == slot parity
            park(cur, ringK[u]);
            __syncthreads();
            request(rK, s + D, ringK[u]);
            if (tid < kAttnTile * 8) {
                const int p = tid >> 3, k = tid & 7, t = s * kAttnTile + p;
                const float* kp = cur + p * rs + k;
                float l = 0.f;
#pragma unroll 16
                for (int j = 0; j < hs; j += 8) l = __fmaf_rn(kp[j], qs[j + k], l);
                const int li = __float_as_int(l);
                float tot = __fadd_rn(0.f, l);
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x101 /* row_shl:1 */, 0xF, 0xF, true)));
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x102, 0xF, 0xF, true)));
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x103, 0xF, 0xF, true)));
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x104, 0xF, 0xF, true)));
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x105, 0xF, 0xF, true)));
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x106, 0xF, 0xF, true)));
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x107, 0xF, 0xF, true)));
                if (k == 0 && t < T) {
                    const float sv = __fmul_rn(tot, scale);         // att.multiply(attn_scale) :443
                    sc[t] = sv;
                    lmax = fmaxf(lmax, sv);
                }
            }
        }
    }
    stamp(1);
    // block max over 16 waves (array_max is order-free)
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    for (int t = tid; t < T; t += kAttnBlock) sc[t] = expf_ref(__fsub_rn(sc[t], m));
    __syncthreads();
    stamp(2);
    if (tid == 0) {                                                // sum += x[i], i ascending (tf_operators.cpp:180-183)
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
    stamp(3);
    const float sum = red[16];
    // att[t] = exp / sum; rows t >= 1 with |att| <= 1e-15 are skipped by the weighted sum (transformer.cpp:449): they are
    // stored as exact zeros so that the PV chain can tell them apart with one wave-uniform test per four positions
    for (int t = tid; t < T; t += kAttnBlock) { const float w = __fdiv_rn(sc[t], sum); sc[t] = (t > 0 && fabsf(w) <= 1e-15f) ? 0.f : w; }
    // (the first barrier of the loop below orders these writes before the PV reads)
    // ---- o[d] = sum_t att[t] V[t][d]: one thread per output dimension, t ascending (the reference's chain) over
    //      the LDS tiles.
    float o = 0.f;
    for (int base = 0; base < nt; base += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int i = base + u;
            if (i >= nt) break;                                     // (uniform)
            float* cur = (u & 1) ? tile1 : tile0;
            park(cur, ringV[u]);
            __syncthreads();
            request(rV, i + D, ringV[u]);
            if (tid < hs) {
                const float* vp = cur + tid;
                const float* wp = sc + i * kAttnTile;
                const int np = (T - i * kAttnTile) < kAttnTile ? (T - i * kAttnTile) : kAttnTile;
                // weights are wave-uniform (one per position): lane p looks at weight p once per tile; if no row of the
                // tile is skipped, the walk is nothing but LDS reads at immediate offsets and dependent FMAs
                const float wl = lane < np ? wp[lane] : 1.f;
                const bool dense = __all(wl != 0.f) != 0;
                int p = 0;
                if (i == 0) { o = __fmul_rn(vp[0], wp[0]); p = 1; }  // row 0 always (tf_operators.cpp:331-336)
                if (dense) {
                    for (; p < np && (p & 7); ++p) o = __fmaf_rn(vp[p * rs], wp[p], o);
                    if (p + 8 <= np) {
                        const float* vq = vp + p * rs; const float* wq = wp + p;
                        float4 wa = *reinterpret_cast<const float4*>(wq), wb = *reinterpret_cast<const float4*>(wq + 4);
                        float a0 = vq[0], a1 = vq[rs], a2 = vq[2 * rs], a3 = vq[3 * rs], a4 = vq[4 * rs], a5 = vq[5 * rs], a6 = vq[6 * rs], a7 = vq[7 * rs];
                        for (; p + 16 <= np; p += 8) {
                            vq += 8 * rs; wq += 8;
                            const float4 wc = *reinterpret_cast<const float4*>(wq), wd = *reinterpret_cast<const float4*>(wq + 4);
                            const float b0 = vq[0], b1 = vq[rs], b2 = vq[2 * rs], b3 = vq[3 * rs], b4 = vq[4 * rs], b5 = vq[5 * rs], b6 = vq[6 * rs], b7 = vq[7 * rs];
                            o = __fmaf_rn(a0, wa.x, o); o = __fmaf_rn(a1, wa.y, o); o = __fmaf_rn(a2, wa.z, o); o = __fmaf_rn(a3, wa.w, o);
                            o = __fmaf_rn(a4, wb.x, o); o = __fmaf_rn(a5, wb.y, o); o = __fmaf_rn(a6, wb.z, o); o = __fmaf_rn(a7, wb.w, o);
                            wa = wc; wb = wd; a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
                        }
                        o = __fmaf_rn(a0, wa.x, o); o = __fmaf_rn(a1, wa.y, o); o = __fmaf_rn(a2, wa.z, o); o = __fmaf_rn(a3, wa.w, o);
                        o = __fmaf_rn(a4, wb.x, o); o = __fmaf_rn(a5, wb.y, o); o = __fmaf_rn(a6, wb.z, o); o = __fmaf_rn(a7, wb.w, o);
                        p += 8;
                    }
                    for (; p < np; ++p) o = __fmaf_rn(vp[p * rs], wp[p], o);
                } else {
                    // some row is skipped (weight stored as exact 0, threshold of transformer.cpp:449): it leaves o untouched
                    for (; p < np; ++p) { const float w = wp[p]; o = w == 0.f ? o : __fmaf_rn(vp[p * rs], w, o); }
                }
            }
        }
    }
    stamp(4);
    if (tid < hs) st_agent(orow + (size_t)h * hs + tid, o);
    __syncthreads();                                                // the LDS is free for whoever runs next on it (k_token)
}
template <bool COH>
__device__ __forceinline__ void attn_head_any(const AttnArgs& a, const int h, char* lds, const int T, const float* qrow, float* orow) {
    if (a.hs <= 64) attn_head<1, COH>(a, h, lds, T, qrow, orow); else if (a.hs <= 128) attn_head<2, COH>(a, h, lds, T, qrow, orow); else attn_head<4, COH>(a, h, lds, T, qrow, orow);
}
// batched prefill: workgroup (h, i) is query i of the batch, at position pos0 + i, over the cache rows 0 .. pos0 + i
__global__ void __launch_bounds__(kAttnBlock) k_attn_prefill(const AttnArgs a, int pos0, int row_stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int i = blockIdx.y;
    attn_head_any<false>(a, blockIdx.x, lds, pos0 + i + 1, a.q + (size_t)i * row_stride, a.out + (size_t)i * row_stride);
}
__global__ void __launch_bounds__(kAttnBlock) k_attn_decode(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    attn_head_any<false>(a, blockIdx.x, lds, *a.pos_ptr + 1, a.q, a.out);
}

// ------------------------------------------------------------------------------------------
// Attention and the output projection in ONE launch (single GPU): workgroups [0, n_heads) run one attention head each,
// the others run the Wo GEMV.  The GEMV workgroups request their first steps of Wo -- with 16.8 MB over ~224 CUs that is
// every block they will ever need -- the moment the kernel starts, and only then wait for the heads: the weight fetch,
// ~3.5 us of the stand-alone attn_o kernel, and one kernel boundary (1.6 us) disappear behind the attention.
// The heads publish their output with write-through stores (st_agent) and, once those have completed, each writes `target`
// (layer + 1) into its own 64-byte flag line; in every GEMV workgroup lane i polls head i's line (the pattern of
// grid_barrier: a shared counter cost 2.7 us from the last head's bump to the last poll's success), then the activation
// is read with coherent loads.  All workgroups are resident (grid <= CUs, one 1024-thread workgroup per CU); a poll that
// never succeeds gives up after ~20 ms and raises *err.  The flag lines are zero when the token starts (k_embed).
constexpr int kFlagStride = 16;      // dwords
template <int QT, int XR>
__global__ void __launch_bounds__(kGemvBlock, 4) k_attn_o(const AttnArgs aa, const GemvArgs a, const int n_heads, unsigned* flag, const unsigned target, int* err) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    auto stamp = [&](int k) { if (kAblate && a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime(); };   // tools/trace_ao.py
    stamp(0);
    if ((int)blockIdx.x < n_heads) {
        attn_head_any<false>(aa, blockIdx.x, lds, *aa.pos_ptr + 1, aa.q, aa.out);
        stamp(1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                  // this wave's stores have completed
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag + blockIdx.x * kFlagStride, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp(4);
        return;
    }
    GemvCtx<QT, EPI_RESIDUAL> g;
    g.init(a, blockIdx.x - n_heads, gridDim.x - n_heads, lds);
    g.issue(a.ablate);
    stamp(1);
    if ((int)(threadIdx.x & ~63u) < n_heads) {                              // the waves that own at least one head's flag: lane i polls head i's line
        const bool mine = (int)threadIdx.x < n_heads;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (true) {
            const unsigned f = mine ? __hip_atomic_load(flag + threadIdx.x * kFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
            if (__all(f >= target)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
    stamp(2);
    float4 xv[XR > 0 ? XR : 1], nv[XR > 0 ? XR : 1];
    gemv_preload<QT, PRO_QUANT, XR, true>(a, xv, nv);
    gemv_prologue<QT, PRO_QUANT, XR>(a, lds, xv, nv, [](int) {});
    stamp(3);
    g.run(a, lds, [](int) {});
    stamp(4);
}

// ------------------------------------------------------------------------------------------
// The persistent whole-token kernel (single GPU).  One 16-wave workgroup per CU walks the token's
// phases  L x { qkv, attention, attn_o, ffn13, ffn2 }, cls  with a grid barrier between phases instead of
// a kernel boundary, so that
//   * the weight stream does not drain at every phase change: each wave issues the first 8 KiB of the NEXT
//     phase's weights BEFORE it arrives at the barrier (weights never depend on activations), and
//   * there is no launch / drain / argument-fetch latency per phase.
// Loads return in issue order, so a wave with weight loads in flight would see the new activation only
// after them: waves 0..3 ("activation waves", one per SIMD) therefore postpone their weight prefetch
// until they have issued the activation loads right after the barrier.
// Every workgroup must be resident at once (grid <= CUs, one workgroup per CU by LDS footprint); a barrier
// that does not complete within seconds sets *err and lets the kernel run to its end instead of hanging.
// Results are the same bits as the per-phase kernels: same device functions, same chains.
// ------------------------------------------------------------------------------------------
struct TokenArgs {
    const GemvArgs* gemv;        // device array: per layer { qkv, attn_o, ffn13, ffn2 }, then { cls }
    const AttnArgs* attn;        // device array: per layer
    int n_layers, n_heads, with_cls;
    unsigned* bar;               // grid barrier counter, zero at kernel start (k_embed resets it)
    int* err;
    unsigned long long* trace;   // FLM_ABLATE builds: [workgroup][phase (<= 15)][8] s_memtime stamps of the first phases
};

// the argument tables are written by the host before the launch; every workgroup reads the same entry.  Each dword
// goes through readfirstlane so that the compiler knows it is wave-uniform (SGPRs): a buffer descriptor built from
// a value it believes divergent would be wrapped in a waterfall loop.
template <class A> __device__ __forceinline__ A kload(const A* p) {
    static_assert(sizeof(A) % 4 == 0, "dword-sized argument blocks");
    constexpr int N = sizeof(A) / 4;
    union { A a; unsigned u[N]; } r;
    const unsigned* s = reinterpret_cast<const unsigned*>(p);
#pragma unroll
    for (int i = 0; i < N; ++i) r.u[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)s[i]);
    return r.a;
}

constexpr int kActWaves = 4;     // activation waves
constexpr int kNormRounds = 2;   // rmsnorm phases: n <= 2 * 4096 (host falls back to the per-phase kernels otherwise)

// Grid barrier.  All global data that crosses workgroups is written with st_agent (write-through), so "release" is
// just: each wave has waited for its own stores (vmcnt) BEFORE it queued the next phase's weight loads (token_phase
// does that; waiting here would wait for the prefetch too).
// Measured on MI355X, 256 workgroups (tools/ubench/barrier.hip): one atomic counter 3.7 us; 16 group counters + root
// 2.5 us; flags packed in 1 KiB 3.2 us (write-through stores to a shared line serialise); ONE 64-BYTE LINE PER
// WORKGROUP, polled by one lane each: 1.4 us -- less than a kernel boundary.  So: workgroup i publishes
// flag[i] = epoch (one write-through store to its own line); lane j of the first waves polls flag[j] coherently
// until it reaches the epoch.  No read-modify-write, no shared line.
// t.bar: [grid] flags 64 bytes apart, zeroed by k_embed at the start of the token.
__device__ __forceinline__ void grid_barrier(const TokenArgs& t, unsigned& epoch) {
    __syncthreads();
    epoch += 1;
    const unsigned nwg = gridDim.x;
    if (threadIdx.x == 0) __hip_atomic_store(t.bar + blockIdx.x * kFlagStride, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((threadIdx.x & ~63u) < nwg) {                                     // the waves that own at least one flag
        const bool mine = threadIdx.x < nwg;
        unsigned spins = 0;
        while (true) {
            const unsigned f = mine ? __hip_atomic_load(t.bar + threadIdx.x * kFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
            if (__all(f >= epoch)) break;
            // a workgroup that never arrives (not resident) must not hang the GPU: give up after ~1 s, flag it, and let
            // every later barrier of this token fall through at once
            if ((++spins & 255u) == 0 && (spins > (1u << 20) || __hip_atomic_load(t.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_store(t.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// activation prologue of k_token: x (complete in global memory since the barrier) -> xq / xs in LDS
template <int QT, int PRO, int EPI>
__device__ __forceinline__ void mega_prologue(const GemvArgs& a, char* lds, GemvCtx<QT, EPI>& g) {
    using T = QTraits<QT>;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int n = a.n, tid = threadIdx.x, n4 = n / 4, ns = n4 + 8;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const GemvLds L = gemv_lds_layout(n, T::kEsz, true, a.rows_per_pass, 64 >> a.cb_shift, false);   // fixed offsets only
    char*  xq = lds;
    float* xs = reinterpret_cast<float*>(lds + L.off_xs);
    float* red = reinterpret_cast<float*>(lds + L.off_red);
    float* scratch = reinterpret_cast<float*>(lds + L.off_scr);
    constexpr int kXChunk = 12;                                               // float4 loads per lane and chunk (256 lanes: 48 KiB)
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRO == PRO_RMSNORM_QUANT ? a.norm_w : a.x), 0, n * 4, 0x00020000);
    float4 nw[kNormRounds];
    auto load_nw = [&]() {
        if constexpr (PRO == PRO_RMSNORM_QUANT) {
#pragma unroll
            for (int i = 0; i < kNormRounds; ++i) {
                const v4f u = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rn, (tid * 4 + i * kGemvBlock * 4) * 4, 0, 0));
                nw[i] = make_float4(u.x, u.y, u.z, u.w);
            }
        }
    };
    if (wave < kActWaves) {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, n * 4, 0x00020000);
        for (int base = 0; base < n; base += kXChunk * kActWaves * 64 * 4) {
            v4f v[kXChunk];
#pragma unroll
            for (int j = 0; j < kXChunk; ++j) v[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, (base + j * kActWaves * 64 * 4 + tid * 4) * 4, 0, kAuxCoherent));
            if (base == 0) load_nw();
#pragma unroll
            for (int j = 0; j < kXChunk; ++j) {
                const int e = base + j * kActWaves * 64 * 4 + tid * 4;
                if (e < n) {
                    if constexpr (PRO == PRO_RMSNORM_QUANT) { const int k = e >> 2; scratch[k] = v[j].x; scratch[ns + k] = v[j].y; scratch[2 * ns + k] = v[j].z; scratch[3 * ns + k] = v[j].w; }
                    else *reinterpret_cast<float4*>(scratch + e) = make_float4(v[j].x, v[j].y, v[j].z, v[j].w);
                }
            }
            if (base == 0) g.issue(a.ablate);                             // the postponed weight prefetch: behind the activation in the return order
        }
    } else load_nw();
    __syncthreads();
    float r = 1.0f;
    if constexpr (PRO == PRO_RMSNORM_QUANT) {
        if (tid < 4 && !(a.ablate & 2)) red[8 + tid] = sq_chain(scratch + tid * ns, n4);
        __syncthreads();
        const float ss = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, red[8]), red[9]), red[10]), red[11]);
        r = rms_scale(ss, n);
    }
    const int rounds = (n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
    for (int i = 0; i < rounds; ++i) {
        const int e = tid * 4 + i * kGemvBlock * 4;
        const bool act = e < n;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) {
            if constexpr (PRO == PRO_RMSNORM_QUANT) { const int k = e >> 2; v = make_float4(scratch[k], scratch[ns + k], scratch[2 * ns + k], scratch[3 * ns + k]); }
            else v = *reinterpret_cast<const float4*>(scratch + e);
        }
        if constexpr (PRO == PRO_RMSNORM_QUANT) {
            const float4 w = i == 0 ? nw[0] : nw[kNormRounds - 1];
            // multiply_avx256 (x86_simd.cpp:1360-1372): (x*w)*r
            v.x = __fmul_rn(__fmul_rn(v.x, w.x), r); v.y = __fmul_rn(__fmul_rn(v.y, w.y), r);
            v.z = __fmul_rn(__fmul_rn(v.z, w.z), r); v.w = __fmul_rn(__fmul_rn(v.w, w.w), r);
        }
        // group max over the 16 lanes that share this 64-element group (order-free, exact)
        const float mx = row16_max(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        const float sc = __fdiv_rn(mx, T::kF);           // scale = max|x| / F
        if (act) {
            const int q0 = quant_elem(v.x, sc), q1 = quant_elem(v.y, sc), q2 = quant_elem(v.z, sc), q3 = quant_elem(v.w, sc);
            if constexpr (QT == QT_INT8) {
                *reinterpret_cast<uint32_t*>(xq + e) = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
            } else {
                uint2 pk;
                pk.x = (uint32_t)(q0 & 0xffff) | ((uint32_t)(q1 & 0xffff) << 16);
                pk.y = (uint32_t)(q2 & 0xffff) | ((uint32_t)(q3 & 0xffff) << 16);
                *reinterpret_cast<uint2*>(xq + (size_t)e * 2) = pk;
            }
            if ((tid & 15) == 0) xs[e / kGroup] = sc;
        }
    }
    __syncthreads();
}

// One GEMV phase of k_token: [release my stores] -> prefetch the phase's first weights -> grid barrier ->
// activation prologue -> GEMV.  ATTN_O runs the attention heads between two barriers first.
// Deliberately NOT inlined: one register allocation per phase keeps the prefetched weight sets in
// registers (inlined into one body, the allocator spills them across the neighbouring phases).
struct TokenState { unsigned epoch; int stored; int phase; };

template <int QT, int PRO, int EPI, bool ATTN>
__device__ __attribute__((noinline)) void token_phase(const TokenArgs& t, const GemvArgs* ap, const AttnArgs* aap, char* lds, TokenState& ts, const int barrier) {
    const u32 wg = blockIdx.x, nwg = gridDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto nostamp = [](int) {};
    GemvCtx<QT, EPI> g;
    auto stamp = [&](int k) { if (kAblate && t.trace && threadIdx.x == 0 && ts.phase < 16) t.trace[((size_t)blockIdx.x * 16 + ts.phase) * 8 + k] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    const GemvArgs a = kload(ap);
    unsigned epoch = ts.epoch;
    // the phase's first weight loads; activation waves wait until they have asked for the activation
    auto prefetch = [&]() { g.init(a, wg, nwg, lds); if (wave >= kActWaves) g.issue(a.ablate); };
    // my stores of the previous phase must have completed before the prefetch is queued behind them (one vmcnt counter)
    if (ts.stored) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if constexpr (ATTN) {
        const bool attn_wg = (int)wg < t.n_heads;                               // this workgroup runs attention heads next
        if (!attn_wg) prefetch();                                               // (a head's K/V loads must not queue behind weight loads)
        grid_barrier(t, epoch);
        if (attn_wg) {   // one head per workgroup
            const AttnArgs aa = kload(aap);
            for (int h = wg; h < t.n_heads; h += nwg) attn_head_any<true>(aa, h, lds, *aa.pos_ptr + 1, aa.q, aa.out);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            prefetch();
        }
        stamp(1);
        grid_barrier(t, epoch);
    } else {
        prefetch();
        stamp(1);
        if (barrier) grid_barrier(t, epoch);
    }
    stamp(2);
    mega_prologue<QT, PRO, EPI>(a, lds, g);
    stamp(3);
    g.run(a, lds, nostamp);
    stamp(4);
    ts.epoch = epoch; ts.stored = g.stored ? 1 : 0; ts.phase += 1;
}

template <int QT>
__global__ void __launch_bounds__(kGemvBlock, 4) k_token(const TokenArgs t) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    TokenState ts{0u, 0, 0};
    for (int l = 0; l < t.n_layers; ++l) {
        const GemvArgs* ga = t.gemv + 4 * l;
        token_phase<QT, PRO_RMSNORM_QUANT, EPI_ROPE_KV, false>(t, ga + 0, nullptr, lds, ts, l > 0);   // layer 0: the residual stream comes from k_embed
        token_phase<QT, PRO_QUANT, EPI_RESIDUAL, true>(t, ga + 1, t.attn + l, lds, ts, 1);            // attention, ATTN_O + residual
        token_phase<QT, PRO_RMSNORM_QUANT, EPI_SWIGLU, false>(t, ga + 2, nullptr, lds, ts, 1);        // FFN13 + SwiGLU
        token_phase<QT, PRO_QUANT, EPI_RESIDUAL, false>(t, ga + 3, nullptr, lds, ts, 1);              // FFN2 + residual
    }
    if (t.with_cls) token_phase<QT, PRO_RMSNORM_QUANT, EPI_STORE, false>(t, t.gemv + 4 * t.n_layers, nullptr, lds, ts, 1);   // final norm + classifier
}

// ------------------------------------------------------------------------------------------
// Batched prefill (ParallelTransformer::forward with bs > 1, transformer.cpp:105-161).  A prompt's tokens before the
// last one only have to leave their K/V rows in the cache; every (token, row) value is produced by the SAME chain as in
// the single-token kernels (group dots exact, acc = fma(sW*sX, float(dot), acc) with groups ascending; per-row rmsnorm
// chains; per-query attention), so the cache -- and therefore the logits of the last token, which runs through the
// decode kernels -- is bit-identical to feeding the prompt token by token, at a fraction of the time: the weights are
// streamed once per 64 tokens instead of once per token.
//   k_embed_rows        x[b] = embedding[token b]
//   k_rows_prologue     per token row: (rmsnorm,) quantize -> xq[b], xs[b]   (the decode prologue, one workgroup per row)
//   k_gemm_q            out[b][r] (+)= W[r] . xq[b] for a 64 x 64 (rows x tokens) tile per workgroup
//   k_rope_kv_rows      RoPE on q and k of every token, K/V rows appended to the cache
//   k_attn_prefill      causal attention: one workgroup per (head, query), the decode attention with T = pos + i + 1
//   k_swiglu_rows       hd[b] = swiglu(gate[b], up[b])
// ------------------------------------------------------------------------------------------
__global__ void k_embed_rows(float* x, const void* emb, const float* emb_s, int emb_qt, int dim, const int* tokens) {
    const int tok = tokens[blockIdx.x];
    float* xo = x + (size_t)blockIdx.x * dim;
    for (int e = threadIdx.x; e < dim; e += blockDim.x) {
        float v;
        if (emb_qt == 0) v = reinterpret_cast<const float*>(emb)[(size_t)tok * dim + e];
        else {
            const float s = emb_s[((size_t)tok * dim + e) / kGroup];
            const int q = emb_qt == QT_INT8 ? (int)reinterpret_cast<const int8_t*>(emb)[(size_t)tok * dim + e]
                                            : (int)reinterpret_cast<const int16_t*>(emb)[(size_t)tok * dim + e];
            v = __fmul_rn((float)q, s);                              // dequantize_ quant_operators.cpp:49-65
        }
        xo[e] = v;
    }
}

struct RowsArgs {
    const float* x;          // [B][n]
    const float* norm_w;     // [n] (RMSNORM_QUANT)
    void* xq; float* xs;     // [B][n] quantized, [B][n/64] scales
    int n;
};
template <int QT, int PRO, int XR>
__global__ void __launch_bounds__(kGemvBlock) k_rows_prologue(const RowsArgs r) {
    using T = QTraits<QT>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    GemvArgs a{};
    a.n = r.n; a.x = r.x + (size_t)blockIdx.x * r.n; a.norm_w = r.norm_w;
    a.rows_per_pass = 4; a.cb_shift = 4;                                     // (only the fixed LDS offsets are used)
    float4 xv[XR > 0 ? XR : 1], nv[XR > 0 ? XR : 1];
    gemv_preload<QT, PRO, XR>(a, xv, nv);
    gemv_prologue<QT, PRO, XR>(a, lds, xv, nv, [](int) {});
    const GemvLds L = gemv_lds_layout(r.n, T::kEsz, true, 4, 4, false);
    const int nb16 = r.n * T::kEsz / 16, sn = r.n / kGroup;
    int4* qo = reinterpret_cast<int4*>(reinterpret_cast<char*>(r.xq) + (size_t)blockIdx.x * r.n * T::kEsz);
    for (int c = threadIdx.x; c < nb16; c += kGemvBlock) qo[c] = reinterpret_cast<const int4*>(lds)[c];
    const float* xs = reinterpret_cast<const float*>(lds + L.off_xs);
    for (int g = threadIdx.x; g < sn; g += kGemvBlock) r.xs[(size_t)blockIdx.x * sn + g] = xs[g];
}

struct GemmArgs {
    const void* W; const float* sW;      // [rows][n], [rows][n/64]
    const void* Xq; const float* Xs;     // [B][n], [B][n/64]
    float* out; int ldo;                 // out[b * ldo + row]
    int n, rows, B;
};
// One workgroup: 64 rows x 64 tokens, thread (ty, tx) owns rows 4ty..4ty+3 x tokens 4tx..4tx+3.  Per quant group the
// 64-row and 64-token slices (64 or 128 bytes each) go through LDS (double buffered; rows padded by 16 B: conflict-free
// 16-byte reads), int32 dots with v_dot4 / v_dot2, then the reference's fp32 chain step for the 16 outputs of the thread.
template <int QT, int EPI>
__global__ void __launch_bounds__(256) k_gemm_q(const GemmArgs a) {
    using T = QTraits<QT>;
    constexpr int GB = kGroup * T::kEsz;          // bytes of a group in one row
    constexpr int NCH = GB / 16;                  // 16-byte chunks per group
    constexpr int LS = GB + 16;                   // LDS row stride
    constexpr int NLD = 64 * NCH / 256;           // 16-byte pieces per thread and tile
    __shared__ __attribute__((aligned(16))) char Wt[2][64 * LS];
    __shared__ __attribute__((aligned(16))) char Xt[2][64 * LS];
    __shared__ float sWt[2][64], sXt[2][64];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int ntt = (a.B + 63) / 64;
    const int r0 = (blockIdx.x / ntt) * 64, b0 = (blockIdx.x % ntt) * 64;   // token tile fastest: neighbours share the weight rows
    const int sn = a.n / kGroup;
    const size_t rowbytes = (size_t)a.n * T::kEsz;
    const char* Wb = reinterpret_cast<const char*>(a.W);
    const char* Xb = reinterpret_cast<const char*>(a.Xq);
    v4i wr[NLD], xr[NLD]; float sr = 0.f;
    auto fetch = [&](int g) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + k * 256, row = idx / NCH, ch = idx % NCH;
            wr[k] = (r0 + row < a.rows) ? *reinterpret_cast<const v4i*>(Wb + (size_t)(r0 + row) * rowbytes + (size_t)g * GB + ch * 16) : v4i{0, 0, 0, 0};
            xr[k] = (b0 + row < a.B)    ? *reinterpret_cast<const v4i*>(Xb + (size_t)(b0 + row) * rowbytes + (size_t)g * GB + ch * 16) : v4i{0, 0, 0, 0};
        }
        if (tid < 64) sr = (r0 + tid < a.rows) ? a.sW[(size_t)(r0 + tid) * sn + g] : 0.f;
        else if (tid < 128) sr = (b0 + tid - 64 < a.B) ? a.Xs[(size_t)(b0 + tid - 64) * sn + g] : 0.f;
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + k * 256, row = idx / NCH, ch = idx % NCH;
            *reinterpret_cast<v4i*>(&Wt[buf][row * LS + ch * 16]) = wr[k];
            *reinterpret_cast<v4i*>(&Xt[buf][row * LS + ch * 16]) = xr[k];
        }
        if (tid < 64) sWt[buf][tid] = sr; else if (tid < 128) sXt[buf][tid - 64] = sr;
    };
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    fetch(0); park(0);
    __syncthreads();
    for (int g = 0; g < sn; ++g) {
        const int buf = g & 1;
        if (g + 1 < sn) fetch(g + 1);
        int d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = 0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            v4i w[4], x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = *reinterpret_cast<const v4i*>(&Wt[buf][(ty * 4 + i) * LS + ch * 16]);
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = *reinterpret_cast<const v4i*>(&Xt[buf][(tx * 4 + j) * LS + ch * 16]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (QT == QT_INT8) d[i][j] = dot16_i8(w[i], x[j], d[i][j]); else d[i][j] = dot8_i16(w[i], x[j], d[i][j]);
                }
        }
        float sw[4], sx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { sw[i] = sWt[buf][ty * 4 + i]; sx[i] = sXt[buf][tx * 4 + i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __fmaf_rn(__fmul_rn(sw[i], sx[j]), (float)d[i][j], acc[i][j]);   // quant_operators.cpp:274
        if (g + 1 < sn) park(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = b0 + tx * 4 + j;
        if (b >= a.B) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + ty * 4 + i;
            if (row >= a.rows) continue;
            float* o = a.out + (size_t)b * a.ldo + row;
            if constexpr (EPI == EPI_RESIDUAL) *o = __fadd_rn(*o, acc[i][j]); else *o = acc[i][j];
        }
    }
}

// The int8 GEMM on the matrix cores (gfx950 v_mfma_i32_32x32x32_i8): same 64 x 64 workgroup tile, four waves each owning a
// 32 x 32 (rows x tokens) quadrant.  Per quant group two MFMAs (K = 2 x 32) accumulate the group's 1024 int32 dots exactly
// (integer sums are order-free; A and B use the same byte -> k assignment: lane half h takes bytes 32kk + 16h .. +15 of the
// group); then every lane applies the reference's fp32 chain step to its 16 results -- that VALU work, not the MFMA, is what
// bounds the kernel.  C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
typedef int v16i __attribute__((ext_vector_type(16)));
template <int EPI>
__global__ void __launch_bounds__(256) k_gemm_q8_mfma(const GemmArgs a) {
    constexpr int GB = kGroup;                    // bytes of a group in one row (int8)
    constexpr int LS = GB + 16;                   // LDS row stride
    __shared__ __attribute__((aligned(16))) char Wt[2][64 * LS];
    __shared__ __attribute__((aligned(16))) char Xt[2][64 * LS];
    __shared__ __attribute__((aligned(16))) float sWt[2][64];
    __shared__ float sXt[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntt = (a.B + 63) / 64;
    const int r0 = (blockIdx.x / ntt) * 64, b0 = (blockIdx.x % ntt) * 64;   // token tile fastest: neighbours share the weight rows
    const int wr0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;                // this wave's quadrant inside the tile
    const int sn = a.n / kGroup;
    const size_t rowbytes = (size_t)a.n;
    const char* Wb = reinterpret_cast<const char*>(a.W);
    const char* Xb = reinterpret_cast<const char*>(a.Xq);
    v4i wr, xr; float sr = 0.f;
    const int lrow = tid >> 2, lch = tid & 3;                               // loader: 64 rows x 4 chunks of 16 B
    auto fetch = [&](int g) {
        wr = (r0 + lrow < a.rows) ? *reinterpret_cast<const v4i*>(Wb + (size_t)(r0 + lrow) * rowbytes + (size_t)g * GB + lch * 16) : v4i{0, 0, 0, 0};
        xr = (b0 + lrow < a.B)    ? *reinterpret_cast<const v4i*>(Xb + (size_t)(b0 + lrow) * rowbytes + (size_t)g * GB + lch * 16) : v4i{0, 0, 0, 0};
        if (tid < 64) sr = (r0 + tid < a.rows) ? a.sW[(size_t)(r0 + tid) * sn + g] : 0.f;
        else if (tid < 128) sr = (b0 + tid - 64 < a.B) ? a.Xs[(size_t)(b0 + tid - 64) * sn + g] : 0.f;
    };
    auto park = [&](int buf) {
        *reinterpret_cast<v4i*>(&Wt[buf][lrow * LS + lch * 16]) = wr;
        *reinterpret_cast<v4i*>(&Xt[buf][lrow * LS + lch * 16]) = xr;
        if (tid < 64) sWt[buf][tid] = sr; else if (tid < 128) sXt[buf][tid - 64] = sr;
    };
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int am = wr0 + (lane & 31), bn = wc0 + (lane & 31), kh = (lane >> 5) * 16;
    fetch(0); park(0);
    __syncthreads();
    for (int g = 0; g < sn; ++g) {
        const int buf = g & 1;
        if (g + 1 < sn) fetch(g + 1);
        const v4i a0 = *reinterpret_cast<const v4i*>(&Wt[buf][am * LS + kh]), a1 = *reinterpret_cast<const v4i*>(&Wt[buf][am * LS + 32 + kh]);
        const v4i x0 = *reinterpret_cast<const v4i*>(&Xt[buf][bn * LS + kh]), x1 = *reinterpret_cast<const v4i*>(&Xt[buf][bn * LS + 32 + kh]);
        v16i d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, x0, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, x1, d, 0, 0, 0);
        const float sx = sXt[buf][bn];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 sw = *reinterpret_cast<const float4*>(&sWt[buf][wr0 + 8 * q + 4 * (lane >> 5)]);
            acc[4 * q + 0] = __fmaf_rn(__fmul_rn(sw.x, sx), (float)d[4 * q + 0], acc[4 * q + 0]);   // quant_operators.cpp:274
            acc[4 * q + 1] = __fmaf_rn(__fmul_rn(sw.y, sx), (float)d[4 * q + 1], acc[4 * q + 1]);
            acc[4 * q + 2] = __fmaf_rn(__fmul_rn(sw.z, sx), (float)d[4 * q + 2], acc[4 * q + 2]);
            acc[4 * q + 3] = __fmaf_rn(__fmul_rn(sw.w, sx), (float)d[4 * q + 3], acc[4 * q + 3]);
        }
        if (g + 1 < sn) park(buf ^ 1);
        __syncthreads();
    }
    const int b = b0 + bn;
    if (b < a.B) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = r0 + wr0 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            if (row >= a.rows) continue;
            float* o = a.out + (size_t)b * a.ldo + row;
            if constexpr (EPI == EPI_RESIDUAL) *o = __fadd_rn(*o, acc[i]); else *o = acc[i];
        }
    }
}

// qkv[b] = [q ; k ; v] (dim each) of token b at position pos0 + b: RoPE on q and k (rope_v2 pairs), q -> qout[b], k / v -> cache rows
__global__ void k_rope_kv_rows(const float* qkv, float* qout, float* kcache, float* vcache, const float* rope_cos, const float* rope_sin,
                               int dim, int hs, int max_seq, int pos0) {
    const int b = blockIdx.x, pos = pos0 + b;
    const float* in = qkv + (size_t)b * 3 * dim;
    for (int i = threadIdx.x; i < dim / 2; i += blockDim.x) {
        const int row = 2 * i, h = row / hs, d = row - h * hs;
        const float c = rope_cos[(size_t)pos * (hs / 2) + d / 2], s = rope_sin[(size_t)pos * (hs / 2) + d / 2];
        float o0, o1;
        rope_pair(in[row], in[row + 1], c, s, o0, o1);
        qout[(size_t)b * dim + row] = o0; qout[(size_t)b * dim + row + 1] = o1;
        rope_pair(in[dim + row], in[dim + row + 1], c, s, o0, o1);
        float* kp = kcache + ((size_t)h * max_seq + pos) * hs + d; kp[0] = o0; kp[1] = o1;
        float* vp = vcache + ((size_t)h * max_seq + pos) * hs + d; vp[0] = in[2 * dim + row]; vp[1] = in[2 * dim + row + 1];
    }
}

__global__ void k_swiglu_rows(float* hd, const float* gu, int hidden) {
    const float* g = gu + (size_t)blockIdx.x * 2 * hidden;
    float* o = hd + (size_t)blockIdx.x * hidden;
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) o[i] = swiglu_elem(g[i], g[hidden + i]);   // o1.swiglu(o3) transformer.cpp:481
}

// ------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------
// x1 = embedding[token] (copy or dequantize; transformer.cpp:115-122)
__global__ void k_embed(float* x, const void* emb, const float* emb_s, int emb_qt, int dim, const int* tok_ptr, unsigned* bar) {
    const int tok = *tok_ptr;
    if (bar && blockIdx.x == 0) { bar[threadIdx.x * 16] = 0; bar[(threadIdx.x + blockDim.x) * 16] = 0; }   // grid barrier flags (<= 512 workgroups, 64 B apart) of the k_token that follows
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
// square_sum both ways: out[0] the wave-parallel evaluation (sq_chain_wave), out[1] the plain sequential chains (sq_chain)
__global__ void __launch_bounds__(256) k_op_square_sum(float* out, const float* x, int n) {
    extern __shared__ float sm[];
    const int n4 = n / 4, ns = n4 + 8;
    for (int e = threadIdx.x; e < n4 * 4; e += blockDim.x) sm[(e & 3) * ns + (e >> 2)] = x[e];
    for (int i = threadIdx.x; i < 4 * 8; i += blockDim.x) sm[(i >> 3) * ns + n4 + (i & 7)] = 0.f;
    __shared__ float red[8];
    __syncthreads();
    int its = 0;
    const float l = sq_chain_wave(sm + (threadIdx.x >> 6) * ns, n4, &its);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = l; out[6 + (threadIdx.x >> 6)] = (float)its; }
    if (threadIdx.x < 4) red[4 + threadIdx.x] = sq_chain(sm + threadIdx.x * ns, n4);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, red[0]), red[1]), red[2]), red[3]);
        out[1] = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, red[4]), red[5]), red[6]), red[7]);
        out[2] = red[0]; out[3] = red[1]; out[4] = red[2]; out[5] = red[3];
    }
}
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
