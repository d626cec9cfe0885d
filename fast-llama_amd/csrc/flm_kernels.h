// flm_kernels.h -- hand-written gfx950 (CDNA4) kernels for the fast-llama per-token hot path.
//
// Everything here is written for wave64 / MI355X only.  Reference citations are paths inside
// CoderLSF/fast-llama (the CPU engine whose arithmetic these kernels reproduce).
//
// Kernel inventory (one decode token = embed + L x {qkv, attn, attn_o, ffn13, ffn2} + cls + argmax):
//   k_gemv<QT,PRO,EPI>   group-quantized GEMV  out = W.q(x), HBM-bound; fused prologue
//                        (rmsnorm+quantize | quantize | split-attention combine+quantize) and
//                        epilogue (store | residual add | SwiGLU | RoPE + KV-cache append)
//   k_attn_decode        fp32 single-query attention over the fp32 KV cache, split over positions
//   k_embed, k_argmax    embedding row gather, first-max argmax
// plus small op-level kernels that expose the same __device__ functions to the parity tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flm {

constexpr int kWave = 64;
constexpr int kBlock = 256;           // 4 waves per workgroup everywhere
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kGroup = 64;            // quantization group (QUANT_GROUP_SIZE, the only value the reference uses)

enum { QT_INT16 = 1, QT_INT8 = 2 };
enum Prologue { PRO_NONE = 0, PRO_QUANT = 1, PRO_RMSNORM_QUANT = 2, PRO_ATTN_COMBINE_QUANT = 3 };
enum Epilogue { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2, EPI_ROPE_KV = 3 };

template <int QT> struct QTraits;
template <> struct QTraits<QT_INT8>  { using elem = int8_t;  static constexpr int kEsz = 1; static constexpr int kEPC = 16; static constexpr float kF = 127.0f; };
template <> struct QTraits<QT_INT16> { using elem = int16_t; static constexpr int kEsz = 2; static constexpr int kEPC = 8;  static constexpr float kF = 5792.0f; };
// kEPC = elements per 16-byte chunk; lanes per quant group = 64 / kEPC

// ------------------------------------------------------------------------------------------
// wave / block reductions (wave64: DPP-backed __shfl_xor)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}
// sum over the 256 threads of the block in a fixed order; red = 4 floats of LDS.  All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ------------------------------------------------------------------------------------------
// scalar pieces shared by the fused kernels and the op-level test kernels
// ------------------------------------------------------------------------------------------
// quant::quantize<T> element step (src/blas/quant_operators.cpp:26-47): q = (T)(x / r), C truncation.
// r == 0 (all-zero group): x/r is NaN; the x86 reference yields 0, stated explicitly here.
__device__ __forceinline__ int quant_elem(float x, float r) {
    float t = __fdiv_rn(x, r);          // IEEE-correct fp32 divide, never the fast reciprocal
    return (r == 0.0f) ? 0 : (int)t;    // v_cvt_i32_f32 truncates toward zero
}
// simd::rmsnorm scale (src/platforms/arch/x86_simd.cpp:1754-1756): r = float(1. / sqrtf(ss/n + 1e-5f))
__device__ __forceinline__ float rms_scale(float ss, int n) {
    float v = __fadd_rn(__fdiv_rn(ss, (float)n), 1e-5f);
    return (float)(1.0 / (double)__fsqrt_rn(v));
}
// simd::swiglu (x86_simd.cpp:1766-1770): evaluated in double, rounded to float
__device__ __forceinline__ float swiglu_elem(float a, float b) {
    return (float)((double)a / (1.0 + (double)expf(-a)) * (double)b);
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
__device__ __forceinline__ int dot8_i16(const v4i& w, const v4i& a, int acc) {
    acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, w.x), __builtin_bit_cast(short2_t, a.x), acc, false);
    acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, w.y), __builtin_bit_cast(short2_t, a.y), acc, false);
    acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, w.z), __builtin_bit_cast(short2_t, a.z), acc, false);
    acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, w.w), __builtin_bit_cast(short2_t, a.w), acc, false);
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
    // prologue inputs
    const float* x;                             // fp32 activation [n]          (QUANT / RMSNORM_QUANT)
    const float* norm_w;                        // rmsnorm weight [n]           (RMSNORM_QUANT)
    const void*  xq; const float* xs;           // pre-quantized activation     (NONE)
    const float* att_part; int n_splits; int hs;// split-attention partials     (ATTN_COMBINE_QUANT)
    // epilogue outputs
    float* out;                                 // STORE: out[row]; RESIDUAL: out[row] += ; SWIGLU: hd[i]; ROPE_KV: q[row]
    float* kcache; float* vcache;               // ROPE_KV: this layer's caches [heads][max_seq][hs]
    const float* rope_cos; const float* rope_sin; // [max_seq][hs/2]
    const int* pos_ptr;                         // device-resident position
    int dim; int kv_dim; int max_seq;           // ROPE_KV geometry (hs reused)
    // debugging taps used by the op-level exports (may be null)
    void* dbg_xq; float* dbg_xs; float* dbg_xn;
};

// LDS layout: [xq : n*esz bytes (16-aligned)] [xs : n/64 floats] [red : 8 floats]
__host__ __device__ inline size_t gemv_lds_bytes(int n, int esz) {
    return (size_t)n * esz + (size_t)(n / kGroup) * 4 + 64;
}

// ------------------------------------------------------------------------------------------
// Prologue: produce the quantized activation vector in LDS.  Every workgroup recomputes it
// (n <= 16K floats out of L2) so that no separate norm/quantize kernel sits on the critical path.
//   RMSNORM_QUANT == x2.rmsnorm(x1, w) ; qx.quantize(x2)   (transformer.cpp:132-134, 144-146, 155-156)
//   QUANT         == qx.quantize(x2) / qh.quantize(hd)     (transformer.cpp:138, 149)
// Thread t owns elements 4t..4t+3 (+1024 per round): 16 consecutive lanes own one 64-group, so the
// group max is a 16-lane xor-butterfly.
// ------------------------------------------------------------------------------------------
// The first XR rounds of x (and of the norm weight) are handed in as registers that the caller
// loaded BEFORE issuing its first batch of weight loads: loads return in issue order, so an x load
// issued behind 32 HBM weight loads would make the whole prologue wait for them.
constexpr int kAttnPartPad = 4;       // split-attention partial record: [m, l, -, -, o[hs]]

template <int QT, int PRO, int XR>
__device__ __forceinline__ void gemv_preload(const GemvArgs& a, float4 (&xv)[XR > 0 ? XR : 1], float4 (&wv)[XR > 0 ? XR : 1]) {
    if constexpr (PRO == PRO_QUANT || PRO == PRO_RMSNORM_QUANT) {
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int e = threadIdx.x * 4 + i * kBlock * 4;
            xv[i] = e < a.n ? *reinterpret_cast<const float4*>(a.x + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PRO == PRO_RMSNORM_QUANT)
                wv[i] = e < a.n ? *reinterpret_cast<const float4*>(a.norm_w + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

template <int QT, int PRO, int XR>
__device__ __forceinline__ void gemv_prologue(const GemvArgs& a, char* lds, float4 (&xv)[XR > 0 ? XR : 1], float4 (&wv)[XR > 0 ? XR : 1]) {
    using T = QTraits<QT>;
    const int n = a.n;
    const int tid = threadIdx.x;
    char*  xq = lds;
    float* xs = reinterpret_cast<float*>(lds + (size_t)n * T::kEsz);
    float* red = xs + n / kGroup;

    if constexpr (PRO == PRO_NONE) {
        // copy pre-quantized activation (op-level matmul and generic callers)
        const int nb16 = n * T::kEsz / 16;
        for (int c = tid; c < nb16; c += kBlock)
            reinterpret_cast<int4*>(xq)[c] = reinterpret_cast<const int4*>(a.xq)[c];
        for (int g = tid; g < n / kGroup; g += kBlock) xs[g] = a.xs[g];
        __syncthreads();
        return;
    } else {
        const int rounds = (n + kBlock * 4 - 1) / (kBlock * 4);
        float r = 1.0f;
        if constexpr (PRO == PRO_RMSNORM_QUANT) {
            float ss = 0.f;
            auto sq = [&](const float4& v) {
                ss = __fmaf_rn(v.x, v.x, ss); ss = __fmaf_rn(v.y, v.y, ss);
                ss = __fmaf_rn(v.z, v.z, ss); ss = __fmaf_rn(v.w, v.w, ss);
            };
#pragma unroll
            for (int i = 0; i < XR; ++i) sq(xv[i]);                     // out-of-range lanes hold zeros
            for (int i = XR; i < rounds; ++i) {
                const int e = tid * 4 + i * kBlock * 4;
                if (e < n) sq(*reinterpret_cast<const float4*>(a.x + e));
            }
            ss = block_sum(ss, red);
            r = rms_scale(ss, n);
        }
        // one round: (normalise,) group max over 16 lanes, quantize, pack into LDS
        auto round = [&](int i, float4 v, float4 w) {
            const int e = tid * 4 + i * kBlock * 4;
            const bool act = e < n;
            if constexpr (PRO == PRO_RMSNORM_QUANT) {
                // multiply_avx256 (x86_simd.cpp:1360-1372): (x*w)*r
                v.x = __fmul_rn(__fmul_rn(v.x, w.x), r); v.y = __fmul_rn(__fmul_rn(v.y, w.y), r);
                v.z = __fmul_rn(__fmul_rn(v.z, w.z), r); v.w = __fmul_rn(__fmul_rn(v.w, w.w), r);
            }
            if (act && a.dbg_xn && blockIdx.x == 0) *reinterpret_cast<float4*>(a.dbg_xn + e) = v;
            // group max over the 16 lanes that share this 64-element group (order-free, exact)
            float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, kWave));
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
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (PRO == PRO_ATTN_COMBINE_QUANT) {
            for (int i = 0; i < rounds; ++i) {
                const int e = tid * 4 + i * kBlock * 4;
                float4 v = z4;
                if (e < n) {
                    // merge the split-attention partials of the head that owns e: [m, l, -, -, o[hs]] per split
                    const int hs = a.hs, S = a.n_splits, ps_stride = hs + kAttnPartPad;
                    const int h = e / hs, d = e - h * hs;
                    const float* p = a.att_part + (size_t)h * S * ps_stride;
                    float M = -INFINITY;
                    for (int s = 0; s < S; ++s) M = fmaxf(M, p[(size_t)s * ps_stride]);
                    float L = 0.f; float4 o = z4;
                    for (int s = 0; s < S; ++s) {
                        const float* ps = p + (size_t)s * ps_stride;
                        const float ls = ps[1];
                        if (ls > 0.f) {
                            const float w = expf(ps[0] - M);
                            L = __fmaf_rn(ls, w, L);
                            const float4 os = *reinterpret_cast<const float4*>(ps + kAttnPartPad + d);
                            o.x = __fmaf_rn(os.x, w, o.x); o.y = __fmaf_rn(os.y, w, o.y);
                            o.z = __fmaf_rn(os.z, w, o.z); o.w = __fmaf_rn(os.w, w, o.w);
                        }
                    }
                    v = make_float4(__fdiv_rn(o.x, L), __fdiv_rn(o.y, L), __fdiv_rn(o.z, L), __fdiv_rn(o.w, L));
                }
                round(i, v, z4);
            }
        } else {
#pragma unroll
            for (int i = 0; i < XR; ++i) { if (i < rounds) round(i, xv[i], wv[i]); }
            for (int i = XR; i < rounds; ++i) {
                const int e = tid * 4 + i * kBlock * 4;
                float4 v = z4, w = z4;
                if (e < n) {
                    v = *reinterpret_cast<const float4*>(a.x + e);
                    if constexpr (PRO == PRO_RMSNORM_QUANT) w = *reinterpret_cast<const float4*>(a.norm_w + e);
                }
                round(i, v, w);
            }
        }
        __syncthreads();
        if (a.dbg_xq && blockIdx.x == 0) {
            const int nb4 = n * T::kEsz / 4;
            for (int c = tid; c < nb4; c += kBlock) reinterpret_cast<uint32_t*>(a.dbg_xq)[c] = reinterpret_cast<uint32_t*>(xq)[c];
            for (int g = tid; g < n / kGroup; g += kBlock) a.dbg_xs[g] = xs[g];
        }
    }
}

// ------------------------------------------------------------------------------------------
// The GEMV.  quant::matmul<T> at w == 1 (src/blas/quant_operators.cpp:252-284):
//     out[r] = sum_g (sW[r,g] * sX[g]) * float( sum_{k<64} W[r,64g+k] * X[64g+k] )
// Mapping: one wave per row, lanes along K in 16-byte chunks (lane l owns chunks l, l+64, ...), so
// every weight load is a fully coalesced 1 KiB global_load_dwordx4 and a row is contiguous in HBM.
// Each lane applies the group scale to its own 16-element (8 for int16) partial dot -- exact in
// fp32 for int8 (|partial| < 2^24) -- and a 6-step xor butterfly finishes the row.
// Rows are processed kRows at a time so that kRows*4 x 16 B loads per lane are in flight; with
// several workgroups per CU that keeps > 100 KB outstanding per CU, enough to cover HBM latency.
// ------------------------------------------------------------------------------------------
constexpr int kRows = 4;

template <int QT, int PRO, int EPI, int XR>
__global__ void __launch_bounds__(kBlock) k_gemv(const GemvArgs a) {
    using T = QTraits<QT>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int n = a.n;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int RPI = (EPI == EPI_SWIGLU || EPI == EPI_ROPE_KV) ? 2 : 1;   // rows per item
    constexpr int IPB = kRows / RPI;                                          // items per batch
    const int rowbytes = n * T::kEsz;
    const int nchunks = rowbytes / 16;
    const int sn = n / kGroup;
    constexpr int LPG = 64 / T::kEPC;                                         // lanes per quant group

    // balanced contiguous item range for this wave
    const long long nw = (long long)gridDim.x * kWavesPerBlock;
    const long long gw = (long long)blockIdx.x * kWavesPerBlock + wave;
    const int it0 = (int)((long long)a.items * gw / nw);
    const int it1 = (int)((long long)a.items * (gw + 1) / nw);

    auto row_w = [&](int item, int which) -> const char* {
        if constexpr (EPI == EPI_SWIGLU) return reinterpret_cast<const char*>(which ? a.W2nd : a.W) + (size_t)item * rowbytes;
        else return reinterpret_cast<const char*>(a.W) + (size_t)(item * RPI + which) * rowbytes;
    };
    auto row_s = [&](int item, int which) -> const float* {
        if constexpr (EPI == EPI_SWIGLU) return (which ? a.sW2nd : a.sW) + (size_t)item * sn;
        else return a.sW + (size_t)(item * RPI + which) * sn;
    };

    // 1. the activation (L2-resident) first, 2. then the first batch of weight loads, both BEFORE
    // the prologue: weights do not depend on the activation, so HBM latency overlaps norm/quantize.
    float4 xv[XR > 0 ? XR : 1], nv[XR > 0 ? XR : 1];
    gemv_preload<QT, PRO, XR>(a, xv, nv);

    v4i   w[kRows][4];
    float sw[kRows][4];
    // FULL: all kRows rows exist and all four 64-chunk columns are inside the row -> no predication,
    // one base address per row, the four chunk loads differ by an immediate offset of 1 KiB.
    auto load_batch = [&](int item0, int jb) {
        const bool full = (item0 + IPB <= it1) && (64 * (jb + 4) <= nchunks);
        if (full) {
#pragma unroll
            for (int rr = 0; rr < kRows; ++rr) {
                const v4i*   wp = reinterpret_cast<const v4i*>(row_w(item0 + rr / RPI, rr % RPI)) + lane + 64 * jb;
                const float* sp = row_s(item0 + rr / RPI, rr % RPI) + lane / LPG + (64 / LPG) * jb;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    w[rr][jj]  = __builtin_nontemporal_load(wp + 64 * jj);
                    sw[rr][jj] = __builtin_nontemporal_load(sp + (64 / LPG) * jj);
                }
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < kRows; ++rr) {
                const int item = item0 + rr / RPI;
                const bool rv = item < it1;
                const char*  wp = row_w(rv ? item : it0, rr % RPI);
                const float* sp = row_s(rv ? item : it0, rr % RPI);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int c = lane + 64 * (jb + jj);
                    const bool v = rv && c < nchunks;
                    const v4i z = {0, 0, 0, 0};
                    w[rr][jj] = v ? __builtin_nontemporal_load(reinterpret_cast<const v4i*>(wp) + c) : z;
                    sw[rr][jj] = v ? __builtin_nontemporal_load(sp + c / LPG) : 0.f;
                }
            }
        }
    };
    if (it0 < it1) load_batch(it0, 0);

    gemv_prologue<QT, PRO, XR>(a, lds, xv, nv);

    const v4i*   xq = reinterpret_cast<const v4i*>(lds);
    const float* xs = reinterpret_cast<const float*>(lds + (size_t)n * T::kEsz);
    const int NJ = (nchunks + 63) / 64;
    int pos = 0;
    if constexpr (EPI == EPI_ROPE_KV) pos = *a.pos_ptr;

    for (int item0 = it0; item0 < it1; item0 += IPB) {
        // epilogue operands that live in memory are fetched up front (behind the weight loads, no extra round trip)
        float resid = 0.f, rc = 0.f, rs = 0.f;
        if constexpr (EPI == EPI_RESIDUAL) {
            if (lane < kRows && item0 + lane < it1) resid = a.out[item0 + lane];
        }
        if constexpr (EPI == EPI_ROPE_KV) {
            const int i = item0 + lane;
            if (lane < IPB && i < it1 && 2 * i < a.dim + a.kv_dim) {
                const int row = 2 * i, rr = row < a.dim ? row : row - a.dim;
                const int d = rr % a.hs;
                rc = a.rope_cos[(size_t)pos * (a.hs / 2) + d / 2];
                rs = a.rope_sin[(size_t)pos * (a.hs / 2) + d / 2];
            }
        }
        float acc[kRows];
#pragma unroll
        for (int rr = 0; rr < kRows; ++rr) acc[rr] = 0.f;
        for (int jb = 0; jb < NJ; jb += 4) {
            if (!(item0 == it0 && jb == 0)) load_batch(item0, jb);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int c = lane + 64 * (jb + jj);
                if (c < nchunks) {
                    const v4i av = xq[c];
                    const float sx = xs[c / LPG];
#pragma unroll
                    for (int rr = 0; rr < kRows; ++rr) {
                        const int p = dot_chunk<QT>(w[rr][jj], av);
                        acc[rr] = __fmaf_rn(__fmul_rn(sw[rr][jj], sx), (float)p, acc[rr]);
                    }
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < kRows; ++rr) acc[rr] = wave_sum(acc[rr]);

        // ---------------- epilogues (lane ii handles item item0+ii) ----------------
        if constexpr (EPI == EPI_STORE || EPI == EPI_RESIDUAL) {
            const float v = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
            const int row = item0 + lane;
            if (lane < kRows && row < it1) {
                if constexpr (EPI == EPI_STORE) a.out[row] = v;
                else a.out[row] = __fadd_rn(resid, v);               // o.add(tmp, offset) transformer.cpp:465,493
            }
        } else if constexpr (EPI == EPI_SWIGLU) {
            // o1.swiglu(o3) transformer.cpp:481
            const float g = lane == 0 ? acc[0] : acc[2];
            const float u = lane == 0 ? acc[1] : acc[3];
            const int i = item0 + lane;
            if (lane < IPB && i < it1) a.out[i] = swiglu_elem(g, u);
        } else {   // EPI_ROPE_KV: rows (2i, 2i+1) of [Wq;Wk;Wv]; RoPE on q and k, append k,v to the cache
            const float x0 = lane == 0 ? acc[0] : acc[2];
            const float x1 = lane == 0 ? acc[1] : acc[3];
            const int i = item0 + lane;
            if (lane < IPB && i < it1) {
                const int row = 2 * i, hs = a.hs;
                if (row < a.dim + a.kv_dim) {
                    const int rr = row < a.dim ? row : row - a.dim;
                    const int h = rr / hs, d = rr - h * hs;
                    float o0, o1;
                    rope_pair(x0, x1, rc, rs, o0, o1);
                    if (row < a.dim) { a.out[row] = o0; a.out[row + 1] = o1; }
                    else { float* kp = a.kcache + ((size_t)h * a.max_seq + pos) * hs + d; kp[0] = o0; kp[1] = o1; }
                } else {
                    const int rr = row - a.dim - a.kv_dim;
                    const int h = rr / hs, d = rr - h * hs;
                    float* vp = a.vcache + ((size_t)h * a.max_seq + pos) * hs + d; vp[0] = x0; vp[1] = x1;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Decode attention (execute_attn at bs == 1, transformer.cpp:397-455), all fp32.
//   att[t] = (K[t].q) * 1/sqrt(hs) ; softmax over t <= pos ; o = sum_t att[t] V[t]
// grid = (heads, splits); a split covers a contiguous range of positions.  With one split the
// kernel writes the normalised head output; with several it writes (max, sum, unnormalised o) and
// the consumer's prologue (PRO_ATTN_COMBINE_QUANT) merges them.
// K/V rows are hs fp32 = hs/4 lanes x float4, so a wave64 load covers 256/hs positions.
// ------------------------------------------------------------------------------------------
struct AttnArgs {
    const float* q;          // [heads*hs], RoPE already applied
    const float* kcache;     // [heads][max_seq][hs]
    const float* vcache;
    float* out;              // n_splits == 1: [heads*hs]; else partials [heads][splits][hs+4] = [m, l, -, -, o[hs]]
    const int* pos_ptr;
    int hs, max_seq, n_splits;
};

__global__ void __launch_bounds__(kBlock) k_attn_decode(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int hs = a.hs, h = blockIdx.x, S = a.n_splits, sp = blockIdx.y;
    const int T = *a.pos_ptr + 1;
    const int per = (T + S - 1) / S;
    const int t0 = sp * per, t1 = min(T, t0 + per);
    const int cnt = max(0, t1 - t0);
    float* sc  = reinterpret_cast<float*>(lds);                 // [per] scores -> probabilities
    float* red = sc + ((per + 3) & ~3);                          // 8
    float* ow  = red + 8;                                        // [4][hs] per-wave partial outputs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int lpp = 1; while (lpp * 4 < hs) lpp <<= 1;                 // lanes per position (power of two)
    const int ppw = 64 / lpp;                                    // positions per wave instruction
    const int sub = lane / lpp, li = lane % lpp;
    const bool dl = li * 4 < hs;                                 // lane carries real dims
    const float* K = a.kcache + (size_t)h * a.max_seq * hs;
    const float* V = a.vcache + (size_t)h * a.max_seq * hs;
    const float scale = (float)(1.0 / (double)__fsqrt_rn((float)hs));   // attn_scale, transformer.cpp:418

    float4 qv = dl ? *reinterpret_cast<const float4*>(a.q + (size_t)h * hs + li * 4) : make_float4(0, 0, 0, 0);
    float lmax = -INFINITY;
    for (int tb = wave * ppw; tb < cnt; tb += kWavesPerBlock * ppw) {
        const int t = tb + sub;
        const bool v = t < cnt && dl;
        float4 kv = v ? *reinterpret_cast<const float4*>(K + (size_t)(t0 + t) * hs + li * 4) : make_float4(0, 0, 0, 0);
        float d = __fmaf_rn(kv.w, qv.w, __fmaf_rn(kv.z, qv.z, __fmaf_rn(kv.y, qv.y, __fmul_rn(kv.x, qv.x))));
        for (int o = lpp >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, kWave);
        d = __fmul_rn(d, scale);                                // att.multiply(attn_scale) :443
        if (t < cnt) { if (li == 0) sc[t] = d; lmax = fmaxf(lmax, d); }
    }
    const float m = block_max(lmax, red);
    float lsum = 0.f;
    for (int t = threadIdx.x; t < cnt; t += kBlock) { const float e = expf(sc[t] - m); sc[t] = e; lsum += e; }   // softmax_sisd :180-183
    const float L = block_sum(lsum, red);
    // single split: p = e / L first (softmax_sisd :184-186), then weighted_sum (tf_operators.cpp:325-350)
    const float inv_mode = (S == 1) ? 1.f : 0.f;
    float4 o = make_float4(0, 0, 0, 0);
    for (int tb = wave * ppw; tb < cnt; tb += kWavesPerBlock * ppw) {
        const int t = tb + sub;
        if (t < cnt && dl) {
            const float4 vv = *reinterpret_cast<const float4*>(V + (size_t)(t0 + t) * hs + li * 4);
            const float p = inv_mode != 0.f ? __fdiv_rn(sc[t], L) : sc[t];
            o.x = __fmaf_rn(vv.x, p, o.x); o.y = __fmaf_rn(vv.y, p, o.y); o.z = __fmaf_rn(vv.z, p, o.z); o.w = __fmaf_rn(vv.w, p, o.w);
        }
    }
    for (int off = lpp; off < 64; off <<= 1) {
        o.x += __shfl_xor(o.x, off, kWave); o.y += __shfl_xor(o.y, off, kWave);
        o.z += __shfl_xor(o.z, off, kWave); o.w += __shfl_xor(o.w, off, kWave);
    }
    if (sub == 0 && dl) *reinterpret_cast<float4*>(ow + wave * hs + li * 4) = o;
    __syncthreads();
    for (int d = threadIdx.x; d < hs; d += kBlock) {
        const float r = ((ow[d] + ow[hs + d]) + ow[2 * hs + d]) + ow[3 * hs + d];
        if (S == 1) a.out[(size_t)h * hs + d] = r;
        else a.out[((size_t)h * S + sp) * (hs + kAttnPartPad) + kAttnPartPad + d] = r;
    }
    if (S > 1 && threadIdx.x == 0) {
        float* p = a.out + ((size_t)h * S + sp) * (hs + kAttnPartPad);
        p[0] = cnt > 0 ? m : -INFINITY; p[1] = cnt > 0 ? L : 0.f;
    }
}
__host__ inline size_t attn_lds_bytes(int per, int hs) { return (size_t)(((per + 3) & ~3) + 8 + 4 * hs) * 4; }

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
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const float v = logits[i]; if (v > best) { best = v; idx = i; } }
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
__global__ void k_op_rope(float* o, const float* x, int n_dims, const float* c, const float* s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n_dims) rope_pair(x[2 * i], x[2 * i + 1], c[i], s[i], o[2 * i], o[2 * i + 1]);
}
// softmax_sisd over n entries, one workgroup
__global__ void __launch_bounds__(kBlock) k_op_softmax(float* x, int n) {
    __shared__ float red[8];
    float lm = -INFINITY;
    for (int i = threadIdx.x; i < n; i += kBlock) lm = fmaxf(lm, x[i]);
    const float m = block_max(lm, red);
    float ls = 0.f;
    for (int i = threadIdx.x; i < n; i += kBlock) { const float e = expf(x[i] - m); x[i] = e; ls += e; }
    const float L = block_sum(ls, red);
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
// merge split-attention partials into [heads*hs] (same math as PRO_ATTN_COMBINE_QUANT), for flm_op_attention
__global__ void k_op_attn_combine(float* out, const float* part, int n_heads, int hs, int S) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_heads * hs) return;
    const int h = e / hs, d = e - h * hs;
    const int st = hs + kAttnPartPad;
    const float* p = part + (size_t)h * S * st;
    float M = -INFINITY;
    for (int s = 0; s < S; ++s) M = fmaxf(M, p[(size_t)s * st]);
    float L = 0.f, o = 0.f;
    for (int s = 0; s < S; ++s) {
        const float* ps = p + (size_t)s * st;
        if (ps[1] > 0.f) { const float w = expf(ps[0] - M); L = __fmaf_rn(ps[1], w, L); o = __fmaf_rn(ps[kAttnPartPad + d], w, o); }
    }
    out[e] = __fdiv_rn(o, L);
}

} // namespace flm
