// flm_prefill.h -- batched prompt processing: row prologues, int8/int16 GEMM tiles (MFMA / v_dot), RoPE + KV rows, SwiGLU rows.
// Part of flm_kernels.h (hand-written gfx950 / CDNA4 kernels of the fast-llama per-token hot path); include that header.
#pragma once
#include "flm_math.h"
#include "flm_gemv.h"
// (bit-exactness hygiene: see flm_math.h -- no implicit FMA contraction in any of these headers)
#pragma clang fp contract(off)

namespace flm {

// ------------------------------------------------------------------------------------------
// Batched prefill (ParallelTransformer::forward with bs > 1, transformer.cpp:105-161).  A prompt's tokens before the
// last one only have to leave their K/V rows in the cache; every (token, row) value is produced by the SAME chain as in
// the single-token kernels (group dots exact, acc = fma(sW*sX, float(dot), acc) with groups ascending; per-row rmsnorm
// chains; per-query attention), so the cache -- and therefore the logits of the last token, which runs through the
// decode kernels -- is bit-identical to feeding the prompt token by token, at a fraction of the time: the weights are
// streamed once per 64 tokens instead of once per token.
//   k_embed_rows        x[b] = embedding[token b]
//   k_rows_prologue     per token row: (rmsnorm,) quantize -> xq[b], xs[b]   (the decode prologue, one workgroup per row)
//   (k_gemm_q, the v_dot4 / v_dot2 tile kernel of round 1, is gone: 256 VGPRs, reachable only through "use_mfma" 0, 3-6x slower than the matrix-core
//    tiles that replaced it; "use_mfma" 0 now selects the 64 x 64 matrix-core tiles)
//   k_rope_kv_rows      RoPE on q and k of every token, K/V rows appended to the cache
//   k_attn_prefill      causal attention: one workgroup per (head, query), the decode attention with T = pos + i + 1
//   k_swiglu_rows       hd[b] = swiglu(gate[b], up[b])
// ------------------------------------------------------------------------------------------
inline __global__ void k_embed_rows(float* x, const void* emb, const float* emb_s, int emb_qt, int dim, const int* tokens) {
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
    float* xst;              // the scales once more, group-major [n/64][B] (the matrix-core GEMM tiles read 64 tokens' scales of a group as one 256-byte run); may be null
};
// COH: the rows were (partly) written by peer GPUs (tensor parallel: the exchange regions) -> system-coherent loads
template <int QT, int PRO, int XR, bool COH = false>
__global__ void __launch_bounds__(kGemvBlock) k_rows_prologue(const RowsArgs r) {
    using T = QTraits<QT>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    GemvArgs a{};
    a.n = r.n; a.x = r.x + (size_t)blockIdx.x * r.n; a.norm_w = r.norm_w;
    a.rows_per_pass = 4; a.cb_shift = 4;                                     // (only the fixed LDS offsets are used)
    float4 xv[XR > 0 ? XR : 1], nv[XR > 0 ? XR : 1];
    gemv_preload<QT, PRO, XR, COH>(a, xv, nv);
    gemv_prologue<QT, PRO, XR, COH>(a, lds, xv, nv, [](int) {});
    const GemvLds L = gemv_lds_layout(r.n, T::kEsz, true, 4, 4, false);
    const int nb16 = r.n * T::kEsz / 16, sn = r.n / kGroup;
    int4* qo = reinterpret_cast<int4*>(reinterpret_cast<char*>(r.xq) + (size_t)blockIdx.x * r.n * T::kEsz);
    for (int c = threadIdx.x; c < nb16; c += kGemvBlock) qo[c] = reinterpret_cast<const int4*>(lds)[c];
    const float* xs = reinterpret_cast<const float*>(lds + L.off_xs);
    for (int g = threadIdx.x; g < sn; g += kGemvBlock) { r.xs[(size_t)blockIdx.x * sn + g] = xs[g]; if (r.xst) r.xst[(size_t)g * gridDim.x + blockIdx.x] = xs[g]; }
}

struct GemmArgs {
    const void* W; const float* sW;      // [rows][n], [rows][n/64]
    const void* Xq; const float* Xs;     // [B][n], [B][n/64]
    float* out; int ldo;                 // out[b * ldo + row]
    int n, rows, B;
    const float* XsT;                    // Xs group-major, [n/64][B] (the matrix-core kernels)
    const float* sWT;                    // sW group-major, [n/64][rows] (the matrix-core kernels; SWIGLU: [n/64][2 rows])
    // EPI_ROPE_KV (the qkv GEMM of k_gemm_q8_mfma): rows [0, dim) = q, [dim, 2 dim) = k, [2 dim, 3 dim) = v of token b at position pos0 + b;
    // RoPE on q and k (rope_v2 pairs), q -> qout[b][dim], k / v -> the layer's cache rows (what k_rope_kv_rows does after a plain store)
    float* qout; float* kcache; float* vcache; const float* rope_cos; const float* rope_sin; int dim, hs, max_seq, pos0;
    // tensor parallel, peer-to-peer (EPI_RESIDUAL / EPI_SWIGLU of k_gemm_q8_mfma): `out` lies in this rank's exchange region (its old value
    // is read coherently) and every result also goes to the same place of every peer's region (system-scope stores), as GemvArgs::out_peer
    float* out_peer[7]; int n_peer;
};
// a result of a batched kernel that the other ranks need: local write-through store + one system-scope store per peer
__device__ __forceinline__ void st_result_tp(float* o, const size_t idx, const float v, float* const (&peer)[7], const int n_peer) {
    if (n_peer == 0) { o[idx] = v; return; }
    st_agent(o + idx, v);
    for (int i = 0; i < n_peer; ++i) __hip_atomic_store(peer[i] + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the group-major copy of a weight matrix's scales (made once, when the first prompt is batched)
inline __global__ void k_transpose_scales(const float* s, float* st, int rows, int sn) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)rows * sn) { const int r = (int)(i / sn), g = (int)(i - (size_t)r * sn); st[(size_t)g * rows + r] = s[i]; }
}
// The epilogue of the int8 matrix-core GEMMs (k_gemm_q8_mfma, k_gemm_q8_ring): a wave holds acc[j][i] = the finished chain of
// (token tb + (i & 3) + 8 (i >> 2) + 4 (lane >> 5), row rbase + 32 j) -- SWIGLU: j = 0 gate, j = 1 up of the SAME row rbase -- with rbase = the
// lane's row (lane & 31 inside a 32-row fragment).  Store | residual add | SwiGLU | RoPE + KV rows.
template <int EPI, int NB>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, const float (&acc)[NB][16], const int rbase, const int tb, const int lane) {
    constexpr bool TWO = EPI == EPI_SWIGLU;
    const int h = lane >> 5;
    if constexpr (EPI == EPI_ROPE_KV) {
        // a fragment's 32 rows lie inside one of q / k / v (dim is a multiple of 32) and a RoPE pair (rows 2i, 2i + 1) in neighbouring lanes
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int row = rbase + j * 32, which = row / a.dim, within = row - which * a.dim, hh = within / a.hs, dd = within - hh * a.hs;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int b = tb + (i & 3) + 8 * (i >> 2) + 4 * h, pos = a.pos0 + b;
                const float mine = acc[j][i], other = __shfl_xor(mine, 1, 64);
                if (row >= a.rows || b >= a.B) continue;
                float v = mine;
                if (which < 2) {
                    const float c = a.rope_cos[(size_t)pos * (a.hs / 2) + dd / 2], sn_ = a.rope_sin[(size_t)pos * (a.hs / 2) + dd / 2];
                    float o0, o1;
                    rope_pair((lane & 1) ? other : mine, (lane & 1) ? mine : other, c, sn_, o0, o1);
                    v = (lane & 1) ? o1 : o0;
                }
                if (which == 0) a.qout[(size_t)b * a.dim + within] = v;
                else (which == 1 ? a.kcache : a.vcache)[((size_t)hh * a.max_seq + pos) * a.hs + dd] = v;
            }
        }
    } else if constexpr (TWO) {
        const int row = rbase;
        if (row < a.rows) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int b = tb + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (b < a.B) st_result_tp(a.out, (size_t)b * a.ldo + row, swiglu_elem(acc[0][i], acc[NB - 1][i]), a.out_peer, a.n_peer);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int row = rbase + j * 32;
            if (row >= a.rows) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int b = tb + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (b >= a.B) continue;
                const size_t idx = (size_t)b * a.ldo + row;
                if constexpr (EPI == EPI_RESIDUAL) st_result_tp(a.out, idx, __fadd_rn(a.n_peer ? ld_agent(a.out + idx) : a.out[idx], acc[j][i]), a.out_peer, a.n_peer);
                else a.out[idx] = acc[j][i];
            }
        }
    }
}
// The int8 GEMM on the matrix cores (gfx950 v_mfma_i32_32x32x32_i8): a 64 x 64 (rows x tokens) workgroup tile, four waves each owning
// 32 tokens x 32 weight rows.  Per quant group two MFMAs (K = 2 x 32) accumulate the group's 1024 int32 dots exactly (integer sums
// are order-free; A and B use the same byte -> k assignment: lane half h takes bytes 32kk + 16h .. +15 of the group); then every
// lane applies the reference's fp32 chain step to its 16 results: v_cvt, v_mul, v_fma per output -- 48 VALU instructions per group
// against 2 MFMAs, so the VALU (and the LDS that feeds it), not the matrix pipe, bounds the kernel.
// A STAGE is kGPS = 2 consecutive quant groups (128 bytes of every row): one barrier and one pair of LDS buffers per stage.  Inside a
// stage the groups are consumed in ascending order: the chain is the reference's.
// Shaped by the counters (profiles/r02_prefill_gemm_pmc.txt): a first version held 190 registers (124 + 64 accumulation registers
// the compiler parked the MFMA results in, one v_accvgpr_read per element to get them back), so two waves shared a SIMD -- and a
// wave issues a VALU instruction every ~6.5 cycles at best (tools/ubench/valu_rate.hip: 1.4-1.9 cycles per instruction need four
// waves per SIMD; scalar fp32 forms overlap the MFMAs completely, packed forms do not): the SIMDs idled 2/3 of the time.
// Here: <= 128 registers (four waves per SIMD, MFMA results in plain VGPRs), one register slot of prefetch, tokens as the MFMA's
// A operand so that a lane owns ONE weight row (one weight scale per lane and group, coalesced stores: 32 lanes = 128 contiguous
// bytes of an output row), one scale load per thread and stage, five adds of address arithmetic per stage.
// C/D layout: col = lane & 31 (weight row), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (token).
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int kGPS = 2;
// Tile shapes: WT x WR waves, a wave = 32 tokens x (NB x 32) weight rows (NB fragments of the B operand share one A fragment and
// one set of activation scales).  <2, 2, 1>: 64 tokens x 64 rows, 256 threads, four workgroups per CU.  <4, 2, 2>: 128 x 128, 512
// threads, two per CU: per 32 x 32 x 64 products the LDS moves 36 instead of 62 cycles' worth (operand reads 12 / 16, scale reads
// 10 / 18, stores of the staged tiles 13 / 28) and half the bytes come through the CU's memory pipeline -- and the LDS is what
// bounds the kernel (removing the chain altogether changed nothing, removing the staging stores and loads halved the time:
// profiles/r02_prefill_gemm_pmc.txt).
template <int WT, int WR, int NB>
struct GemmTile {
    static constexpr int TT = 32 * WT, TR = 32 * NB * WR, NT = 64 * WT * WR;
    static constexpr int SB = kGPS * kGroup;      // bytes of a stage in one row
    static constexpr int LS = SB + 16;            // LDS row stride (conflict-free 16-byte reads of 16 consecutive rows)
    static constexpr int kOffX = TR * LS, kOffS = (TR + TT) * LS, kBuf = kOffS + kGPS * (TR + TT) * 4;   // W rows | X rows | scales [W g0][W g1][X g0][X g1]
    static constexpr int NPW = TR * 8 / NT, NPX = TT * 8 / NT;      // 16-byte pieces per thread and stage
    static constexpr int kLds = 2 * kBuf;
    static_assert(kGPS == 2 && NT == kGPS * (TR + TT) && TR % 64 == 0 && TT % 64 == 0, "one scale per thread and stage, wave-uniform (matrix, group)");
    static_assert(TR * 8 % NT == 0 && TT * 8 % NT == 0, "whole pieces per thread");
};
// EPI_SWIGLU (NB == 2): W = [W1 (gate) ; W3 (up)], a.rows rows each; a tile's rows are TR / 2 rows of W1 and the same rows of W3, a
// wave's two B fragments the same 32 rows of both -- gate and up of one (token, row) meet in one lane, and the epilogue stores
// swiglu(gate, up) (o1.swiglu(o3), transformer.cpp:481) instead of both: no [tokens][2 hidden] round trip, no k_swiglu_rows.
template <int EPI, int WT, int WR, int NB>
__global__ void __launch_bounds__(64 * WT * WR, 4) k_gemm_q8_mfma(const GemmArgs a) {
    using G = GemmTile<WT, WR, NB>;
    constexpr bool TWO = EPI == EPI_SWIGLU;
    static_assert(!TWO || NB == 2, "gate and up are the two B fragments of a wave");
    constexpr int TRH = TWO ? G::TR / 2 : G::TR;              // rows of a.rows one tile advances
    constexpr int SB = G::SB, LS = G::LS, TT = G::TT, TR = G::TR, NT = G::NT, kOffX = G::kOffX, kOffS = G::kOffS, kBuf = G::kBuf, NPW = G::NPW, NPX = G::NPX;
    extern __shared__ __attribute__((aligned(16))) char lds[]; char* const sm = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntt = (a.B + TT - 1) / TT;
    const int nb = gridDim.x, per = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7;              // tiles dealt to the XCDs in contiguous runs (token tile fastest: the tiles that share 64 weight rows are neighbours in the LOGICAL order, and workgroup b runs on XCD b mod 8 with its own L2: weight rows are fetched from HBM once per XCD instead of once per token tile)
    const int tile = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
    const int r0 = (tile / ntt) * TRH, b0 = (tile % ntt) * TT;
    const int sn = a.n / kGroup, nst = (sn + kGPS - 1) / kGPS;
    const unsigned rowbytes = (unsigned)a.n;
    constexpr unsigned kOOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, (int)((TWO ? 2u : 1u) * (unsigned)a.rows * rowbytes), 0x00020000);
    // tile row -> row of W (or kOOB-marker -1): SWIGLU tiles hold TRH rows of W1, then the same TRH rows of W3
    auto wrow = [&](int tr) -> int { const int within = TWO ? tr % TRH : tr, r = r0 + within; return r < a.rows ? (TWO ? (tr / TRH) * a.rows + r : r) : -1; };
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.Xq), 0, (int)((unsigned)a.B * rowbytes), 0x00020000);
    // loader: piece k of a thread = 16 bytes at (row (tid >> 3) + (NT / 8) k, chunk tid & 7); rows outside a matrix read as zero
    const int prow = tid >> 3, pch = tid & 7;
    unsigned woff[NPW], xoff[NPX];
#pragma unroll
    for (int k = 0; k < NPW; ++k) { const int r = wrow(prow + (NT / 8) * k); woff[k] = r >= 0 ? (unsigned)r * rowbytes + pch * 16 : kOOB; }
#pragma unroll
    for (int k = 0; k < NPX; ++k) xoff[k] = (b0 + prow + (NT / 8) * k < a.B) ? (unsigned)(b0 + prow + (NT / 8) * k) * rowbytes + pch * 16 : kOOB;
    const unsigned poff = (unsigned)(prow * LS + pch * 16);
    // scales: one per thread and stage, LDS slots [W g0 (TR)][W g1 (TR)][X g0 (TT)][X g1 (TT)]; (matrix, group) is the same for a whole
    // wave.  They come from GROUP-MAJOR copies of both scale arrays (64 rows' or tokens' scales of a group = one 256-byte run): gathered
    // from the row-major arrays -- a cache line per lane -- these loads made twice the line requests of the tile's data and 19 % of
    // the kernel (profiles/r02_prefill_gemm_pmc.txt, section 7).  A group past the end (odd group count) gets scale zero:
    // fma(0 * sx, float(d), acc) = acc (acc is never -0).
    const int s_slot = wave * 64;
    const bool s_x = s_slot >= kGPS * TR;
    const int s_gi = s_x ? (s_slot - kGPS * TR) / TT : s_slot / TR;
    const int s_row = s_x ? (b0 + (s_slot - kGPS * TR) % TT + lane < a.B ? b0 + (s_slot - kGPS * TR) % TT + lane : -1) : wrow(s_slot % TR + lane);
    const unsigned s_rows = s_x ? (unsigned)a.B : (TWO ? 2u : 1u) * (unsigned)a.rows;
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s_x ? a.XsT : a.sWT), 0, (int)(s_rows * sn * 4), 0x00020000);
    unsigned soff = s_row >= 0 ? ((unsigned)s_gi * s_rows + s_row) * 4 : kOOB;
    const unsigned sstep = kGPS * s_rows * 4;
    const unsigned spoff = (unsigned)(kOffS + tid * 4);
    v4u wr[NPW], xr[NPX]; unsigned sr;
    auto fetch = [&](int st) {                    // stage st -> the register slot (stages past the end: the offsets have run past the rows; never consumed with a non-zero scale)
#pragma unroll
        for (int k = 0; k < NPW; ++k) { wr[k] = __builtin_amdgcn_raw_buffer_load_b128(rW, (int)woff[k], 0, 0); woff[k] += SB; }
#pragma unroll
        for (int k = 0; k < NPX; ++k) { xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rX, (int)xoff[k], 0, 0); xoff[k] += SB; }
        sr = __builtin_amdgcn_raw_buffer_load_b32(rS, (int)(st * kGPS + s_gi < sn ? soff : kOOB), 0, 0); soff += sstep;
    };
    auto park = [&](int buf) {
        char* base = sm + buf * kBuf;
#pragma unroll
        for (int k = 0; k < NPW; ++k) *reinterpret_cast<v4u*>(base + poff + k * (NT / 8) * LS) = wr[k];
#pragma unroll
        for (int k = 0; k < NPX; ++k) *reinterpret_cast<v4u*>(base + kOffX + poff + k * (NT / 8) * LS) = xr[k];
        *reinterpret_cast<unsigned*>(base + spoff) = sr;
    };
    // this wave's 32 tokens x (NB x 32) weight rows
    // (SWIGLU: fragment j = rows wr0 .. wr0 + 31 of half j, i.e. tile rows j TRH + wr0 ..; otherwise fragment j = tile rows wr0 + 32 j ..)
    const int wt0 = (wave % WT) * 32, wr0 = (wave / WT) * 32 * (TWO ? 1 : NB), l31 = lane & 31, h = lane >> 5;
    constexpr int FS = TWO ? TRH : 32;                            // tile rows between a wave's fragments
    const unsigned offA = (unsigned)(kOffX + (wt0 + l31) * LS + h * 16), offB = (unsigned)((wr0 + l31) * LS + h * 16);
    const unsigned offsw = (unsigned)(kOffS + (wr0 + l31) * 4), offsx = (unsigned)(kOffS + (kGPS * TR + wt0 + 4 * h) * 4);
    float acc[NB][16];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    fetch(0); park(0); fetch(1);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const char* base = sm + (st & 1) * kBuf;
#pragma unroll
        for (int gi = 0; gi < kGPS; ++gi) {
            const v4i a0 = *reinterpret_cast<const v4i*>(base + offA + gi * kGroup), a1 = *reinterpret_cast<const v4i*>(base + offA + gi * kGroup + 32);
            float4 sx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sx[q] = *reinterpret_cast<const float4*>(base + offsx + gi * TT * 4 + q * 32);
            v16i d[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const v4i b0v = *reinterpret_cast<const v4i*>(base + offB + j * FS * LS + gi * kGroup), b1v = *reinterpret_cast<const v4i*>(base + offB + j * FS * LS + gi * kGroup + 32);
                const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                d[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0v, z, 0, 0, 0);
                d[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1v, d[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float sw = *reinterpret_cast<const float*>(base + offsw + (gi * TR + j * FS) * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[j][4 * q + 0] = __fmaf_rn(__fmul_rn(sw, sx[q].x), (float)d[j][4 * q + 0], acc[j][4 * q + 0]);   // quant_operators.cpp:274
                    acc[j][4 * q + 1] = __fmaf_rn(__fmul_rn(sw, sx[q].y), (float)d[j][4 * q + 1], acc[j][4 * q + 1]);
                    acc[j][4 * q + 2] = __fmaf_rn(__fmul_rn(sw, sx[q].z), (float)d[j][4 * q + 2], acc[j][4 * q + 2]);
                    acc[j][4 * q + 3] = __fmaf_rn(__fmul_rn(sw, sx[q].w), (float)d[j][4 * q + 3], acc[j][4 * q + 3]);
                }
            }
        }
        park((st & 1) ^ 1); fetch(st + 2);        // unconditional: a stage past the end is parked and never read with a non-zero scale
        __syncthreads();
    }
    gemm_epilogue<EPI, NB>(a, acc, r0 + wr0 + l31, b0 + wt0, lane);
}

// The int16 GEMM on the matrix cores.  gfx950 has no int16 MFMA; an int16 value is split EXACTLY into two signed bytes,
//     w = 256 * wh + wl,   wl = (int8)(w & 0xff),   wh = (w - wl) >> 8   (|w| <= 5792: wh in [-23, 23])
// so a quant group's dot product is  65536 * S(wh, xh) + 256 * (S(wh, xl) + S(wl, xh)) + S(wl, xl)  with four int8 MFMA
// accumulations S (each exact in int32; the combination wraps mod 2^32 on the way and lands on the true value, which the reference
// also holds in an int32: 64 * 5792^2 < 2^31, x86_simd.cpp:1524-1552).  Eight v_mfma_i32_32x32x32_i8 per group and wave replace 1024
// the fp32 chain step per group is unchanged (quant_operators.cpp:274).  The split happens
// once per 16-byte piece when it is parked in LDS (byte planes lo / hi per row).
__device__ __forceinline__ void split16(const v4i& v, unsigned (&lo)[2], unsigned (&hi)[2]) {
    const unsigned w0 = (unsigned)v.x, w1 = (unsigned)v.y, w2 = (unsigned)v.z, w3 = (unsigned)v.w;
    lo[0] = __builtin_amdgcn_perm(w1, w0, 0x06040200u); lo[1] = __builtin_amdgcn_perm(w3, w2, 0x06040200u);      // bytes 0, 2 of w0 then of w1
    const unsigned h0 = __builtin_amdgcn_perm(w1, w0, 0x07050301u), h1 = __builtin_amdgcn_perm(w3, w2, 0x07050301u);
    // wh = high byte + (low byte negative as int8 ? 1 : 0), bytewise without carries between bytes
    const unsigned c0 = (lo[0] >> 7) & 0x01010101u, c1 = (lo[1] >> 7) & 0x01010101u;
    hi[0] = ((h0 & 0x7f7f7f7fu) + c0) ^ (h0 & 0x80808080u);
    hi[1] = ((h1 & 0x7f7f7f7fu) + c1) ^ (h1 & 0x80808080u);
}
// Built like k_gemm_q8_mfma (64 tokens x 64 rows, four waves of 32 x 32, tokens as the A operand, results in VGPRs, one register
// slot of prefetch, one scale per thread) with a stage of ONE group (128 bytes of a row = two byte planes of 64): 42 KB of LDS,
// three workgroups per CU.  The four partial sums share ONE accumulator: d = S(hh); d <<= 8; d += S(hl) + S(lh); d <<= 8; d += S(ll)
// (the shifts between the MFMAs; everything mod 2^32) -- 16 registers instead of 48.
struct Gemm16Tile {
    static constexpr int TS = 64, LS = kGroup + 16;                           // tile side; LDS row stride of a byte plane
    static constexpr int kPlane = TS * LS, kOffS = 4 * kPlane, kBuf = kOffS + 2 * TS * 4;   // W lo | W hi | X lo | X hi | scales [W][X]
    static constexpr int kLds = 2 * kBuf;
};
template <int EPI>
__global__ void __launch_bounds__(256, 3) k_gemm_q16_mfma(const GemmArgs a) {
    using G = Gemm16Tile;
    constexpr int GB = kGroup * 2, LS = G::LS, TS = G::TS, kPlane = G::kPlane, kOffS = G::kOffS, kBuf = G::kBuf;
    extern __shared__ __attribute__((aligned(16))) char lds[]; char* const sm = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntt = (a.B + TS - 1) / TS;
    const int nb = gridDim.x, per = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7;              // tiles dealt to the XCDs in contiguous runs (token tile fastest: the tiles that share 64 weight rows are neighbours in the LOGICAL order, and workgroup b runs on XCD b mod 8 with its own L2: weight rows are fetched from HBM once per XCD instead of once per token tile)
    const int tile = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
    // EPI_SWIGLU: W = [W1 (gate) ; W3 (up)], a.rows rows each; a tile = 32 rows of W1 (the waves 0, 1) and the SAME 32 rows of W3 (the waves 2, 3): the up halves go
    // through LDS to the lanes that hold the gate halves, the epilogue is the SwiGLU (no [tokens][2 hidden] round trip, no k_swiglu_rows)
    constexpr bool TWO = EPI == EPI_SWIGLU;
    constexpr int TRW = TWO ? TS / 2 : TS;                                                     // rows of (each) matrix per tile
    const int r0 = (tile / ntt) * TRW, b0 = (tile % ntt) * TS;
    const int sn = a.n / kGroup;
    const unsigned rowbytes = (unsigned)a.n * 2;
    constexpr unsigned kOOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, (int)((unsigned)a.rows * (TWO ? 2u : 1u) * rowbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.Xq), 0, (int)((unsigned)a.B * rowbytes), 0x00020000);
    // loader: piece k of a thread = 16 bytes (8 elements) at (row (tid >> 3) + 32 k, chunk tid & 7) of both matrices
    const int prow = tid >> 3, pch = tid & 7;
    unsigned woff[2], xoff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if constexpr (TWO) woff[k] = (r0 + prow < a.rows) ? (unsigned)(k * a.rows + r0 + prow) * rowbytes + pch * 16 : kOOB;      // piece 0: W1's row, piece 1: W3's
        else woff[k] = (r0 + prow + 32 * k < a.rows) ? (unsigned)(r0 + prow + 32 * k) * rowbytes + pch * 16 : kOOB;
        xoff[k] = (b0 + prow + 32 * k < a.B)    ? (unsigned)(b0 + prow + 32 * k) * rowbytes + pch * 16 : kOOB;
    }
    const unsigned poff = (unsigned)(prow * LS + pch * 8);
    // scales: wave 0 = the weight rows', wave 1 = the tokens' (lane = row of the tile), from the group-major copies (see k_gemm_q8_mfma);
    // waves 2, 3 load nothing
    const bool s_x = wave == 1;
    const int s_rows = wave < 2 ? (s_x ? a.B : a.rows * (TWO ? 2 : 1)) : 0;
    int s_row = (s_x ? b0 : r0) + lane; bool s_ok = s_row < s_rows;
    if (TWO && !s_x) { s_row = (lane >> 5) * a.rows + r0 + (lane & 31); s_ok = r0 + (lane & 31) < a.rows; }      // (lanes 0..31: W1's rows, 32..63: W3's)
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s_x ? a.XsT : a.sWT), 0, (int)((unsigned)s_rows * sn * 4), 0x00020000);
    unsigned soff = s_ok ? (unsigned)s_row * 4 : kOOB;
    const unsigned spoff = (unsigned)(kOffS + (tid & 127) * 4);
    v4u wr[2], xr[2]; unsigned sr;
    auto fetch = [&](int g) {                     // group g -> the register slot (groups past the end: zero scale, never consumed otherwise)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            wr[k] = __builtin_amdgcn_raw_buffer_load_b128(rW, (int)woff[k], 0, 0); woff[k] += GB;
            xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rX, (int)xoff[k], 0, 0); xoff[k] += GB;
        }
        sr = __builtin_amdgcn_raw_buffer_load_b32(rS, (int)(g < sn ? soff : kOOB), 0, 0); soff += (unsigned)s_rows * 4;
    };
    auto park = [&](int buf) {
        char* base = sm + buf * kBuf;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            unsigned lo[2], hi[2];
            split16(__builtin_bit_cast(v4i, wr[k]), lo, hi);
            *reinterpret_cast<uint2*>(base + poff + k * 32 * LS) = make_uint2(lo[0], lo[1]); *reinterpret_cast<uint2*>(base + kPlane + poff + k * 32 * LS) = make_uint2(hi[0], hi[1]);
            split16(__builtin_bit_cast(v4i, xr[k]), lo, hi);
            *reinterpret_cast<uint2*>(base + 2 * kPlane + poff + k * 32 * LS) = make_uint2(lo[0], lo[1]); *reinterpret_cast<uint2*>(base + 3 * kPlane + poff + k * 32 * LS) = make_uint2(hi[0], hi[1]);
        }
        if (wave < 2) *reinterpret_cast<unsigned*>(base + spoff) = sr;
    };
    const int wt0 = (wave & 1) * 32, wr0 = (wave >> 1) * 32, l31 = lane & 31, h = lane >> 5;
    const unsigned offX = (unsigned)(2 * kPlane + (wt0 + l31) * LS + h * 16), offW = (unsigned)((wr0 + l31) * LS + h * 16);
    const unsigned offsw = (unsigned)(kOffS + (wr0 + l31) * 4), offsx = (unsigned)(kOffS + (TS + wt0 + 4 * h) * 4);
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    fetch(0); park(0); fetch(1);
    __syncthreads();
    for (int g = 0; g < sn; ++g) {
        const char* base = sm + (g & 1) * kBuf;
        const v4i xl0 = *reinterpret_cast<const v4i*>(base + offX), xl1 = *reinterpret_cast<const v4i*>(base + offX + 32);
        const v4i xh0 = *reinterpret_cast<const v4i*>(base + offX + kPlane), xh1 = *reinterpret_cast<const v4i*>(base + offX + kPlane + 32);
        const v4i wl0 = *reinterpret_cast<const v4i*>(base + offW), wl1 = *reinterpret_cast<const v4i*>(base + offW + 32);
        const v4i wh0 = *reinterpret_cast<const v4i*>(base + offW + kPlane), wh1 = *reinterpret_cast<const v4i*>(base + offW + kPlane + 32);
        const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(xh0, wh0, z, 0, 0, 0);
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(xh1, wh1, d, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = (int)((unsigned)d[i] << 8);
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(xh0, wl0, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(xh1, wl1, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(xl0, wh0, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(xl1, wh1, d, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = (int)((unsigned)d[i] << 8);
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(xl0, wl0, d, 0, 0, 0);                  // wraps mod 2^32 onto the exact int32 dot
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(xl1, wl1, d, 0, 0, 0);
        const float sw = *reinterpret_cast<const float*>(base + offsw);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 sx = *reinterpret_cast<const float4*>(base + offsx + q * 32);
            acc[4 * q + 0] = __fmaf_rn(__fmul_rn(sw, sx.x), (float)d[4 * q + 0], acc[4 * q + 0]);   // quant_operators.cpp:274
            acc[4 * q + 1] = __fmaf_rn(__fmul_rn(sw, sx.y), (float)d[4 * q + 1], acc[4 * q + 1]);
            acc[4 * q + 2] = __fmaf_rn(__fmul_rn(sw, sx.z), (float)d[4 * q + 2], acc[4 * q + 2]);
            acc[4 * q + 3] = __fmaf_rn(__fmul_rn(sw, sx.w), (float)d[4 * q + 3], acc[4 * q + 3]);
        }
        park((g & 1) ^ 1); fetch(g + 2);          // unconditional: a group past the end is parked and never read with a non-zero scale
        __syncthreads();
    }
    if constexpr (EPI == EPI_ROPE_KV) {
        float accw[1][16];
#pragma unroll
        for (int i = 0; i < 16; ++i) accw[0][i] = acc[i];
        gemm_epilogue<EPI_ROPE_KV, 1>(a, accw, r0 + wr0 + l31, b0 + wt0, lane);
        return;
    }
    if constexpr (TWO) {
        // the up halves (waves 2, 3) to the lanes that hold the gate halves of the same (row, tokens): through LDS (the loop's last barrier has passed: the stages are dead)
        float* xch = reinterpret_cast<float*>(sm);
        if (wave >= 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) xch[((wave & 1) * 64 + lane) * 17 + i] = acc[i];
        }
        __syncthreads();
        if (wave >= 2) return;
        const int row = r0 + l31;
        if (row < a.rows) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int b = b0 + wt0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (b < a.B) st_result_tp(a.out, (size_t)b * a.ldo + row, swiglu_elem(acc[i], xch[((wave & 1) * 64 + lane) * 17 + i]), a.out_peer, a.n_peer);   // o1.swiglu(o3) transformer.cpp:481
            }
        }
        return;
    }
    const int row = r0 + wr0 + l31;
    if (row < a.rows) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int b = b0 + wt0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (b >= a.B) continue;
            const size_t idx = (size_t)b * a.ldo + row;
            if constexpr (EPI == EPI_RESIDUAL) st_result_tp(a.out, idx, __fadd_rn(a.n_peer ? ld_agent(a.out + idx) : a.out[idx], acc[i]), a.out_peer, a.n_peer);   // (peers: tensor parallel, see GemmArgs)
            else a.out[idx] = acc[i];
        }
    }
}

// qkv[b] = [q ; k ; v] (dim each) of token b at position pos0 + b: RoPE on q and k (rope_v2 pairs), q -> qout[b], k / v -> cache rows
inline __global__ void k_rope_kv_rows(const float* qkv, float* qout, float* kcache, float* vcache, const float* rope_cos, const float* rope_sin,
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

// gu[b] = [gate ; up] (hidden each, this rank's slice) -> hd[b * ldo + i]; tensor parallel: hd = this rank's columns of every rank's [tokens][hidden_dim]
struct SwigluPeers { float* p[7]; int n; };
inline __global__ void k_swiglu_rows(float* hd, const float* gu, int hidden, int ldo, const SwigluPeers peers) {
    const float* g = gu + (size_t)blockIdx.x * 2 * hidden;
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) st_result_tp(hd, (size_t)blockIdx.x * ldo + i, swiglu_elem(g[i], g[hidden + i]), peers.p, peers.n);   // o1.swiglu(o3) transformer.cpp:481
}

} // namespace flm
