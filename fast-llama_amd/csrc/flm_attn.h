// flm_attn.h -- single-query attention (attn_head, k_attn_decode, k_attn_prefill), prompt attention on the fp32 matrix cores (k_qk_mfma,
// k_attn_pv_mfma) and on VALU chains (k_attn_prefill_mq), and the fused launches of the decode path (k_attn_o: attention + Wo; k_ffn: FFN13 + FFN2).
// Part of flm_kernels.h (hand-written gfx950 / CDNA4 kernels of the fast-llama per-token hot path); include that header.
#pragma once
#include "flm_math.h"
#include "flm_gemv.h"
// (bit-exactness hygiene: see flm_math.h -- no implicit FMA contraction in any of these headers)
#pragma clang fp contract(off)

namespace flm {

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
    int kv_rows;             // rows per head in the caches (0 = max_seq): the context pads the stride between two heads' rows -- at a power-of-two stride the heads' K / V streams share memory channels
    unsigned long long* trace;   // FLM_ABLATE builds: [head][8] s_memtime stamps
    // k_attn_o with head_size a multiple of 64: the head also leaves its output QUANTIZED (quant::quantize, quant_operators.cpp:26-47,
    // on the 64 values one wave holds) for the Wo GEMV that waits in the same launch: oq [heads*hs] int8 / int16, os [heads*hs/64]
    void* oq; float* os; int oqt;
    // long contexts: a head is spread over G workgroups of the same launch (attn_head, G > 1).  sc_global [heads][max_seq] carries
    // the scores between them, flag_sc one 64-byte line per (head, part) whose value reaches `epoch` when that part's scores
    // are in memory; err as in k_attn_o.
    int G; float* sc_global; unsigned* flag_sc; unsigned epoch; int* err;
    // tensor parallel, peer-to-peer: the head's output also goes to the same place of every peer rank's buffer (as GemvArgs::out_peer)
    float* out_peer[7]; int n_peer;
    // k_layers' granule hand-offs (flm_gemv.h: granule_t): non-null = the head's output leaves as {value, tag} granules (tag = the launch's flag value, attn_head's epoch_arg) into
    // every rank's granule vector (offset like `out`) instead of plain stores -- the Wo workgroups sweep them, nobody drains stores or raises a line for them
    unsigned long long* gout; unsigned long long* gout_peer[7];
    // ... and its INPUT: q [heads * hs] and this token's K / V cache row [heads * hs] as granules written by the QKV phase of the same launch (attn_head's gr_in: the head sweeps
    // its own pieces instead of polling the QKV workgroups' lines and reading behind them; the cache rows themselves are written for the NEXT tokens and not waited for)
    const unsigned long long* qg; const unsigned long long* kg; const unsigned long long* vg;
};

#ifndef FLM_V_LATE_NS
#define FLM_V_LATE_NS 1
#endif
#ifndef FLM_SPLIT_V_LATE
#define FLM_SPLIT_V_LATE 1
#endif
constexpr int kAttnBlock = 1024;      // 16 waves
constexpr int kAttnTile = 64;         // positions per LDS tile
constexpr int kAttnDepth = 2;         // tiles in flight per stream (K, V), register rings: both streams start when the kernel does
// LDS row stride of a tile in floats: compile-time per instantiation (64*NF + 8), so that the chains' LDS reads use
// immediate offsets; +8: the 8x8 (position, accumulator) score lanes and the PV lanes hit distinct banks
__host__ __device__ inline int attn_row_stride(int hs) { const int nf = hs <= 64 ? 1 : hs <= 128 ? 2 : 4; return nf * 64 + 8; }
__host__ inline size_t attn_lds_bytes(int max_seq, int hs, bool split = false) {
    size_t tiles = (size_t)(2 * kAttnTile + 12) * attn_row_stride(hs);                      // + 12 slack rows: the PV read-ahead
    const size_t vslice = (size_t)(1024 + 4) * 32 + 64;                                       // split heads: the part's whole V slice, transposed [32][kSplitMaxSeq + 4], lies where the K tiles were
    if (split && vslice > tiles) tiles = vslice;
    return (size_t)(hs + 32 + 64 + ((max_seq + 3) & ~3) + 64 + tiles) * 4;       // q, red, the exp table, scores, tiles
}

// NF = 16-byte pieces of a tile per thread = ceil(hs / 64)
// COH: K/V/q were (partly) written by other workgroups of the SAME kernel (a fused launch) -> coherent sc0|sc1 loads; the
// per-phase kernels read them after a kernel boundary and use ordinary cached loads
// G > 1 (long contexts; all G workgroups of a head run in the SAME launch): part g computes the scores of its share of the
// position tiles (K traffic / G), the parts exchange scores through memory (flag lines, as k_attn_o's hand-off), every part
// then runs the softmax over all T scores -- the same operations in the same order, so the same bits in every part -- and
// the weighted V sum of ITS hs / G output dimensions over all positions (V traffic / G; t ascending per dimension: the
// reference's chain).  A head's 2 T hs 4 bytes then flow through G CUs instead of one.
// SPLIT (the G > 1 instantiation; nd == kSplitDims, max_seq <= kSplitMaxSeq): streaming is latency-bound -- a tile costs a memory
// round trip unless it was requested long before it is needed -- so the K ring is 4 tiles deep (a part's whole share up to T = 1024)
// and the part's ENTIRE V slice (T x 32 floats <= 128 KiB) is requested when the kernel starts, waits in registers (8 x 16 bytes per
// thread) under the scores and the softmax, and is then parked in LDS where the K tiles were: the weighted sum runs without a barrier.
constexpr int kSplitDims = 32, kSplitMaxSeq = 1024, kSplitVRegs = kSplitMaxSeq * (kSplitDims / 4) / 1024;
// PRE (the whole-layer launches: q and the cache rows of THIS token come from workgroups of the same launch, behind a flag round = mid()): the rows of earlier tokens are
// requested in FRONT of mid() and land while the lines are polled; behind it come q and the one new row of K and of V, patched into the ring registers of the
// thread whose piece it is.  The arithmetic is untouched.
struct AttnNoMid { __device__ __forceinline__ void operator()() const {} };
template <int NF, bool COH, bool SPLIT = false, bool PRE = false, class Mid = AttnNoMid, bool GRIN = false>
__device__ __forceinline__ void attn_head(const AttnArgs& a, const int h, char* lds, const int T, const float* qrow, float* orow, const int g = 0, const int G = 1, Mid&& mid = Mid(), const unsigned epoch_arg = 0u, const bool gr_out = false, const bool gr_sc = false, const float* kpre = nullptr) {
    const unsigned xepoch = epoch_arg ? epoch_arg : a.epoch;      // what the parts of a split head raise / wait for in their score exchange
    typedef float v4f __attribute__((ext_vector_type(4)));
    constexpr int D = SPLIT ? 4 : kAttnDepth;         // K ring depth
    constexpr int DV = SPLIT ? 1 : kAttnDepth;        // V ring depth (SPLIT: unused, the slice sits in vall)
    const int hs = a.hs, tid = threadIdx.x;
#ifdef FLM_TRACE_PRO_RT
    auto stamp = [&](int k) { if (kAblate && a.trace && threadIdx.x == 0) a.trace[(h * G + g) * 16 + k] = __builtin_amdgcn_s_memrealtime(); };   // (tools/trace_back.py: the 100 MHz clock of the launch's other stamps)
#else
    auto stamp = [&](int k) { if (kAblate && a.trace && threadIdx.x == 0) a.trace[(h * G + g) * 8 + k] = __builtin_amdgcn_s_memtime(); };
#endif
    stamp(0);
    constexpr int rs = NF * 64 + 8;
    const int f4r = hs >> 2, tile_f4 = kAttnTile * f4r;
    float* qs   = reinterpret_cast<float*>(lds);                 // [hs]
    float* red  = qs + hs;                                       // 32
    unsigned long long* etab = reinterpret_cast<unsigned long long*>(red + 32);   // kExp2fTab's 32 entries: the softmax's lookups stay on the CU
    float* sc   = red + 32 + 64;                                 // [T] scores -> probabilities (+ slack for the sum ring's read-ahead)
    float* tile0 = sc + ((a.max_seq + 3) & ~3) + 64;
    float* tile1 = tile0 + kAttnTile * rs;
    const int lane = tid & 63, wave = tid >> 6;
    const float* K = a.kcache + (size_t)h * (a.kv_rows ? a.kv_rows : a.max_seq) * hs;
    const float* V = a.vcache + (size_t)h * (a.kv_rows ? a.kv_rows : a.max_seq) * hs;
    // K/V rows of this token may have been written by other workgroups of the same kernel (a fused launch): coherent loads
    // (sc0|sc1) through buffer descriptors; positions past T get an out-of-range offset and read as zero
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K), 0, a.max_seq * hs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V), 0, a.max_seq * hs * 4, 0x00020000);
    const float scale = (float)(1.0 / (double)__builtin_sqrtf((float)hs));   // attn_scale, transformer.cpp:418
    const int nt = (T + kAttnTile - 1) / kAttnTile;
    // this part's share: score tiles [sb, se), output dimensions [d0, d0 + nd)
    const int tpp = (nt + G - 1) / G, sb = g * tpp, se = (sb + tpp < nt) ? sb + tpp : nt;
    const int nd = hs / G, d0 = g * nd, f4v = nd >> 2, tile_f4v = kAttnTile * f4v;
    // Two streams of tiles (K for the scores, V for the weighted sum), each through a ring of D register sets; both are
    // requested when the kernel starts (the V tiles arrive under the softmax), tile i+D when tile i has been parked.
    v4f ringK[D][NF], ringV[DV][NF];
    v4f vall[SPLIT ? kSplitVRegs : 1];
    // this thread's pieces of a tile: row / byte offsets computed once (an integer division per piece per tile would
    // cost more than the tile's arithmetic).  K: whole rows; V: the part's nd columns of every row.
    int prow[NF], goff[NF], loff[NF], prowv[NF], goffv[NF], loffv[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int f = tid + j * kAttnBlock, row = f / f4r, c4 = f - row * f4r;
        prow[j] = f < tile_f4 ? row : (1 << 28);                    // pieces past the tile never pass the t < T test
        goff[j] = (row * hs + c4 * 4) * 4;
        loff[j] = row * rs + c4 * 4;
        const int rowv = f / f4v, c4v = f - rowv * f4v;
        prowv[j] = f < tile_f4v ? rowv : (1 << 28);
        goffv[j] = (rowv * hs + d0 + c4v * 4) * 4;
        loffv[j] = rowv * rs + c4v * 4;
    }
    const int tile_bytes = kAttnTile * hs * 4;
    const int Told = PRE ? T - 1 : T;                               // rows below Told were written by earlier launches
    auto request = [&](const __amdgpu_buffer_rsrc_t& r, int tile, int tile_end, const int (&pr)[NF], const int (&go)[NF], v4f (&reg)[NF], const int bound) {
        const int t0 = tile * kAttnTile;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const unsigned off = (tile < tile_end && t0 + pr[j] < bound) ? (unsigned)(tile * tile_bytes + go[j]) : 0x80000000u;
            reg[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, COH ? kAuxCoherent : 0));
        }
    };
    // PRE: the piece of row T - 1 (this token's) of an already requested tile, behind the flag round
    auto patch = [&](const __amdgpu_buffer_rsrc_t& r, int tile, int tile_end, const int (&pr)[NF], const int (&go)[NF], v4f (&reg)[NF]) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
            if (tile < tile_end && tile * kAttnTile + pr[j] == T - 1)
                reg[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(unsigned)(tile * tile_bytes + go[j]), 0, kAuxCoherent));
    };
    auto park = [&](float* buf, const int (&pr)[NF], const int (&lo)[NF], const v4f (&reg)[NF]) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
            if (pr[j] < kAttnTile) *reinterpret_cast<float4*>(buf + lo[j]) = make_float4(reg[j].x, reg[j].y, reg[j].z, reg[j].w);
    };
    // q first: loads return in issue order, and q requested behind the K and V tiles would arrive behind 4 tiles of data
    // (PRE: q does not exist yet -- the earlier tokens' rows now, q behind the flag round)
    unsigned etv = 0;
    if (tid < 64) etv = reinterpret_cast<const unsigned*>(kExp2fTab)[tid];   // (requested first: it returns first)
    float qv[NF * 64 / kAttnBlock + 1];
    if constexpr (!PRE) {
#pragma unroll
        for (int i = 0; i < NF * 64 / kAttnBlock + 1; ++i) { const int d = tid + i * kAttnBlock; qv[i] = d < hs ? (COH ? ld_agent(qrow + (size_t)h * hs + d) : qrow[(size_t)h * hs + d]) : 0.f; }
    }
    // (SPLIT + PRE: only the first FLM_SPLIT_K_EARLY tiles of the part in front of the q flag round -- the round's looks return behind whatever the wave requested before them, and
    //  the later tiles are not needed before the first step's scores are done: they are requested behind the round, with this token's row in them)
#ifndef FLM_SPLIT_K_EARLY
#define FLM_SPLIT_K_EARLY 2
#endif
    constexpr int KE = (SPLIT && PRE) ? (FLM_SPLIT_K_EARLY < D ? FLM_SPLIT_K_EARLY : D) : D;
    // kpre (SPLIT + PRE, k_layers): the part's first two K tiles were brought into LDS by LDS-DMA UNDER the layer's QKV phase (attn_kpre_issue: rows of earlier tokens depend on nothing
    // of the layer), in the tile layout at kpre / kpre + one tile: nothing of them is requested here, the first step's scores start when q is there
    // (FLM_KPRE_EARLY3: with the first two tiles pre-landed by DMA, the part's OTHER two tiles go into the ring's registers in front of the round, the whole share of a part up to
    //  T = 1024 -- a part of three or four tiles (from position 513 on) otherwise waits a whole round trip for them behind the first step's scores.  us per 32-layer token at positions
    //  300 / 516 / 600 / 900, same box: 0 tiles 1772 / 1927 / 1965 / 2133, 1 tile 1777 / 1915 / 1957 / 2127, 2 tiles 1765 / 1906 / 1947 / 2112)
#ifndef FLM_KPRE_EARLY3
#define FLM_KPRE_EARLY3 2
#endif
    constexpr int KE3 = (SPLIT && PRE) ? (KE + FLM_KPRE_EARLY3 < D ? FLM_KPRE_EARLY3 : D - KE) : 0;
    // (SPLIT + GRIN: q and this token's K / V row come as granules, swept below -- everything of the earlier tokens' rows is requested in FRONT of the sweep, whose loads return
    //  behind it: a tile requested behind the sweep and patched with a select would be waited for at the select)
    constexpr bool swept_s = PRE && SPLIT && GRIN;
    static_assert(!swept_s || KE + KE3 == D, "SPLIT + GRIN: the whole ring in front of the sweep");
    if (swept_s && kpre == nullptr) {
#pragma unroll
        for (int u = 0; u < D; ++u) request(rK, sb + u, se, prow, goff, ringK[u], Told);
    } else
    if (!(SPLIT && PRE) || kpre == nullptr) {
#pragma unroll
        for (int u = 0; u < KE; ++u) request(rK, sb + u, se, prow, goff, ringK[u], Told);
    } else {
#pragma unroll
        for (int u = KE; u < KE + KE3; ++u) request(rK, sb + u, se, prow, goff, ringK[u], Told);
    }
    if constexpr (SPLIT) {
        // a thread's pieces: the 16-byte column tid % 8 of the rows 4 (tid / 8) + (j % 4) + 512 (j / 4) -- four CONSECUTIVE rows per half, so
        // that the transposed parking below writes four positions of a dimension with one 16-byte store.  The first half of the slice
        // (rows < 512) now, the second half when the K ring's registers are free (behind the scores; it lands under the exchange and
        // the softmax): all of it at once is 4 registers more than a 1024-thread workgroup has
        // (PRE: behind the flag round -- V is not needed before the weighted sum, and in front of the round it is half of what the round's looks queue behind)
        if constexpr (!(PRE && FLM_SPLIT_V_LATE)) {
#pragma unroll
            for (int j = 0; j < kSplitVRegs / 2; ++j) {
                const int row = 4 * (tid >> 3) + (j & 3) + (kSplitMaxSeq / 2) * (j >> 2);
                const unsigned off = row < Told ? (unsigned)((row * hs + d0 + (tid & 7) * 4) * 4) : 0x80000000u;
                vall[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rV, (int)off, 0, COH ? kAuxCoherent : 0));
            }
        }
    } else {
        if constexpr (!(PRE && FLM_V_LATE_NS)) {
#pragma unroll
            for (int u = 0; u < DV; ++u) request(rV, u, nt, prowv, goffv, ringV[u], Told);
        }
    }
    constexpr bool swept = PRE && !SPLIT && GRIN;                  // (GRIN: k_layers' one-launch token, one workgroup per head)
    if constexpr (swept) {
        {
            // q and this token's K / V row as granules (tag epoch_arg) from the QKV phase of this launch: every thread re-reads ITS pieces until their tags match -- q element
            // tid, and at most one 16-byte piece of the K tile and of the V tile that hold row T - 1 (T <= 2 tiles: everything was requested above).  No line round, no read behind it.
            typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(a.qg) + (size_t)h * hs, 0, hs * 8, 0x00020000);
            const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(a.kg) + (size_t)h * hs, 0, hs * 8, 0x00020000);
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(a.vg) + (size_t)h * hs, 0, hs * 8, 0x00020000);
            int kcol = -1, vcol = -1;                                   // first column of the thread's piece of row T - 1 (none: -1)
#pragma unroll
            for (int u = 0; u < D; ++u)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    if (sb + u < se && (sb + u) * kAttnTile + prow[j] == T - 1) kcol = (goff[j] >> 2) % hs;
                    if (u < DV && u < nt && u * kAttnTile + prowv[j] == T - 1) vcol = (goffv[j] >> 2) % hs;
                }
            unsigned qbits = 0; v4u32 KA = {0u, 0u, 0u, 0u}, KC = KA, VA = KA, VC = KA;
            bool okq = tid >= hs, okk = kcol < 0, okv = vcol < 0;
            const int gave_up = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (true) {
                asm volatile("" ::: "memory");
                uint2 qq = make_uint2(0u, 0u);
                if (!okq) qq = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rq, tid * 8, 0, kAuxCoherent));
                if (!okk) { KA = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rk, kcol * 8, 0, kAuxCoherent)); KC = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rk, kcol * 8 + 16, 0, kAuxCoherent)); }
                if (!okv) { VA = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rv, vcol * 8, 0, kAuxCoherent)); VC = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rv, vcol * 8 + 16, 0, kAuxCoherent)); }
                if (!okq) { okq = qq.y == epoch_arg; qbits = qq.x; }
                if (!okk) okk = KA.y == epoch_arg && KA.w == epoch_arg && KC.y == epoch_arg && KC.w == epoch_arg;
                if (!okv) okv = VA.y == epoch_arg && VA.w == epoch_arg && VC.y == epoch_arg && VC.w == epoch_arg;
                if (__all(okq && okk && okv) || gave_up) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
            stamp(7);
            qv[0] = tid < hs ? __uint_as_float(qbits) : 0.f;
            const v4f kp = {__uint_as_float(KA.x), __uint_as_float(KA.z), __uint_as_float(KC.x), __uint_as_float(KC.z)};
            const v4f vp = {__uint_as_float(VA.x), __uint_as_float(VA.z), __uint_as_float(VC.x), __uint_as_float(VC.z)};
#pragma unroll
            for (int u = 0; u < D; ++u)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    if (sb + u < se && (sb + u) * kAttnTile + prow[j] == T - 1) ringK[u][j] = kp;
            // the V tiles behind the sweep (as behind the flag round before): the earlier rows from the cache, the piece of row T - 1 from its granules
#pragma unroll
            for (int u = 0; u < DV; ++u)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    if (u < nt && u * kAttnTile + prowv[j] == T - 1) ringV[u][j] = vp;
                    else {
                        const unsigned off = (u < nt && u * kAttnTile + prowv[j] < Told) ? (unsigned)(u * tile_bytes + goffv[j]) : 0x80000000u;
                        ringV[u][j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rV, (int)off, 0, COH ? kAuxCoherent : 0));
                    }
                }
        }
    }
    v4f vnew = {0.f, 0.f, 0.f, 0.f}, knew = vnew;                   // SPLIT + GRIN: this thread's piece of this token's V / K row (if it owns one)
#ifndef FLM_SPLIT_V_BEHIND
#define FLM_SPLIT_V_BEHIND 0      /* measured neutral (positions 516 / 600 / 900: 1888 / 1930 / 2097 us per token either way): the slice stays behind the sweep */
#endif
    const bool v_behind = FLM_SPLIT_V_BEHIND && SPLIT && PRE && GRIN && (kpre == nullptr || se - sb > KE);          // (workgroup-uniform)
    auto vfirst = [&]() {
        if constexpr (SPLIT) {
#pragma unroll
            for (int j = 0; j < kSplitVRegs / 2; ++j) {
                const int row = 4 * (tid >> 3) + (j & 3) + (kSplitMaxSeq / 2) * (j >> 2);
                if (row == T - 1) vall[j] = vnew;
                else {
                    const unsigned off = row < Told ? (unsigned)((row * hs + d0 + (tid & 7) * 4) * 4) : 0x80000000u;
                    vall[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rV, (int)off, 0, COH ? kAuxCoherent : 0));
                }
            }
        }
    };
    const int tlast = (T - 1) / kAttnTile;
    auto patch_lds = [&](float* buf, int tile) {                    // this token's K piece into tile `tile` parked at buf
#pragma unroll
        for (int j = 0; j < NF; ++j)
            if (tile == tlast && tile < se && tile * kAttnTile + prow[j] == T - 1) *reinterpret_cast<float4*>(buf + loff[j]) = make_float4(knew.x, knew.y, knew.z, knew.w);
    };
    if constexpr (swept_s) {
        // the split head's part: q element tid, the one 16-byte piece of this token's K row if it lies in one of the part's tiles, and the piece of this token's V row in the part's
        // 32 dimensions (8 threads own them) -- as granules (tag epoch_arg) from the QKV phase of this launch, every thread re-reading ITS pieces until their tags match
        static_assert(!swept_s || FLM_SPLIT_V_LATE, "the V slice is requested behind the sweep");
        typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(a.qg) + (size_t)h * hs, 0, hs * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(a.kg) + (size_t)h * hs, 0, hs * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(a.vg) + (size_t)h * hs, 0, hs * 8, 0x00020000);
        int kcol = -1;
#pragma unroll
        for (int j = 0; j < NF; ++j)
            if (tlast >= sb && tlast < se && tlast * kAttnTile + prow[j] == T - 1) kcol = (goff[j] >> 2) % hs;
        const bool vown = (((T - 1) & (kSplitMaxSeq / 2 - 1)) >> 2) == (tid >> 3);     // rows 4 (tid / 8) + (j % 4) + 512 (j / 4) of the slice are this thread's
        const int vcol = d0 + (tid & 7) * 4;
        unsigned qbits = 0; v4u32 KA = {0u, 0u, 0u, 0u}, KC = KA, VA = KA, VC = KA;
        bool okq = tid >= hs, okk = kcol < 0, okv = !vown;
        const int gave_up = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (true) {
            asm volatile("" ::: "memory");
            uint2 qq = make_uint2(0u, 0u);
            if (!okq) qq = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rq, tid * 8, 0, kAuxCoherent));
            if (!okk) { KA = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rk, kcol * 8, 0, kAuxCoherent)); KC = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rk, kcol * 8 + 16, 0, kAuxCoherent)); }
            if (!okv) { VA = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rv, vcol * 8, 0, kAuxCoherent)); VC = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rv, vcol * 8 + 16, 0, kAuxCoherent)); }
            if (!okq) { okq = qq.y == epoch_arg; qbits = qq.x; }
            if (!okk) okk = KA.y == epoch_arg && KA.w == epoch_arg && KC.y == epoch_arg && KC.w == epoch_arg;
            if (!okv) okv = VA.y == epoch_arg && VA.w == epoch_arg && VC.y == epoch_arg && VC.w == epoch_arg;
            if (__all(okq && okk && okv) || gave_up) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        }
        // every wave: what it requested before the sweep has landed -- the pre-landed tiles' DMA among it (the QKV phase of the granule form ends without a drain; the first
        // step's barrier follows) -- loads return in order, so this waits for nothing a sweeping wave has not already seen
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(7);
        qv[0] = tid < hs ? __uint_as_float(qbits) : 0.f;
        knew = v4f{__uint_as_float(KA.x), __uint_as_float(KA.z), __uint_as_float(KC.x), __uint_as_float(KC.z)};
        vnew = v4f{__uint_as_float(VA.x), __uint_as_float(VA.z), __uint_as_float(VC.x), __uint_as_float(VC.z)};
        // (this token's K piece goes into its tile IN LDS: into a pre-landed tile now, into a ring tile when it is parked -- whichever tile of the part it is)
        if (kpre != nullptr) {
#pragma unroll
            for (int u = 0; u < KE; ++u) patch_lds(const_cast<float*>(kpre) + u * kAttnTile * rs, sb + u);
        }
        // the first half of the V slice behind the sweep (the earlier tokens' rows from the cache, this token's piece from its granules) -- where the part's scores come from the
        // pre-landed tiles alone.  A part that PARKS ring tiles gets it behind its scores: the parks sit in a loop, the compiler merges the loop's entry with its back edge and waits
        // there for (nearly) everything in flight -- a slice requested here would be waited for before the part's second step (measured: 3.9 us from the sweep to the last score)
        if (!v_behind) vfirst();
    }
    if constexpr (PRE && !swept && !swept_s) {
        mid();                                                      // the flag round: q and this token's cache rows are in memory
        stamp(7);
#pragma unroll
        for (int i = 0; i < NF * 64 / kAttnBlock + 1; ++i) { const int d = tid + i * kAttnBlock; qv[i] = d < hs ? ld_agent(qrow + (size_t)h * hs + d) : 0.f; }
        if (SPLIT && kpre != nullptr) {
            // this token's row, if it lies in one of the two pre-landed tiles: straight into its place in LDS (the first step's barrier follows)
#pragma unroll
            for (int u = 0; u < KE; ++u)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    if (sb + u < se && (sb + u) * kAttnTile + prow[j] == T - 1) {
                        const v4f t4 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rK, (int)(unsigned)((sb + u) * tile_bytes + goff[j]), 0, kAuxCoherent));
                        *reinterpret_cast<float4*>(const_cast<float*>(kpre) + u * kAttnTile * rs + loff[j]) = make_float4(t4.x, t4.y, t4.z, t4.w);
                    }
        } else {
#pragma unroll
        for (int u = 0; u < KE; ++u) patch(rK, sb + u, se, prow, goff, ringK[u]);
        }
#pragma unroll
        for (int u = KE; u < D; ++u) {
            if (u < KE + KE3 && SPLIT && kpre != nullptr) patch(rK, sb + u, se, prow, goff, ringK[u]);
            else request(rK, sb + u, se, prow, goff, ringK[u], T);
        }
        if constexpr (SPLIT) {
#pragma unroll
            for (int j = 0; j < kSplitVRegs / 2; ++j) {
                const int row = 4 * (tid >> 3) + (j & 3) + (kSplitMaxSeq / 2) * (j >> 2);
                if constexpr (FLM_SPLIT_V_LATE) {
                    const unsigned off = row < T ? (unsigned)((row * hs + d0 + (tid & 7) * 4) * 4) : 0x80000000u;
                    vall[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rV, (int)off, 0, kAuxCoherent));
                } else if (row == T - 1) vall[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rV, (int)(unsigned)((row * hs + d0 + (tid & 7) * 4) * 4), 0, kAuxCoherent));
            }
        } else if constexpr (FLM_V_LATE_NS) {
#pragma unroll
            for (int u = 0; u < DV; ++u) request(rV, u, nt, prowv, goffv, ringV[u], T);
        } else {
#pragma unroll
            for (int u = 0; u < DV; ++u) patch(rV, u, nt, prowv, goffv, ringV[u]);
        }
    }
#pragma unroll
    for (int i = 0; i < NF * 64 / kAttnBlock + 1; ++i) { const int d = tid + i * kAttnBlock; if (d < hs) qs[d] = qv[i]; }
    if (tid < 64) reinterpret_cast<unsigned*>(etab)[tid] = etv;

    // ---- scores: lane = (position p, strided accumulator k) -- the 8 lanes of dot_product_avx256; each lane's
    //      chain is i ascending, then the 8 partials are added 0..7 (lane k = 0 collects them with DPP row shifts).
    // (no early exit from the unrolled tile loops: with a break inside them the compiler loses count of the loads in flight and
    //  waits for nearly all of them -- the V tiles included -- before the first K tile is parked.  A tile past the end is zeros.)
    float lmax = -INFINITY;
    // one (position, accumulator) lane of a tile that lies in LDS at `cur`
    auto score_lane = [&](const float* cur, int s, int p, int k) {
        const int t = s * kAttnTile + p;
        const float* kp = cur + p * rs + k;
        float l = 0.f;
#pragma unroll 8
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
            if (G > 1) {
                if (gr_sc) __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.sc_global) + (size_t)h * a.max_seq + t, ((unsigned long long)xepoch << 32) | (unsigned long long)__float_as_uint(sv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else st_agent(a.sc_global + (size_t)h * a.max_seq + t, sv);
            }
        }
    };
    if constexpr (SPLIT) {
        // two tiles per step: 128 positions x 8 accumulators keep all 16 waves busy; two barriers per step (park -> compute -> next park)
        for (int base = sb; base < se; base += D) {
#pragma unroll
            for (int u = 0; u < D; u += 2) {
                const int s = base + u;
                const bool pre = PRE && u == 0 && base == sb && kpre != nullptr;   // (wave-uniform: the two pre-landed tiles)
                if (!pre) {
                    park(tile0, prow, loff, ringK[u]); park(tile1, prow, loff, ringK[u + 1]);
                    if constexpr (swept_s) { patch_lds(tile0, s); patch_lds(tile1, s + 1); }
                }
                if (base == sb) stamp(u == 0 ? 10 : 13);
                __syncthreads();
                if (base == sb) stamp(u == 0 ? 5 : 14);
                request(rK, s + D, se, prow, goff, ringK[u], swept_s ? Told : T); request(rK, s + D + 1, se, prow, goff, ringK[u + 1], swept_s ? Told : T);
                const int half = tid >> 9, s2 = s + half;
                const float* c0 = pre ? kpre : tile0; const float* c1 = pre ? kpre + kAttnTile * rs : tile1;
                if (s2 < se) score_lane(half ? c1 : c0, s2, (tid >> 3) & 63, tid & 7);
                if (base == sb) stamp(u == 0 ? 11 : 15);
                __syncthreads();
                if (base == sb && u == 2) stamp(9);
            }
        }
    } else {
        // two tiles per step, one per half of the workgroup (contexts of 65 .. 128 positions: one step instead of two tile times)
        static_assert(D == 2, "one step = the ring's two tiles");
        for (int s = sb; s < se; s += 2) {
            park(tile0, prow, loff, ringK[0]); park(tile1, prow, loff, ringK[1]);
            __syncthreads();
            if (s == sb) stamp(5);
            request(rK, s + 2, se, prow, goff, ringK[0], T); request(rK, s + 3, se, prow, goff, ringK[1], T);
            const int half = tid >> 9, s2 = s + half;
            if (s2 < se) score_lane(half ? tile1 : tile0, s2, (tid >> 3) & 63, tid & 7);
            __syncthreads();
        }
    }
    if constexpr (SPLIT) {
        if (v_behind) vfirst();
#pragma unroll
        for (int j = kSplitVRegs / 2; j < kSplitVRegs; ++j) {
            const int row = 4 * (tid >> 3) + (j & 3) + (kSplitMaxSeq / 2) * (j >> 2);
            if (swept_s && row == T - 1) vall[j] = vnew;           // (granule form: this token's row is not waited for in the cache)
            else {
            const unsigned off = row < (swept_s ? Told : T) ? (unsigned)((row * hs + d0 + (tid & 7) * 4) * 4) : 0x80000000u;
            vall[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rV, (int)off, 0, COH ? kAuxCoherent : 0));
            }
        }
    }
    stamp(1);
    if (G > 1 && gr_sc) {
        // exchange on granules (k_layers' granule hand-offs; sc_global holds 8 bytes per score): the parts' scores are {score, tag = this layer's flag value} stores, every thread
        // re-reads the other parts' scores it needs until their tags match -- no drained stores, no line per part, no second round trip
        __syncthreads();                                            // (this part's own scores are in sc[])
        lmax = -INFINITY;
        const unsigned long long* sg = reinterpret_cast<const unsigned long long*>(a.sc_global) + (size_t)h * a.max_seq;
        const int gave_up = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int t0 = 0; t0 < T; t0 += kAttnBlock) {                // (T <= kSplitMaxSeq = one round; wave-uniform trip count)
            const int t = t0 + tid, tl = t / kAttnTile;
            const bool mine = t >= T || (tl >= sb && tl < se);
            bool ok = mine; unsigned bits = 0;
            const unsigned long long ts = __builtin_amdgcn_s_memrealtime();
            while (true) {
                asm volatile("" ::: "memory");
                if (!ok) { const unsigned long long gv = __hip_atomic_load(sg + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = (unsigned)(gv >> 32) == xepoch; bits = (unsigned)gv; }
                if (__all(ok) || gave_up) break;
                if (__builtin_amdgcn_s_memrealtime() - ts > 2000000ull) { __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            if (t < T) {
                const float sv = mine ? sc[t] : __uint_as_float(bits);
                sc[t] = sv;
                lmax = fmaxf(lmax, sv);
            }
        }
    } else
    if (G > 1) {
        // exchange: my scores are in memory -> raise my line; wait for the other parts' lines; fetch their scores
        wait_stores_done();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.flag_sc + (h * G + g) * 16, xepoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 64) {
            const bool mine = tid < G;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (true) {
                const unsigned f = mine ? __hip_atomic_load(a.flag_sc + (h * G + tid) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : xepoch;
                if (__all((int)(f - xepoch) >= 0)) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __syncthreads();
        lmax = -INFINITY;
        for (int t = tid; t < T; t += kAttnBlock) {
            const int tl = t / kAttnTile;
            const float sv = (tl >= sb && tl < se) ? sc[t] : ld_agent(a.sc_global + (size_t)h * a.max_seq + t);
            sc[t] = sv;
            lmax = fmaxf(lmax, sv);
        }
    }
    // Short contexts (one tile, one part): the whole softmax in the registers of ONE wave, lane t = position t -- no trip through LDS
    // and no workgroup barrier between max, exp, sum and divide (three barriers and ~2.5 us of a 7 us head at T = 21).  The same
    // operations in the same order: max order-free; expf_ref; the sum t ascending as 63 dependent adds along the lanes (step k:
    // every lane adds its own term to its left neighbour's running sum, v_add with DPP wave_shr:1; after step k lanes 0..k hold
    // their exact prefix, so lane T - 1 ends with sum_{t < T} in the reference's order); divide; skip rule.
    // (SPLIT: the part's V slice is parked -- transposed, where the K tiles were: every wave is behind the scores' last barrier -- by the 15 waves that only wait for the sum; wave 0
    //  parks its pieces behind its chain.  The layout is described at the weighted sum below.)
#ifndef FLM_SPLIT_PARK_EARLY
#define FLM_SPLIT_PARK_EARLY 1
#endif
    auto park_v = [&]() {
        if constexpr (SPLIT) {
            constexpr int VS = kSplitMaxSeq + 4;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {                            // 4 rows x 4 dimensions in registers -> 4 dimensions x 4 positions: four 16-byte stores
                const int p0 = 4 * (tid >> 3) + (kSplitMaxSeq / 2) * hh;     // (rows past T were never loaded: zeros)
                if (p0 < T) {
                    float* dst = tile0 + (tid & 7) * 4 * VS + p0;
                    const v4f r0 = vall[4 * hh], r1 = vall[4 * hh + 1], r2 = vall[4 * hh + 2], r3 = vall[4 * hh + 3];
                    *reinterpret_cast<float4*>(dst)          = make_float4(r0.x, r1.x, r2.x, r3.x);
                    *reinterpret_cast<float4*>(dst + VS)     = make_float4(r0.y, r1.y, r2.y, r3.y);
                    *reinterpret_cast<float4*>(dst + 2 * VS) = make_float4(r0.z, r1.z, r2.z, r3.z);
                    *reinterpret_cast<float4*>(dst + 3 * VS) = make_float4(r0.w, r1.w, r2.w, r3.w);
                }
            }
        }
    };
    constexpr bool park_early = SPLIT && FLM_SPLIT_PARK_EARLY;
    const bool short_ctx = !SPLIT && G == 1 && T <= 64;
    if (short_ctx) {
        __syncthreads();                                            // the scores are in sc[]
        if (wave == 0) {
            const float sv = lane < T ? sc[lane] : -INFINITY;
            const float m = wave_max(sv);
            const float e = lane < T ? expf_ref(__fsub_rn(sv, m), etab) : 0.f;
            float pre = e;                                          // lane 0: 0 + e_0 = e_0
#pragma unroll 4
            for (int k = 1; k < T; ++k)
                pre = __fadd_rn(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pre), 0x138 /* wave_shr:1 */, 0xF, 0xF, true)), e);
            const float sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pre), T - 1));
            const float w = __fdiv_rn(e, sum);
            if (lane < T) sc[lane] = (lane > 0 && fabsf(w) <= 1e-15f) ? 0.f : w;   // rows t >= 1 with |att| <= 1e-15 are skipped (transformer.cpp:449)
        }
        stamp(2); stamp(3);
    } else {
    // block max over 16 waves (array_max is order-free)
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    for (int t = tid; t < T; t += kAttnBlock) sc[t] = expf_ref(__fsub_rn(sc[t], m), etab);
    __syncthreads();
    stamp(2);
    if (park_early && wave != 0) park_v();
    if (tid == 0) {                                                // sum += x[i], i ascending (tf_operators.cpp:180-183)
        // a lone lane: the T dependent adds, the operands through two groups of four 16-byte registers filled by INLINE-ASSEMBLY LDS reads with ONE s_waitcnt per 16 elements -- the
        // compiler's own reads get a wait in front of every register's first use (6 instructions per 4 elements; here 21 per 16), and a lone wave issues an instruction every ~8 clocks
        // whatever its kind (tools/ubench/sumchain.hip: T = 301 / 517 / 901 1.76 / 2.93 / 4.94 -> 1.44 / 2.34 / 3.87 us; 64 lanes with a DPP add per step: 1.93 / 3.39 / 5.57 -- the
        // s_nop the hazard between two dependent DPP operations needs is an instruction like any other)
        float sum = 0.f;
        int t = 0;
#define FLM_RD4(a0, a1, a2, a3, addr) asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48" : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(addr) : "memory");
#define FLM_WAIT4(a0, a1, a2, a3, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "memory");
#define FLM_ADD4(q) sum = __fadd_rn(sum, q.x); sum = __fadd_rn(sum, q.y); sum = __fadd_rn(sum, q.z); sum = __fadd_rn(sum, q.w);
        if (T >= 32) {
            unsigned ad = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)sc;
            v4f q0, q1, q2, q3, q4, q5, q6, q7;
            FLM_RD4(q0, q1, q2, q3, ad) { const unsigned a2 = ad + 64; FLM_RD4(q4, q5, q6, q7, a2) }
            for (; t + 32 <= T; t += 32) {                      // (reads run up to 63 elements past the last full block: sc's slack)
                ad += 128;
                FLM_WAIT4(q0, q1, q2, q3, 4) FLM_ADD4(q0) FLM_ADD4(q1) FLM_ADD4(q2) FLM_ADD4(q3) FLM_RD4(q0, q1, q2, q3, ad)
                { const unsigned a2 = ad + 64; FLM_WAIT4(q4, q5, q6, q7, 4) FLM_ADD4(q4) FLM_ADD4(q5) FLM_ADD4(q6) FLM_ADD4(q7) FLM_RD4(q4, q5, q6, q7, a2) }
            }
            FLM_WAIT4(q0, q1, q2, q3, 0) FLM_WAIT4(q4, q5, q6, q7, 0)
            if (t + 4 <= T) { FLM_ADD4(q0) t += 4; } if (t + 4 <= T) { FLM_ADD4(q1) t += 4; } if (t + 4 <= T) { FLM_ADD4(q2) t += 4; } if (t + 4 <= T) { FLM_ADD4(q3) t += 4; }
            if (t + 4 <= T) { FLM_ADD4(q4) t += 4; } if (t + 4 <= T) { FLM_ADD4(q5) t += 4; } if (t + 4 <= T) { FLM_ADD4(q6) t += 4; }
        }
#undef FLM_ADD4
        for (; t < T; ++t) sum = __fadd_rn(sum, sc[t]);
        red[16] = sum;
        red[20] = 1.f;                                              // (park_early: "no row is skipped" until the divide loop says otherwise)
    }
    if (park_early && wave == 0) park_v();
    __syncthreads();
    stamp(3);
    const float sum = red[16];
    // att[t] = exp / sum; rows t >= 1 with |att| <= 1e-15 are skipped by the weighted sum (transformer.cpp:449): they are
    // stored as exact zeros so that the PV chain can tell them apart with one wave-uniform test per four positions
    bool zero_w = false;
    for (int t = tid; t < T; t += kAttnBlock) { const float w = __fdiv_rn(sc[t], sum); const bool z = t > 0 && fabsf(w) <= 1e-15f; sc[t] = z ? 0.f : w; zero_w = zero_w || z; }
    if (park_early && zero_w) red[20] = 0.f;                        // some row is skipped: the chain takes the careful path
    }
    // (the first barrier of the loop below orders these writes before the PV reads)
    // ---- o[d] = sum_t att[t] V[t][d]: one thread per output dimension (of this part), t ascending (the reference's chain) over
    //      the LDS tiles.
    float o = 0.f;
    constexpr int vrs = rs;
    if constexpr (SPLIT) {
        // The part's V slice goes to LDS TRANSPOSED -- vT[d][t], rows kSplitVS floats apart (a multiple of 4, = 4 mod 64: the
        // parking stores and the chain's 16-byte reads are conflict-free) -- so that the chain of dimension d reads four consecutive
        // positions with one ds_read_b128.  The chain itself is the reference's (row 0 by multiplication, t ascending, FMA, rows whose
        // weight fell under the threshold skipped); with operands streaming through a 4-slot register ring it runs at the pace
        // of the dependent FMAs (measured before this layout: 35 cycles per position, the LDS latency of scalar reads).
        constexpr int VS = kSplitMaxSeq + 4;
        static_assert(kSplitVRegs == 8, "two halves of four consecutive rows per thread");
        if constexpr (!FLM_SPLIT_PARK_EARLY) {
            if (tid == 0) red[20] = 1.f;
            __syncthreads();                                        // every wave is done with the K tiles (and has written its weights)
            {
                bool zw = false;
                for (int t = tid; t < T; t += kAttnBlock) zw = zw || (t > 0 && sc[t] == 0.f);
                if (zw) red[20] = 0.f;                              // some row is skipped: the chain takes the careful path
            }
            park_v();
        }
        __syncthreads();                                            // (park_early: the slice was parked under the sum chain; the weights and the skip flag are written)
        if (tid < nd) {
            const float* vp = tile0 + tid * VS;
            const float* wp = sc;
            o = __fmul_rn(vp[0], wp[0]);                            // row 0 always (tf_operators.cpp:331-336)
            int p = 1;
            if (red[20] != 0.f) {
                for (; p < T && (p & 3); ++p) o = __fmaf_rn(vp[p], wp[p], o);
                // (operands as in the sum chain above: two groups of 16 positions -- four 16-byte registers of V, four of weights -- by inline-assembly LDS reads, one s_waitcnt per group)
#define FLM_PV4(vv, ww) o = __fmaf_rn(vv.x, ww.x, o); o = __fmaf_rn(vv.y, ww.y, o); o = __fmaf_rn(vv.z, ww.z, o); o = __fmaf_rn(vv.w, ww.w, o);
#define FLM_WAIT8(a0, a1, a2, a3, b0, b1, b2, b3, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) :: "memory");
                if (p + 32 <= T) {
                    unsigned av = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)vp + (unsigned)p * 4u, aw = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)wp + (unsigned)p * 4u;
                    v4f v0, v1, v2, v3, w0, w1, w2, w3, v4, v5, v6, v7, w4, w5, w6, w7;
                    FLM_RD4(v0, v1, v2, v3, av) FLM_RD4(w0, w1, w2, w3, aw)
                    { const unsigned av2 = av + 64, aw2 = aw + 64; FLM_RD4(v4, v5, v6, v7, av2) FLM_RD4(w4, w5, w6, w7, aw2) }
                    for (; p + 32 <= T; p += 32) {                  // (reads run up to 63 positions past the last full block: the row's / sc's slack, inside the LDS)
                        av += 128; aw += 128;
                        FLM_WAIT8(v0, v1, v2, v3, w0, w1, w2, w3, 8) FLM_PV4(v0, w0) FLM_PV4(v1, w1) FLM_PV4(v2, w2) FLM_PV4(v3, w3) FLM_RD4(v0, v1, v2, v3, av) FLM_RD4(w0, w1, w2, w3, aw)
                        { const unsigned av2 = av + 64, aw2 = aw + 64; FLM_WAIT8(v4, v5, v6, v7, w4, w5, w6, w7, 8) FLM_PV4(v4, w4) FLM_PV4(v5, w5) FLM_PV4(v6, w6) FLM_PV4(v7, w7) FLM_RD4(v4, v5, v6, v7, av2) FLM_RD4(w4, w5, w6, w7, aw2) }
                    }
                    FLM_WAIT8(v0, v1, v2, v3, w0, w1, w2, w3, 0) FLM_WAIT8(v4, v5, v6, v7, w4, w5, w6, w7, 0)
                    if (p + 4 <= T) { FLM_PV4(v0, w0) p += 4; } if (p + 4 <= T) { FLM_PV4(v1, w1) p += 4; } if (p + 4 <= T) { FLM_PV4(v2, w2) p += 4; } if (p + 4 <= T) { FLM_PV4(v3, w3) p += 4; }
                    if (p + 4 <= T) { FLM_PV4(v4, w4) p += 4; } if (p + 4 <= T) { FLM_PV4(v5, w5) p += 4; } if (p + 4 <= T) { FLM_PV4(v6, w6) p += 4; }
                }
#undef FLM_WAIT8
#undef FLM_PV4
#undef FLM_RD4
#undef FLM_WAIT4
                for (; p < T; ++p) o = __fmaf_rn(vp[p], wp[p], o);
            } else {
                for (; p < T; ++p) { const float w = wp[p]; o = w == 0.f ? o : __fmaf_rn(vp[p], w, o); }
            }
        }
    }
    for (int base = 0; base < (SPLIT ? 0 : nt); base += DV) {
#pragma unroll
        for (int u = 0; u < DV; ++u) {
            const int i = base + u;
            float* cur = (u & 1) ? tile1 : tile0;
            park(cur, prowv, loffv, ringV[u]);
            if (i == 0) stamp(12);
            __syncthreads();
            if (i == 0) stamp(8);
            request(rV, i + DV, nt, prowv, goffv, ringV[u], T);
            const float* wp = sc + i * kAttnTile;
            const int np = (T - i * kAttnTile) < kAttnTile ? (T - i * kAttnTile) : kAttnTile;
            unsigned long long live = 0;
            if (i < nt) {                                           // (whole waves: the map below is a ballot over 64 lanes)
                const float wl = lane < np ? wp[lane] : 0.f;
                live = __ballot(wl != 0.f);
                if (i == 0) live |= 1ull;
            }
            if (i < nt && tid < nd) {
                const float* vp = cur + tid;
                // The chain is T dependent FMAs; everything else must stay out of its way.  Weights are wave-uniform (one per position): lane p looks at weight p once per
                // tile and the ballot is the tile's map of rows that count -- a row whose weight was stored as exact 0 (the threshold of transformer.cpp:449) leaves o
                // untouched; row 0 counts always, by multiplication (tf_operators.cpp:331-336).  Groups of four positions (four LDS reads at immediate offsets + one 16-byte read
                // of the weights) go through a ring of three register sets: a group is read two groups (~100 cycles, the LDS latency) before its FMAs run, and the wait in
                // front of a group asks only for that group.  A group is four FMAs (no row skipped), nothing (all skipped: a peaked softmax leaves long runs of them) or
                // FMAs under scalar bit tests.  Straight-line code, 16 groups with early exits: around a loop's back edge the compiler's own s_waitcnt would drain the
                // ring once per trip.  Reads run up to 11 rows / weights past the tile's last position (the slack rows of attn_lds_bytes, sc's slack); never used.
                const int ng = (np + 3) >> 2;
                float A0, A1, A2, A3, B0, B1, B2, B3, C0, C1, C2, C3;
#define FLM_PVRD(G, g) { G##0 = vp[(4 * (g)) * vrs]; G##1 = vp[(4 * (g) + 1) * vrs]; G##2 = vp[(4 * (g) + 2) * vrs]; G##3 = vp[(4 * (g) + 3) * vrs]; \
                         G##W = *reinterpret_cast<const float4*>(wp + 4 * (g)); __builtin_amdgcn_sched_barrier(0); }
#define FLM_PVUSE(G, g) { const unsigned m = (unsigned)(live >> (4 * (g))) & 15u; \
                          if (m == 15u) { \
                              o = (i == 0 && (g) == 0) ? __fmul_rn(G##0, G##W.x) : __fmaf_rn(G##0, G##W.x, o); \
                              o = __fmaf_rn(G##1, G##W.y, o); o = __fmaf_rn(G##2, G##W.z, o); o = __fmaf_rn(G##3, G##W.w, o); \
                          } else if (m != 0u) { \
                              if (m & 1u) o = (i == 0 && (g) == 0) ? __fmul_rn(G##0, G##W.x) : __fmaf_rn(G##0, G##W.x, o); \
                              if (m & 2u) o = __fmaf_rn(G##1, G##W.y, o); if (m & 4u) o = __fmaf_rn(G##2, G##W.z, o); if (m & 8u) o = __fmaf_rn(G##3, G##W.w, o); \
                          } \
                          __builtin_amdgcn_sched_barrier(0); }
                const unsigned long long full = np >= 64 ? ~0ull : ((1ull << np) - 1ull);
                if ((live & full) == full) {
                    // No row skipped -- the usual case.  A lone wave issues an instruction every ~8 cycles whatever its kind, so what counts is the NUMBER of instructions per
                    // position: a group is a compare, a branch not taken, ONE wait, four FMAs and its successor's five reads, in a loop (12 positions per trip).  The reads
                    // and the waits are inline assembly: the compiler's own s_waitcnt drains the ring at a loop head (and straight-line code over 16 groups spilled).
                    // (row 0 by multiplication = a chain that starts at -0: x y + (-0) is x y, sign of zero included)
                    if (i == 0) o = -0.f;
                    unsigned va = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)vp;
                    unsigned wa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)wp;
                    v4f AW, BW, CW;
#define FLM_PVRDA(G, g) asm volatile("ds_read_b32 %0, %5 offset:%6\n\tds_read_b32 %1, %5 offset:%7\n\tds_read_b32 %2, %5 offset:%8\n\tds_read_b32 %3, %5 offset:%9\n\tds_read_b128 %4, %10 offset:%11" \
                                     : "=&v"(G##0), "=&v"(G##1), "=&v"(G##2), "=&v"(G##3), "=&v"(G##W) \
                                     : "v"(va), "n"((4 * (g)) * vrs * 4), "n"((4 * (g) + 1) * vrs * 4), "n"((4 * (g) + 2) * vrs * 4), "n"((4 * (g) + 3) * vrs * 4), "v"(wa), "n"(16 * (g)));
#define FLM_PVFMA(G) { asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(G##0), "+v"(G##1), "+v"(G##2), "+v"(G##3), "+v"(G##W)); \
                       o = __fmaf_rn(G##0, G##W.x, o); o = __fmaf_rn(G##1, G##W.y, o); o = __fmaf_rn(G##2, G##W.z, o); o = __fmaf_rn(G##3, G##W.w, o); }
                    const int nfull = np >> 2;
                    FLM_PVRDA(A, 0) FLM_PVRDA(B, 1) FLM_PVRDA(C, 2)
                    int g = 0;
                    // whole trips of three groups without the exit tests (4 of a group's 14 instructions), then the rest under them
#pragma unroll 1
                    for (int trips = nfull / 3; trips > 0; --trips) {
                        FLM_PVFMA(A) FLM_PVRDA(A, 3) FLM_PVFMA(B) FLM_PVRDA(B, 4) FLM_PVFMA(C) FLM_PVRDA(C, 5)
                        va += 12 * vrs * 4; wa += 48; g += 3;
                    }
#pragma unroll 1
                    while (true) {
                        if (g >= nfull) break; FLM_PVFMA(A) FLM_PVRDA(A, 3) ++g;
                        if (g >= nfull) break; FLM_PVFMA(B) FLM_PVRDA(B, 4) ++g;
                        if (g >= nfull) break; FLM_PVFMA(C) FLM_PVRDA(C, 5) ++g;
                        va += 12 * vrs * 4; wa += 48;
                    }
                    // the tile's last, partial group sits in the ring already
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(AW), "+v"(B0), "+v"(B1), "+v"(B2), "+v"(B3), "+v"(BW), "+v"(C0), "+v"(C1), "+v"(C2), "+v"(C3), "+v"(CW));
                    const int rem = np & 3, slot = g % 3;
                    if (rem) {
                        const float t0 = slot == 0 ? A0 : slot == 1 ? B0 : C0, t1 = slot == 0 ? A1 : slot == 1 ? B1 : C1, t2 = slot == 0 ? A2 : slot == 1 ? B2 : C2;
                        const v4f tw = slot == 0 ? AW : slot == 1 ? BW : CW;
                        o = __fmaf_rn(t0, tw.x, o);
                        if (rem > 1) o = __fmaf_rn(t1, tw.y, o);
                        if (rem > 2) o = __fmaf_rn(t2, tw.z, o);
                    }
#undef FLM_PVRDA
#undef FLM_PVFMA
                } else {
                    float4 AW, BW, CW;
                    __builtin_amdgcn_sched_barrier(0);
                    FLM_PVRD(A, 0) FLM_PVRD(B, 1) FLM_PVRD(C, 2)
                    do {
                        FLM_PVUSE(A, 0) FLM_PVRD(A, 3) if (ng <= 1) break;
                        FLM_PVUSE(B, 1) FLM_PVRD(B, 4) if (ng <= 2) break;
                        FLM_PVUSE(C, 2) FLM_PVRD(C, 5) if (ng <= 3) break;
                        FLM_PVUSE(A, 3) FLM_PVRD(A, 6) if (ng <= 4) break;
                        FLM_PVUSE(B, 4) FLM_PVRD(B, 7) if (ng <= 5) break;
                        FLM_PVUSE(C, 5) FLM_PVRD(C, 8) if (ng <= 6) break;
                        FLM_PVUSE(A, 6) FLM_PVRD(A, 9) if (ng <= 7) break;
                        FLM_PVUSE(B, 7) FLM_PVRD(B, 10) if (ng <= 8) break;
                        FLM_PVUSE(C, 8) FLM_PVRD(C, 11) if (ng <= 9) break;
                        FLM_PVUSE(A, 9) FLM_PVRD(A, 12) if (ng <= 10) break;
                        FLM_PVUSE(B, 10) FLM_PVRD(B, 13) if (ng <= 11) break;
                        FLM_PVUSE(C, 11) FLM_PVRD(C, 14) if (ng <= 12) break;
                        FLM_PVUSE(A, 12) FLM_PVRD(A, 15) if (ng <= 13) break;
                        FLM_PVUSE(B, 13) FLM_PVRD(B, 16) if (ng <= 14) break;
                        FLM_PVUSE(C, 14) FLM_PVRD(C, 17) if (ng <= 15) break;
                        FLM_PVUSE(A, 15)
                    } while (false);
                }
#undef FLM_PVRD
#undef FLM_PVUSE
                if (i == 0) stamp(9);
            }
        }
    }
    stamp(4);
    if (tid < nd) {
        if (gr_out) {                                                         // (k_layers' granule hand-offs: AttnArgs::gout)
            const unsigned long long gv = ((unsigned long long)epoch_arg << 32) | (unsigned long long)__float_as_uint(o);
            __hip_atomic_store(a.gout + (size_t)h * hs + d0 + tid, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = 0; i < a.n_peer; ++i) __hip_atomic_store(a.gout_peer[i] + (size_t)h * hs + d0 + tid, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
        st_agent(orow + (size_t)h * hs + d0 + tid, o);
        for (int i = 0; i < a.n_peer; ++i) __hip_atomic_store(a.out_peer[i] + (size_t)h * hs + d0 + tid, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (a.oq && G == 1) {
        // qx.quantize(x2) (transformer.cpp:138) for this head's groups: wave w holds the 64 outputs of group h * hs/64 + w;
        // the max is order-free, the element step is quant_elem.  Packed through LDS (q's place: long since consumed) so that
        // the values leave as dwords.
        const int esz = a.oqt == QT_INT8 ? 1 : 2;
        if (tid < hs) {
            const float mx = wave_max(fabsf(o));
            const float sc = __fdiv_rn(mx, a.oqt == QT_INT8 ? QTraits<QT_INT8>::kF : QTraits<QT_INT16>::kF);
            const int q = quant_elem(o, sc);
            if (esz == 1) reinterpret_cast<signed char*>(qs)[tid] = (signed char)q; else reinterpret_cast<short*>(qs)[tid] = (short)q;
            if (lane == 0) st_agent(a.os + (size_t)h * (hs >> 6) + wave, sc);
        }
        __syncthreads();
        const int nq = hs * esz / 4;
        if (tid < nq) __hip_atomic_store(reinterpret_cast<unsigned*>(a.oq) + (size_t)h * nq + tid, reinterpret_cast<const unsigned*>(qs)[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stamp(6);
    __syncthreads();                                                // the LDS is free for whoever runs next on it
}
// The first two K tiles of a split head's part by LDS-DMA into the tile layout at lds + off (row stride attn_row_stride(hs) floats; one row per instruction: hs / 4 lanes x 16 bytes -- the
// rows are padded, so a wave-wide DMA cannot span two), rows of EARLIER tokens only (t < T - 1: this token's row is patched in behind the q flag round).  Issued by every wave of the
// part's workgroup under the layer's QKV phase (layer_body); every wave waits for its own DMA (vmcnt) at the phase's end.
__device__ __forceinline__ void attn_kpre_issue(const AttnArgs& a, const int h, const int g, const int G, const int T, const char* lds, const unsigned off) {
    const int hs = a.hs, rs = attn_row_stride(hs), f4r = hs >> 2, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nt = (T + kAttnTile - 1) / kAttnTile, tpp = (nt + G - 1) / G, sb = g * tpp, se = (sb + tpp < nt) ? sb + tpp : nt;
    const float* K = a.kcache + (size_t)h * (a.kv_rows ? a.kv_rows : a.max_seq) * hs;
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K), 0, a.max_seq * hs * 4, 0x00020000);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)lds + off;
    for (int r = wave; r < 2 * kAttnTile; r += kAttnBlock / 64) {
        const int u = r / kAttnTile, row = r - u * kAttnTile, t = (sb + u) * kAttnTile + row;
        if (sb + u >= se || t >= T - 1) continue;                                  // (wave-uniform)
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)((u * kAttnTile + row) * rs * 4)));
        const unsigned src = (unsigned)(t * hs * 4 + lane * 16);
        if (lane < f4r) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(src), "s"(rK), "s"(dst) : "memory", "m0");
    }
}
// SPLIT is a template argument of the kernels (not a run-time branch inside one kernel): the two forms keep different things in
// registers, and compiled into one function they spilled a 16-byte register -- behind an s_waitcnt vmcnt(0) on the whole prefetch
template <bool COH, bool SPLIT, bool PRE = false, bool GRIN = false, class Mid = AttnNoMid>
// epoch: the value the parts of a split head raise / wait for in their score exchange (0: a.epoch; k_layers passes the layer's flag target, which counts from the token's epoch base)
__device__ __forceinline__ void attn_head_any(const AttnArgs& a, const int h, char* lds, const int T, const float* qrow, float* orow, const int g = 0, const int G = 1, Mid&& mid = Mid(), const unsigned epoch = 0u, const bool gr_out = false, const bool gr_sc = false, const float* kpre = nullptr) {
    if constexpr (SPLIT) {      // the host picks G = hs / kSplitDims (attn_parts): every part owns 32 output dimensions; hs <= 128
        if (a.hs <= 64) attn_head<1, COH, true, PRE, Mid, GRIN>(a, h, lds, T, qrow, orow, g, G, static_cast<Mid&&>(mid), epoch, gr_out, gr_sc, kpre); else attn_head<2, COH, true, PRE, Mid, GRIN>(a, h, lds, T, qrow, orow, g, G, static_cast<Mid&&>(mid), epoch, gr_out, gr_sc, kpre);
    } else {
        if (a.hs <= 64) attn_head<1, COH, false, PRE, Mid, GRIN>(a, h, lds, T, qrow, orow, 0, 1, static_cast<Mid&&>(mid), epoch, gr_out); else if (a.hs <= 128) attn_head<2, COH, false, PRE, Mid, GRIN>(a, h, lds, T, qrow, orow, 0, 1, static_cast<Mid&&>(mid), epoch, gr_out); else attn_head<4, COH, false, PRE, Mid, GRIN>(a, h, lds, T, qrow, orow, 0, 1, static_cast<Mid&&>(mid), epoch, gr_out);
    }
}
// batched prefill: workgroup (h, i) is query i of the batch, at position pos0 + i, over the cache rows 0 .. pos0 + i
inline __global__ void __launch_bounds__(kAttnBlock) k_attn_prefill(const AttnArgs a, int pos0, int row_stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int i = blockIdx.y;
    attn_head_any<false, false>(a, blockIdx.x, lds, pos0 + i + 1, a.q + (size_t)i * row_stride, a.out + (size_t)i * row_stride);
}
// ------------------------------------------------------------------------------------------
// Batched prefill, several queries per workgroup (hs <= 128).  Workgroup (h, g) takes the kMqQueries consecutive queries
// i0 = g * kMqQueries ... of head h: every K and V tile is brought to LDS ONCE for all of them, and the 16 waves are all busy
// with chains (scores: 4 queries per thread on the same K operands; softmax: one wave per query; PV: one thread per (query,
// output dimension)) where the one-query kernel keeps 2 of them busy.  Every query's arithmetic is attn_head's, operation
// for operation: 8 strided accumulators added 0..7, * attn_scale, max, expf_ref, the sum t ascending, divide, row 0 by
// multiplication, rows t >= 1 by FMA with |w| <= 1e-15 skipped (causal: query j sees the rows t < T_j only).
// ------------------------------------------------------------------------------------------
constexpr int kMqQueries = 8;
__host__ inline size_t attn_mq_lds_bytes(int max_seq, int hs) {
    return (size_t)(kMqQueries * hs + kMqQueries * (((max_seq + 3) & ~3) + 8) + 2 * kAttnTile * attn_row_stride(hs)) * 4;
}
// PRE: the scores come from k_qk_mfma (a.sc_global: [heads][B][max_seq], already scaled): the scores pass is a copy
template <int NF, bool PRE = false>
__device__ __forceinline__ void attn_prefill_mq(const AttnArgs& a, const int h, char* lds, const int pos0, const int i0, const int nq, const int row_stride, const int B = 0) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    constexpr int NQ = kMqQueries, rs = NF * 64 + 8;
    const int hs = a.hs, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int scs = ((a.max_seq + 3) & ~3) + 8;                                 // stride of a query's score row
    float* qs = reinterpret_cast<float*>(lds);                                  // [NQ][hs]
    float* sc = qs + NQ * hs;                                                   // [NQ][scs]
    float* tile0 = sc + NQ * scs;
    float* tile1 = tile0 + kAttnTile * rs;
    const int Tmax = pos0 + i0 + nq;                                            // rows the last query of this workgroup sees
    const int nt = (Tmax + kAttnTile - 1) / kAttnTile;
    const float* K = a.kcache + (size_t)h * (a.kv_rows ? a.kv_rows : a.max_seq) * hs;
    const float* V = a.vcache + (size_t)h * (a.kv_rows ? a.kv_rows : a.max_seq) * hs;
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K), 0, a.max_seq * hs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V), 0, a.max_seq * hs * 4, 0x00020000);
    const float scale = (float)(1.0 / (double)__builtin_sqrtf((float)hs));      // attn_scale, transformer.cpp:418
    // this thread's pieces of a tile (as in attn_head)
    const int f4r = hs >> 2, tile_f4 = kAttnTile * f4r, tile_bytes = kAttnTile * hs * 4;
    int prow[NF], goff[NF], loff[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int f = tid + j * kAttnBlock, row = f / f4r, c4 = f - row * f4r;
        prow[j] = f < tile_f4 ? row : (1 << 28);
        goff[j] = (row * hs + c4 * 4) * 4;
        loff[j] = row * rs + c4 * 4;
    }
    auto request = [&](const __amdgpu_buffer_rsrc_t& r, int tile, v4f (&reg)[NF]) {
        const int t0 = tile * kAttnTile;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const unsigned off = (tile < nt && t0 + prow[j] < Tmax) ? (unsigned)(tile * tile_bytes + goff[j]) : 0x80000000u;
            reg[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
        }
    };
    auto park = [&](float* buf, const v4f (&reg)[NF]) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
            if (prow[j] < kAttnTile) *reinterpret_cast<float4*>(buf + loff[j]) = make_float4(reg[j].x, reg[j].y, reg[j].z, reg[j].w);
    };
    v4f ring[NF];
    if constexpr (PRE) {
        // wave pair (2j, 2j+1) copies query j's row of scores
        const int j = wave >> 1;
        if (j < nq) {
            const int T = pos0 + i0 + j + 1;
            const float* src = a.sc_global + ((size_t)h * B + (i0 + j)) * a.max_seq;
            for (int t = (tid & 127); t < T; t += 128) sc[j * scs + t] = src[t];
        }
    } else {
    request(rK, 0, ring);
    for (int e = tid; e < NQ * hs; e += kAttnBlock) {
        const int j = e / hs, d = e - j * hs;
        qs[e] = j < nq ? a.q[(size_t)(i0 + j) * row_stride + (size_t)h * hs + d] : 0.f;
    }
    // ---- scores.  thread = (query half, position p, accumulator k); its 4 queries share the K operand of every step
    {
        const int k = tid & 7, p = (tid >> 3) & 63, jh = (tid >> 9) * 4;
        for (int s = 0; s < nt; ++s) {
            float* cur = (s & 1) ? tile1 : tile0;
            park(cur, ring);
            __syncthreads();                                                    // (also orders qs before its first use)
            request(rK, s + 1, ring);
            const float* kp = cur + p * rs + k;
            const float* q0 = qs + (jh + 0) * hs + k; const float* q1 = qs + (jh + 1) * hs + k;
            const float* q2 = qs + (jh + 2) * hs + k; const float* q3 = qs + (jh + 3) * hs + k;
            float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
#pragma unroll 8
            for (int i = 0; i < hs; i += 8) {
                const float kv = kp[i];
                l0 = __fmaf_rn(kv, q0[i], l0); l1 = __fmaf_rn(kv, q1[i], l1); l2 = __fmaf_rn(kv, q2[i], l2); l3 = __fmaf_rn(kv, q3[i], l3);
            }
            const int t = s * kAttnTile + p;
            auto fold = [&](float l, int j) {                                   // the 8 partials added 0..7 (lane k = 0 collects them)
                const int li = __float_as_int(l);
                float tot = __fadd_rn(0.f, l);
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x101 /* row_shl:1 */, 0xF, 0xF, true)));
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x102, 0xF, 0xF, true)));
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x103, 0xF, 0xF, true)));
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x104, 0xF, 0xF, true)));
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x105, 0xF, 0xF, true)));
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x106, 0xF, 0xF, true)));
                tot = __fadd_rn(tot, __int_as_float(__builtin_amdgcn_update_dpp(0, li, 0x107, 0xF, 0xF, true)));
                if (k == 0 && j < nq && t < pos0 + i0 + j + 1) sc[j * scs + t] = __fmul_rn(tot, scale);   // att.multiply(attn_scale) :443
            };
            fold(l0, jh + 0); fold(l1, jh + 1); fold(l2, jh + 2); fold(l3, jh + 3);
        }
    }
    }
    request(rV, 0, ring);                                                       // the first V tile travels under the softmax
    __syncthreads();
    // ---- softmax: wave j = query j (max is order-free; the sum is the reference's sequential one, tf_operators.cpp:180-183)
    if (wave < nq) {
        const int T = pos0 + i0 + wave + 1;
        float* row = sc + wave * scs;
        float m = -INFINITY;
        for (int t = lane; t < T; t += 64) m = fmaxf(m, row[t]);
        m = wave_max(m);
        for (int t = lane; t < T; t += 64) row[t] = expf_ref(__fsub_rn(row[t], m));
        float sum = 0.f;
        if (lane == 0) {                                                        // T dependent adds; the LDS reads run 16 elements ahead
            int t = 0;
            if (T >= 16) {
                const float4* r4 = reinterpret_cast<const float4*>(row);
                float4 a0 = r4[0], a1 = r4[1], a2 = r4[2], a3 = r4[3];
#define FLM_ADD4(q) sum = __fadd_rn(sum, q.x); sum = __fadd_rn(sum, q.y); sum = __fadd_rn(sum, q.z); sum = __fadd_rn(sum, q.w);
                for (; t + 32 <= T; t += 16) {
                    const float4 b0 = r4[t / 4 + 4], b1 = r4[t / 4 + 5], b2 = r4[t / 4 + 6], b3 = r4[t / 4 + 7];
                    FLM_ADD4(a0) FLM_ADD4(a1) FLM_ADD4(a2) FLM_ADD4(a3)
                    a0 = b0; a1 = b1; a2 = b2; a3 = b3;
                }
                FLM_ADD4(a0) FLM_ADD4(a1) FLM_ADD4(a2) FLM_ADD4(a3)
#undef FLM_ADD4
                t += 16;
            }
            for (; t < T; ++t) sum = __fadd_rn(sum, row[t]);
        }
        sum = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sum)));
        // att[t] = exp / sum; rows t >= 1 with |att| <= 1e-15 are skipped by the weighted sum (transformer.cpp:449): stored as exact zeros
        for (int t = lane; t < T; t += 64) { const float w = __fdiv_rn(row[t], sum); row[t] = (t > 0 && fabsf(w) <= 1e-15f) ? 0.f : w; }
    }
    // ---- o[j][d] = sum_t att_j[t] V[t][d]: one thread per (query, output dimension), t ascending, over the shared V tiles
    const int tpq = hs <= 64 ? 64 : 128;                                        // threads per query (a multiple of the wave: the skip test stays wave-uniform)
    const int j = tid / tpq, d = tid - j * tpq;
    const bool mine = j < nq && d < hs;
    const int Tj = pos0 + i0 + j + 1;
    float o = 0.f;
    for (int i = 0; i < nt; ++i) {
        float* cur = (i & 1) ? tile1 : tile0;
        park(cur, ring);
        __syncthreads();                                                        // (the first one also orders the softmax's writes before the reads below)
        request(rV, i + 1, ring);
        if (mine) {
            const float* vp = cur + d;
            const float* wp = sc + j * scs + i * kAttnTile;
            int np = Tj - i * kAttnTile; np = np < 0 ? 0 : (np > kAttnTile ? kAttnTile : np);
            int p = 0;
            if (i == 0 && np > 0) { o = __fmul_rn(vp[0], wp[0]); p = 1; }       // row 0 always (tf_operators.cpp:331-336)
            // the weights are wave-uniform (a wave's lanes belong to ONE query): if no row of the tile is skipped the walk is reads at
            // immediate offsets, 8 positions ahead of the 8 dependent FMAs (as in attn_head)
            const float wl = lane < np ? wp[lane] : 1.f;
            if (__all(wl != 0.f)) {
                for (; p < np && (p & 7); ++p) o = __fmaf_rn(vp[p * rs], wp[p], o);
                for (; p + 8 <= np; p += 8) {
                    const float* vq = vp + p * rs;
                    const float4 wa = *reinterpret_cast<const float4*>(wp + p), wb = *reinterpret_cast<const float4*>(wp + p + 4);
                    const float a0 = vq[0], a1 = vq[rs], a2 = vq[2 * rs], a3 = vq[3 * rs], a4 = vq[4 * rs], a5 = vq[5 * rs], a6 = vq[6 * rs], a7 = vq[7 * rs];
                    o = __fmaf_rn(a0, wa.x, o); o = __fmaf_rn(a1, wa.y, o); o = __fmaf_rn(a2, wa.z, o); o = __fmaf_rn(a3, wa.w, o);
                    o = __fmaf_rn(a4, wb.x, o); o = __fmaf_rn(a5, wb.y, o); o = __fmaf_rn(a6, wb.z, o); o = __fmaf_rn(a7, wb.w, o);
                }
                for (; p < np; ++p) o = __fmaf_rn(vp[p * rs], wp[p], o);
            } else {
                for (; p < np; ++p) { const float w = wp[p]; o = w == 0.f ? o : __fmaf_rn(vp[p * rs], w, o); }
            }
        }
    }
    if (mine) a.out[(size_t)(i0 + j) * row_stride + (size_t)h * hs + d] = o;
}
template <bool PRE>
__global__ void __launch_bounds__(kAttnBlock) k_attn_prefill_mq(const AttnArgs a, int pos0, int row_stride, int B) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int i0 = blockIdx.y * kMqQueries, nq = B - i0 < kMqQueries ? B - i0 : kMqQueries;
    if (a.hs <= 64) attn_prefill_mq<1, PRE>(a, blockIdx.x, lds, pos0, i0, nq, row_stride, B); else attn_prefill_mq<2, PRE>(a, blockIdx.x, lds, pos0, i0, nq, row_stride, B);
}

// ------------------------------------------------------------------------------------------
// Prefill QK^T on the matrix cores, fp32 in / fp32 accumulate (v_mfma_f32_16x16x4_f32), BIT-IDENTICAL to the reference's score:
//     att[q][t] = ( sum over the 8 strided lanes k of dot_product_avx256, added 0..7 ) * 1/sqrt(hs)      (x86_simd.cpp:1447-1467,
//                                                                                  quant_operators.cpp:340-348, 425-428)
// where lane k's partial is the sequential chain  l_k = fma(K[t][k + 8 s], q[k + 8 s], l_k),  s = 0, 1, ...  On gfx950 an f32 MFMA
// is bit for bit a k-ordered fmaf chain into its accumulator (D = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C))))), so lane k's
// chain is ONE accumulator fed with the head dimension permuted: the MFMA of step s' takes elements k + 8 (4 s' + c), c = 0..3, in
// its four k-slots.  Eight accumulators (one per strided lane) x hs / 32 MFMAs each = the whole contraction with no wasted flop;
// then the eight are added in order 0..7 and scaled on the VALU, exactly as the scalar code does.
// Workgroup (256 threads) = one head x 16 queries; its 4 waves take the 16-position tiles w, w + 4, ... of the rows the tile's queries
// see.  Nothing goes through LDS: lane (li, c) of the A operand needs Q[query li][8 (4 s' + c) + r], r = 0..7 -- 32 contiguous
// bytes per step s', kept in registers for the whole launch -- and of the B operand K[position li][the same 8 elements]: two 16-byte
// loads per step straight from the cache rows (the four k-slot lanes of a row cover 128 contiguous bytes), prefetched one tile ahead.
// (64 queries per workgroup, a wave per 16 of them walking the same rows -- three of four requests for a row would stop in the CU's
// L1 -- was slower: 38.6 vs 28.2 us at 512 tokens, a quarter of the workgroups and four times the serial walk.)
// Output: sc_global[h][query][t] for t <= pos0 + query (causal), consumed by k_attn_pv_mfma / k_attn_prefill_mq<true>.
// ------------------------------------------------------------------------------------------
constexpr int kQkQ = 16;
template <int NS>                                                // NS = hs / 32: MFMAs per accumulator ((hs / 8) chain elements / 4)
__global__ void __launch_bounds__(256) k_qk_mfma(const AttnArgs a, int pos0, int row_stride, int B) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int hs = a.hs, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, q0 = blockIdx.y * kQkQ;
    const int Tmax = pos0 + (q0 + kQkQ < B ? q0 + kQkQ : B);     // positions the last query of the tile sees
    const int nt = (Tmax + 15) >> 4;                             // 16-position tiles
    const float* K = a.kcache + (size_t)h * (a.kv_rows ? a.kv_rows : a.max_seq) * hs;
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K), 0, Tmax * hs * 4, 0x00020000);   // rows past the last position read as zero
    const float scale = (float)(1.0 / (double)__builtin_sqrtf((float)hs));
    const int li = lane & 15, c = lane >> 4;                      // operand lane: row / column li, k-slot c
    // A: this lane's elements of query q0 + li
    v4f qa[NS][2];
    {
        const float* qp = a.q + (size_t)(q0 + li) * row_stride + (size_t)h * hs;
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
            const int e = 8 * (4 * sp + c);
            if (q0 + li < B) { qa[sp][0] = *reinterpret_cast<const v4f*>(qp + e); qa[sp][1] = *reinterpret_cast<const v4f*>(qp + e + 4); }
            else { qa[sp][0] = v4f{0.f, 0.f, 0.f, 0.f}; qa[sp][1] = v4f{0.f, 0.f, 0.f, 0.f}; }
        }
    }
    v4f kb[2][NS][2];
    auto request = [&](int tile, v4f (&r)[NS][2]) {
        const unsigned base = (unsigned)(((tile * 16 + li) * hs + 8 * c) * 4);
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
            r[sp][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rK, (int)(base + sp * 128), 0, 0));
            r[sp][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rK, (int)(base + sp * 128 + 16), 0, 0));
        }
    };
    auto tile_scores = [&](int tile, const v4f (&kr)[NS][2]) {
        v4f acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[sp][r >> 2][r & 3], kr[sp][r >> 2][r & 3], acc[r], 0, 0, 0);
        }
        // D layout: column (position) = lane & 15, row (query) = 4 (lane >> 4) + reg
        const int t = tile * 16 + li;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int q = q0 + 4 * c + reg;
            float tot = __fadd_rn(0.f, acc[0][reg]);
#pragma unroll
            for (int r = 1; r < 8; ++r) tot = __fadd_rn(tot, acc[r][reg]);
            if (q < B && t < pos0 + q + 1) a.sc_global[((size_t)h * B + q) * a.max_seq + t] = __fmul_rn(tot, scale);
        }
    };
    request(wave, kb[0]);
    for (int tile = wave; tile < nt; tile += 8) {                 // two tiles per round: static register slots
        request(tile + 4, kb[1]);
        tile_scores(tile, kb[0]);
        request(tile + 8, kb[0]);
        if (tile + 4 < nt) tile_scores(tile + 4, kb[1]);
    }
}
// ------------------------------------------------------------------------------------------
// Prefill softmax + weighted sum with the weighted sum on the matrix cores (v_mfma_f32_16x16x4_f32), BIT-IDENTICAL to attn_head:
//     o[q][d] = V[0][d] * w[q][0], then o = fma(V[t][d], w[q][t], o) for t = 1, 2, ... ascending, rows with |w| <= 1e-15 skipped
//                                                                                  (tf_operators.cpp:325-350, transformer.cpp:449)
// An f32 MFMA is a k-ordered fmaf chain into its accumulator, so with A = the weights (row = query, k-slot = position) and B = V
// (k-slot = position, column = output dimension) one accumulator element IS the reference's chain of one (query, dimension):
//   * the first row by multiplication:  the accumulator starts at -0.0f, and fma(w, v, -0) == w * v for every w, v (signed zeros
//     included);
//   * a skipped row, and a row the query does not see (causal; the 16 queries of a tile end at 16 different positions), has weight
//     +0: fma(0, v, o) == o for finite v, unless o == -0 (possible only while every earlier product was a zero: the sign of an exact
//     zero output can differ from the reference's; no value downstream can see it -- the quantizer maps both zeros to 0);
//   * V rows past the tile's last position are never loaded (out-of-range buffer offsets read as zero), so no stale cache content
//     meets a zero weight.
// Workgroup (256 threads) = one head x 16 queries.  Scores come from k_qk_mfma (sc_global).  Softmax on the VALU as in
// attn_prefill_mq (max order-free; expf_ref; the sum t ascending: lane q of wave 0 = query q, rows the query does not see add +0;
// divide; skip rule), the weights land in LDS permuted inside blocks of 16 positions so that the A operand of four consecutive
// MFMAs is one 16-byte read.  Weighted sum: wave w owns 32 output dimensions = 2 accumulators (column li <-> dimensions
// 32 w + 2 li and + 1: ONE 8-byte load per lane and 4 positions, 128 contiguous bytes per V row and wave); V never touches LDS,
// it is prefetched kPvRing blocks (16 positions each) ahead into registers.
// ------------------------------------------------------------------------------------------
constexpr int kPvQ = 16, kPvRing = 4;
__host__ __device__ inline int pv_row_stride(int tmax) { return ((((tmax + 15) & ~15) + 63) & ~63) + 4; }     // floats; = 4 mod 64: conflict-free 16-byte reads of 16 rows
__host__ inline size_t pv_mfma_lds_bytes(int tmax, int qw = kPvQ) { return ((size_t)qw * pv_row_stride(tmax) + 16) * 4; }
// queries per workgroup: 16 while their exps fit the LDS (64 bytes per position: up to ~2500 positions), then 8, 4, 2, 1 -- the rows
// of the A operand past qw are fed zeros; the bits of a (query, dimension) chain do not depend on its row
__host__ inline int pv_mfma_queries(int tmax, size_t lds_max) { int qw = kPvQ; while (qw > 1 && pv_mfma_lds_bytes(tmax, qw) > lds_max) qw >>= 1; return qw; }
inline __global__ void __launch_bounds__(256) k_attn_pv_mfma(const AttnArgs a, int pos0, int row_stride, int B, int qw) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef float v2f __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int hs = a.hs, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, q0 = blockIdx.y * qw, nq = B - q0 < qw ? B - q0 : qw;
    const int S = pv_row_stride(pos0 + B);
    float* E = reinterpret_cast<float*>(lds);                    // [qw][S]
    float* sums = E + qw * S;                                    // [qw]
    const int Tmax = pos0 + q0 + nq, Tp = (Tmax + 15) & ~15, nblk = Tp >> 4;
    // ---- scores -> LDS, max, exp: 16 lanes per query (wave w: queries 4w .. 4w+3), 4 consecutive positions per lane and round
    {
        const int q = 4 * wave + (lane >> 4), l16 = lane & 15;
        const int Tq = q < nq ? pos0 + q0 + q + 1 : 0;
        const float* src = a.sc_global + ((size_t)h * B + (q0 + (q < nq ? q : 0))) * a.max_seq;
        float* row = E + q * S;
        const int Tr = q < qw ? Tp : 0;                          // (rows past qw do not exist)
        float m = -INFINITY;
        for (int t = l16 * 4; t < Tr; t += 64) {
            float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            if (t + 3 < Tq) v = *reinterpret_cast<const float4*>(src + t);
            else { if (t < Tq) v.x = src[t]; if (t + 1 < Tq) v.y = src[t + 1]; if (t + 2 < Tq) v.z = src[t + 2]; }
            *reinterpret_cast<float4*>(row + t) = v;
            m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
        m = row16_max(m);
        for (int t = l16 * 4; t < Tr; t += 64) {                 // (each lane re-reads what it wrote)
            float4 v = *reinterpret_cast<const float4*>(row + t);
            v.x = t < Tq ? expf_ref(__fsub_rn(v.x, m)) : 0.f; v.y = t + 1 < Tq ? expf_ref(__fsub_rn(v.y, m)) : 0.f;
            v.z = t + 2 < Tq ? expf_ref(__fsub_rn(v.z, m)) : 0.f; v.w = t + 3 < Tq ? expf_ref(__fsub_rn(v.w, m)) : 0.f;
            *reinterpret_cast<float4*>(row + t) = v;
        }
    }
    __syncthreads();
    // ---- the sums: lane q of wave 0, t ascending (tf_operators.cpp:180-183); positions past a query's own add +0
    if (tid < qw) {
        const float4* r4 = reinterpret_cast<const float4*>(E + tid * S);
        float sum = 0.f;
#pragma unroll 4
        for (int i = 0; i < Tp / 4; ++i) { const float4 v = r4[i]; sum = __fadd_rn(sum, v.x); sum = __fadd_rn(sum, v.y); sum = __fadd_rn(sum, v.z); sum = __fadd_rn(sum, v.w); }
        sums[tid] = sum;
    }
    __syncthreads();
    // ---- weights: divide, skip rule, permute inside the block of 16 positions: float4 c of a block = positions c, 4 + c, 8 + c, 12 + c
    for (int bidx = tid; bidx < qw * nblk; bidx += 256) {
        const int q = bidx / nblk, bi = bidx - q * nblk;
        float* blk = E + q * S + bi * 16;
        const float sum = sums[q];
        float w[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float4 v = *reinterpret_cast<const float4*>(blk + 4 * i); w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float x = q < nq ? __fdiv_rn(w[i], sum) : 0.f;
            w[i] = (bi * 16 + i > 0 && fabsf(x) <= 1e-15f) ? 0.f : x;           // rows t >= 1 with |att| <= 1e-15 are skipped (transformer.cpp:449)
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<float4*>(blk + 4 * c) = make_float4(w[c], w[4 + c], w[8 + c], w[12 + c]);
    }
    __syncthreads();
    // ---- weighted sum on the matrix cores
    const int li = lane & 15, c = lane >> 4, d0 = 32 * wave + 2 * li;
    if (32 * wave >= hs) return;                                  // (hs <= 96: the last waves have no dimensions)
    const float* V = a.vcache + (size_t)h * (a.kv_rows ? a.kv_rows : a.max_seq) * hs;
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V), 0, Tmax * hs * 4, 0x00020000);
    unsigned voff = d0 < hs ? (unsigned)((c * hs + d0) * 4) : 0x80000000u;      // position c of block 0; + 16 hs bytes per step
    const unsigned vstep = (unsigned)(4 * hs * 4);
    v2f ring[kPvRing][4];
    auto request = [&](v2f (&r)[4]) {
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) { r[sp] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rV, (int)voff, 0, 0)); voff += vstep; }
    };
#pragma unroll
    for (int r = 0; r < kPvRing; ++r) request(ring[r]);
    v4f acc0 = {-0.f, -0.f, -0.f, -0.f}, acc1 = {-0.f, -0.f, -0.f, -0.f};
    const float* ap = E + (li < qw ? li : 0) * S + 4 * c;
    for (int b0 = 0; b0 < nblk; b0 += kPvRing) {
#pragma unroll
        for (int r = 0; r < kPvRing; ++r) {
            if (b0 + r < nblk) {                                  // (wave-uniform)
                float4 av = *reinterpret_cast<const float4*>(ap + (b0 + r) * 16);
                if (li >= qw) av = make_float4(0.f, 0.f, 0.f, 0.f);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, ring[r][0].x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, ring[r][0].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, ring[r][1].x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, ring[r][1].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, ring[r][2].x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, ring[r][2].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, ring[r][3].x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, ring[r][3].y, acc1, 0, 0, 0);
            }
            request(ring[r]);                                     // block b0 + r + kPvRing (past the tile's last position: zeros)
        }
    }
    // D layout: column li (dimensions d0, d0 + 1), row (query) = 4 c + reg
    if (d0 < hs) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int q = 4 * c + reg;
            if (q < nq) {
                const size_t idx = (size_t)(q0 + q) * row_stride + (size_t)h * hs + d0;
                if (a.n_peer == 0) *reinterpret_cast<float2*>(a.out + idx) = make_float2(acc0[reg], acc1[reg]);
                else {   // tensor parallel: the local heads' columns of every rank's exchange region
                    st_agent(a.out + idx, acc0[reg]); st_agent(a.out + idx + 1, acc1[reg]);
                    for (int i = 0; i < a.n_peer; ++i) {
                        __hip_atomic_store(a.out_peer[i] + idx, acc0[reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(a.out_peer[i] + idx + 1, acc1[reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
        }
    }
}

// grid = heads * G (G = a.G >= 1 parts per head; all of them resident: the parts wait for each other's scores)
template <bool SPLIT>
__global__ void __launch_bounds__(kAttnBlock) k_attn_decode(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int G = SPLIT ? a.G : 1;
    attn_head_any<false, SPLIT>(a, blockIdx.x / G, lds, *a.pos_ptr + 1, a.q, a.out, blockIdx.x % G, G);
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
// k_attn_o under tensor parallelism (peer to peer): this rank's head parts are lines [line0, line0 + its parts) of n_lines; a head part raises its line in EVERY
// rank's array (system-scope store behind a release fence: its slice of att has gone to every rank's buffer before), the Wo workgroups wait for all n_lines
// lines of the local array and acquire.  The value is the token's epoch base (device memory, advanced by k_embed) + add: the lines are never cleared.
struct AoTp { unsigned* peer_flags[8]; const unsigned* base; unsigned add; int world, line0, n_lines; };
// PREQ: the heads hand over their output already quantized (AttnArgs::oq/os == GemvArgs::xq/xs): the GEMV workgroups
// copy 1 (2) bytes per element into LDS with coherent loads and skip the quantize prologue (~1.8 us of a 12.9 us launch).
template <int QT, int XR, bool PREQ, bool SPLIT = false>
__global__ void __launch_bounds__(kGemvBlock, 4) k_attn_o(const AttnArgs aa, const GemvArgs a, const int n_heads, unsigned* flag, const unsigned target_, int* err, const AoTp tp) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    auto stamp = [&](int k) { if (kAblate && a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime(); };   // tools/trace_ao.py
    stamp(0);
    const unsigned target = tp.world ? *tp.base + tp.add : target_;
    if ((int)blockIdx.x < n_heads) {                                           // n_heads counts (this rank's) head PARTS: heads * G
        const int G = SPLIT ? aa.G : 1;
        attn_head_any<false, SPLIT>(aa, blockIdx.x / G, lds, *aa.pos_ptr + 1, aa.q, aa.out, blockIdx.x % G, G);
        stamp(1);
        wait_stores_done();                                                     // every wave: its part of the head's output is where the others will read it
        __syncthreads();
        if (threadIdx.x == 0) {
            if (tp.world) {
                __atomic_thread_fence(__ATOMIC_RELEASE);
                for (int r = 0; r < tp.world; ++r) __hip_atomic_store(tp.peer_flags[r] + (tp.line0 + blockIdx.x) * kFlagStride, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else __hip_atomic_store(flag + blockIdx.x * kFlagStride, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        stamp(4);
        return;
    }
    GemvCtx<QT, EPI_RESIDUAL> g;
    g.init(a, blockIdx.x - n_heads, gridDim.x - n_heads, lds);
    g.issue(kAblate ? a.ablate : 0);
    stamp(1);
    const int n_poll = tp.world ? tp.n_lines : n_heads;
    if ((int)(threadIdx.x & ~63u) < n_poll) {                               // the waves that own at least one head's flag: lane i polls head i's line
        const bool mine = (int)threadIdx.x < n_poll;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (true) {
            unsigned f = target;
            if (mine) f = tp.world ? __hip_atomic_load(flag + threadIdx.x * kFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(flag + threadIdx.x * kFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(f - target) >= 0)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > (tp.world ? 2000000000ull : 2000000ull)) { __hip_atomic_store(err, tp.world ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }   // (ranks start seconds apart)
            if (tp.world) __builtin_amdgcn_s_sleep(4);
        }
        if (tp.world) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    stamp(2);
    float4 xv[XR > 0 ? XR : 1], nv[XR > 0 ? XR : 1];
    if constexpr (PREQ) {
        gemv_prologue<QT, PRO_NONE, 0, true>(a, lds, xv, nv, [](int) {});
    } else {
        gemv_preload<QT, PRO_QUANT, XR, true>(a, xv, nv);
        gemv_prologue<QT, PRO_QUANT, XR>(a, lds, xv, nv, [](int) {});
    }
    stamp(3);
    g.run(a, lds, [](int) {});
    stamp(4);
}

// ------------------------------------------------------------------------------------------
// QKV (+ RoPE + cache append), attention and the output projection in ONE launch (single GPU): k_attn_o with the QKV GEMV in front.
// Every workgroup runs its rows of [Wq; Wk; Wv] (k_gemv<RMSNORM_QUANT, ROPE_KV> verbatim), publishes them (write-through stores,
// then `target` in its line of flagq).  This hand-off is NOT all-to-all: a head needs its own hs rows of q, k and v only, i.e. the
// lines of the <= 3 * ceil(hs / Rm + 1) workgroups that reduced them -- no wait for the slowest of 256, no 16 KB vector to gather.
// The other workgroups go on as in k_attn_o (Wo prefetch, wait for the heads' lines, GEMV).
// TP: the launch spans the tensor-parallel ranks -- the QKV phase consumes the x1 exchange behind the previous layer's FFN2 (folded flag round, coherent loads; its
// rows are this rank's heads, so the QKV -> heads hand-off stays inside the rank: flagq is local and cleared by k_embed); the heads -> Wo hand-off as in k_attn_o (AoTp).
template <int QT, int XR, bool PREQ, bool SPLIT = false, bool TP = false>
__global__ void __launch_bounds__(kGemvBlock, 4) k_qkv_attn_o(const GemvArgs aq, const AttnArgs aa, const GemvArgs a, const int gridq, const int n_heads, const int grido,
                                                              unsigned* flagq, unsigned* flag, const unsigned target, int* err, const AoTp tp) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    auto nostamp = [](int) {};
    const unsigned htarget = TP ? *tp.base + tp.add : target;                   // the heads' lines (epoch values across ranks); flagq keeps `target`
    if constexpr (TP) { if (aq.xf.world) xchg_fold(aq.xf); }
    if ((int)blockIdx.x < gridq) {
        float4 xq[1], nq[1];
        gemv_preload<QT, PRO_RMSNORM_QUANT, 1, TP>(aq, xq, nq);
        GemvCtx<QT, EPI_ROPE_KV> gq;
        gq.init(aq, blockIdx.x, gridq, lds);
        gemv_prologue<QT, PRO_RMSNORM_QUANT, 1, TP>(aq, lds, xq, nq, [&](int part) { gq.issue(kAblate ? aq.ablate : 0, part); });
        gq.run(aq, lds, nostamp);
        wait_stores_done();                                                     // every wave: its q / cache rows are where the heads will read them
        __syncthreads();                                                        // (and the LDS is free for the next phase)
        if (threadIdx.x == 0) __hip_atomic_store(flagq + blockIdx.x * kFlagStride, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((int)blockIdx.x < n_heads) {                                            // n_heads counts head PARTS: heads * G
        const int G = SPLIT ? aa.G : 1, h = blockIdx.x / G;
        if (threadIdx.x < 256) {
            // lane i: does workgroup i reduce a row of this head's q, k or v?  (pass p = rows [p Rm, (p + 1) Rm) of [Wq; Wk; Wv], workgroup p mod gridq)
            bool need = false;
            if ((int)threadIdx.x < gridq) {
                const unsigned Rm = aq.rows_per_pass, hs = aa.hs, nq = gridq;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const unsigned r0 = (m == 0 ? 0u : m == 1 ? (unsigned)aq.dim : (unsigned)(aq.dim + aq.kv_dim)) + h * hs;
                    const unsigned pa = r0 / Rm, pb = (r0 + hs - 1) / Rm;
                    need |= (threadIdx.x + nq - pa % nq) % nq <= pb - pa;
                }
            }
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (true) {
                const unsigned f = need ? __hip_atomic_load(flagq + threadIdx.x * kFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                if (__all(f >= target)) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __syncthreads();
        attn_head_any<true, SPLIT>(aa, h, lds, *aa.pos_ptr + 1, aa.q, aa.out, blockIdx.x % G, G);
        wait_stores_done();
        __syncthreads();
        if (threadIdx.x == 0) {
            if constexpr (TP) {
                __atomic_thread_fence(__ATOMIC_RELEASE);
                for (int r = 0; r < tp.world; ++r) __hip_atomic_store(tp.peer_flags[r] + (tp.line0 + blockIdx.x) * kFlagStride, htarget, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else __hip_atomic_store(flag + blockIdx.x * kFlagStride, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if ((int)blockIdx.x >= n_heads + grido) return;
    GemvCtx<QT, EPI_RESIDUAL> g;
    g.init(a, blockIdx.x - n_heads, grido, lds);
    g.issue(kAblate ? a.ablate : 0);
    const int n_poll = TP ? tp.n_lines : n_heads;
    if ((int)(threadIdx.x & ~63u) < n_poll) {                               // the waves that own at least one head's flag: lane i polls head i's line
        const bool mine = (int)threadIdx.x < n_poll;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (true) {
            unsigned f = htarget;
            if (mine) f = TP ? __hip_atomic_load(flag + threadIdx.x * kFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(flag + threadIdx.x * kFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(f - htarget) >= 0)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > (TP ? 2000000000ull : 2000000ull)) { __hip_atomic_store(err, TP ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            if constexpr (TP) __builtin_amdgcn_s_sleep(4);
        }
        if constexpr (TP) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    float4 xv[XR > 0 ? XR : 1], nv[XR > 0 ? XR : 1];
    if constexpr (PREQ) {
        gemv_prologue<QT, PRO_NONE, 0, true>(a, lds, xv, nv, [](int) {});
    } else {
        gemv_preload<QT, PRO_QUANT, XR, true>(a, xv, nv);
        gemv_prologue<QT, PRO_QUANT, XR>(a, lds, xv, nv, [](int) {});
    }
    g.run(a, lds, [](int) {});
}

// ------------------------------------------------------------------------------------------
// FFN13 (+ SwiGLU) and FFN2 (+ residual) in ONE launch (single GPU): every workgroup runs its rows of [W1; W3], publishes its slice
// of hd (write-through stores, then its 64-byte flag line = layer + 1), requests its first two steps of W2 -- 128 KiB per CU, two
// thirds of the matrix chip-wide, none of it depending on hd -- and only then waits for the other workgroups' lines (lane i of
// waves 0..3 polls workgroup 64 w + i's).  The tail of FFN13, the hand-off and FFN2's quantize prologue pass while those weights
// stream; as two launches the same interval is a kernel boundary, a ramp and a wait with an empty memory pipeline
// (DESIGN.md section 7).  The two phases are k_gemv<RMSNORM_QUANT, SWIGLU> and k_gemv<QUANT, RESIDUAL> verbatim (hd is read
// with coherent loads).  All workgroups are resident (grid <= CUs, one 1024-thread workgroup per CU); a poll that never succeeds
// gives up after ~20 ms and raises *err (the host then re-runs the call on one kernel per phase).
// Tensor parallel (TP): the launch spans the ranks.  FFN13 consumes the x1 exchange (its flag round folded into this launch: xchg_fold, coherent loads); a rank's
// workgroups count themselves out on a device counter when their rows of hd have gone to every rank's buffer, the last one raises the RANK's line in every rank's
// array (release fence before, epoch value = the token's base + add; 8 lines, never cleared) -- an all-to-all between N x 256 workgroups would be N x 256 lines per
// workgroup to poll --, and the FFN2 workgroups of every rank wait for the N rank lines, acquire, and go on as on a single GPU.
struct FfnTp { unsigned* peer_flags[8]; const unsigned* base; unsigned add; int world, rank; unsigned long long* counter; };
template <int QT, int XR2, bool TP = false>
__global__ void __launch_bounds__(kGemvBlock, 4) k_ffn(const GemvArgs a13, const GemvArgs a2, const int grid13, const int grid2, unsigned* flag, const unsigned target_, int* err, const FfnTp tp) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    auto nostamp = [](int) {};
    const unsigned target = TP ? *tp.base + tp.add : target_;
    if constexpr (TP) { if (a13.xf.world) xchg_fold(a13.xf); }                  // (every workgroup: the barrier inside is the workgroup's own)
    if ((int)blockIdx.x < grid13) {
        float4 xv[1], nv[1];
        gemv_preload<QT, PRO_RMSNORM_QUANT, 1, TP>(a13, xv, nv);
        GemvCtx<QT, EPI_SWIGLU> g;
        g.init(a13, blockIdx.x, grid13, lds);
        gemv_prologue<QT, PRO_RMSNORM_QUANT, 1, TP>(a13, lds, xv, nv, [&](int part) { g.issue(kAblate ? a13.ablate : 0, part); });
        g.run(a13, lds, nostamp);
    }
    wait_stores_done();                                                         // every wave: its rows of hd are where the others will read them
    __syncthreads();                                                            // (and the LDS is free for the second phase)
    if (threadIdx.x == 0) {
        if constexpr (TP) {
            const unsigned long long old = __hip_atomic_fetch_add(tp.counter, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if ((old + 1) % gridDim.x == 0) {                                   // the rank's last workgroup (the counter only ever grows: gridDim.x per launch)
                __atomic_thread_fence(__ATOMIC_RELEASE);
                for (int r = 0; r < tp.world; ++r) __hip_atomic_store(tp.peer_flags[r] + tp.rank * kFlagStride, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        } else __hip_atomic_store(flag + blockIdx.x * kFlagStride, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((int)blockIdx.x >= grid2) return;
    GemvCtx<QT, EPI_RESIDUAL> g2;
    g2.init(a2, blockIdx.x, grid2, lds);
    g2.issue(kAblate ? a2.ablate : 0, 1);                                       // ONE set now (64 KiB per CU: taken by the memory pipeline before the flags come in); hd
    if (threadIdx.x < 256) {                                                    // requested behind two sets would wait for 128 KiB per CU to drain (measured: no gain at all)
        const bool mine = (int)threadIdx.x < (TP ? tp.world : (int)gridDim.x);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (true) {
            unsigned f = target;
            if (mine) f = TP ? __hip_atomic_load(flag + threadIdx.x * kFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(flag + threadIdx.x * kFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(f - target) >= 0)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > (TP ? 2000000000ull : 2000000ull)) { __hip_atomic_store(err, TP ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            if constexpr (TP) __builtin_amdgcn_s_sleep(4);
        }
        if constexpr (TP) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    float4 xv2[XR2 > 0 ? XR2 : 1], nv2[XR2 > 0 ? XR2 : 1];
    gemv_preload<QT, PRO_QUANT, XR2, true>(a2, xv2, nv2);
    gemv_prologue<QT, PRO_QUANT, XR2, true>(a2, lds, xv2, nv2, [&](int) { g2.issue(kAblate ? a2.ablate : 0, 2); });   // the second set once hd has arrived
    g2.run(a2, lds, nostamp);
}

} // namespace flm
