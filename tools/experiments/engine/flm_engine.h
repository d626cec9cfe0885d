// flm_engine.h -- the weight-streaming ENGINE: several dependent GEMVs of a token in ONE launch whose weight stream never stops at a phase edge.
// Part of flm_kernels.h (hand-written gfx950 / CDNA4 kernels of the fast-llama per-token hot path); include that header.
//
// Why: a decode GEMV launch is memory-pipeline-bound from the arrival of its activation to its last load, and everything else -- the
// kernel boundary, the ramp, the wait for the activation, the rmsnorm chain, the tail -- is a window in which the CU has nothing in flight
// (24 of a layer's 58 us).  Symmetric waves cannot close those windows: a wave that waits for an activation cannot
// request weights it has no registers for.  Here the roles are split (MI355X_MICROARCH.md, rows prefetch-credit / ldsdma-fill /
// engine-vs-launches):
//   * 4 LOADER waves per CU stream the CU's share of every phase's weights and scales into a ring of 8.5 KiB slots in LDS with LDS-DMA
//     (buffer_load_dwordx4 ... lds: no registers, no VALU), two fills in flight each -- a DMA stream reaches 6.9 TB/s chip-wide inside a
//     kernel (tools/ubench/ldsdma.hip; the register-load GEMV ~6.0).  Weights do not depend on activations, so the loaders run ahead
//     across every phase edge until the ring is full (14 slots = 112 KiB per CU = ~4 us of stream).
//   * 8 CONSUMER waves per CU do everything else: activation prologues (gather, the reference's rmsnorm chain, quantize), the integer dots
//     out of LDS, the reference's fp32 group chain, epilogues, and the hand-off of results to the other CUs.
// No workgroup barrier after the kernel's first instruction: the roles meet through sequence words in LDS only.
//
// Arithmetic (bit-identical to k_gemv, i.e. to the reference): quant::matmul<T> at w == 1 (src/blas/quant_operators.cpp:252-284)
//     out[r] = sum_g fma(sW[r,g] * sX[g], float(sum_{k<64} W[r,64g+k] * X[64g+k]), acc),   g ascending
// A PIECE is 4 rows x 256 bytes (one wave-wide 1 KiB DMA; 256 contiguous bytes per row keep HBM bursts whole), a UNIT is 4 rows x K.
// Consumer lane (r = lane >> 4, gi = (lane >> 2) & 3, q = lane & 3) holds bytes [64 gi + 16 q, +16) of row r of the piece: v_dot4 x 4,
// DPP quad sum -> the exact int32 dot of group gi of row r.  The row's accumulator is replicated in the 16 lanes of its DPP row, and every
// lane runs the chain acc = fma(sW[g] * sX[g], float(dot_g), acc), g = 0..3 of the piece in order, fetching dot_g from lane 4 g of the
// row with DPP row_newbcast: four dependent FMAs per piece, no LDS parking, no barrier.  SwiGLU: pieces of W1 and W3 alternate per
// column block, two accumulators.
//
// Rows are dealt to (CU, consumer) round robin in units: unit u -> CU u % nCU, the CU's i-th unit -> consumer i % 8.  Fill F of the ring
// belongs to consumer F % 8 and is issued by loader F % 4; a phase contributes 8 * NS fills (NS = the slots of the consumer with most
// work), fills of a consumer that has run out are empty and cost nothing.
//
// Hand-offs between CUs inside the launch: 8-byte granules {value, tag} written by ONE agent-scope store, polled by the data's readers
// (cdna_hip_programming.md Guideline 16, form R2): no flags, no fences.  tag = epoch of the producing phase = token base + phase + 1.
//   residual stream x1 : every owner lane keeps its row's value in a register for the whole launch and publishes it after Wo / FFN2;
//                        every CU gathers all of it for the next rmsnorm.
//   FFN hidden vector  : two hops, so that 11008 values cross as 11 KB instead of 88 KB: the SwiGLU epilogue publishes fp32 granules, the
//                        owner of a 64-row quant group (CU g % nCU) gathers its 64 values, quantizes them (quant::quantize) and publishes
//                        16 dwords + 1 scale; every CU gathers those.
#pragma once
#include "flm_math.h"
#include "flm_gemv.h"
// (bit-exactness hygiene: see flm_math.h -- no implicit FMA contraction in any of these headers)
#pragma clang fp contract(off)

namespace flm {

constexpr int kEngLoaders = 4, kEngConsumers = 12, kEngBlock = 64 * (kEngLoaders + kEngConsumers);
constexpr int kEngPieces = 8, kEngSlotW = kEngPieces * 1024, kEngSlotBytes = kEngSlotW + kEngPieces * 64;   // 8 KiB of weights + their 128 scales
constexpr int kEngSlots = 14;                 // ring slots: a compile-time constant (slot = F % 14 and the sequence number F / 14 are computed per fill by lone
                                              // waves that issue an instruction every ~6 cycles: a division by a run-time value per fill made the loaders the bound)
constexpr int kEngCL = 64 * kEngConsumers;    // consumer lanes of a workgroup
constexpr int kEngMaxOwn = 2;                 // residual rows a consumer lane can own: dim <= 4 * kEngConsumers * kEngMaxOwn * nCU
constexpr int kEngTrace = 512;                // trace words per workgroup (tools/trace_eng.py): consumer w at 16 w, loader L at 256 + 8 L, consumer accumulators at 320 + 4 w
constexpr int kEngEpochStride = 1024;         // the token's epoch base advances by this (k_embed): phases per token < 1024

enum EngPro { EPRO_X_RMS = 0,     // x (plain fp32 array, complete when the launch starts) -> rmsnorm -> quantize
              EPRO_X_Q = 1,       // x (plain) -> quantize
              EPRO_XQ = 2,        // pre-quantized activation (plain arrays xq / xs)
              EPRO_GRAN_RMS = 3,  // x1 from the granules of the previous phase of this launch -> rmsnorm -> quantize
              EPRO_GRAN_HD = 4 }; // hd, pre-quantized by its group owners, from granules

struct EngPhase {
    const void* W; const float* sW;           // [rows (x 2: SWIGLU)][K] values, [..][K / 64] scales
    int K, rows, epi, pro;                    // rows per matrix (SWIGLU: hidden; ROPE_KV: 3 x dim_local)
    const float* x; const float* norm_w; const void* xq; const float* xs;
    float* out;                               // plain results (x1 / hd / q / logits)
    float* kcache; float* vcache;             // ROPE_KV: this layer's caches [heads][max_seq][hs]
    int dim, kv_dim, hs, max_seq;             // ROPE_KV geometry
    int gran_out, index;                      // results also leave as granules (RESIDUAL -> gx1, SWIGLU -> ghd); index of the phase in the token's program (epochs)
    // geometry the host works out for the launch's grid (eng_fill_geom): the first `umod` workgroups have ucu_hi units, the others one fewer;
    // reciprocals for the quotients the loaders and consumers need per fill ([0]: workgroups with ucu_hi units, [1]: the others)
    int ucu_hi, umod;
    unsigned inv_ppuv, inv2SA[2], inv2SB[2];
};
constexpr int kEngMaxPhases = 5;
struct EngArgs {
    EngPhase ph[kEngMaxPhases]; int nph;      // the launch's phases, by value: scalar loads from the kernel-argument segment
    unsigned long long* gx1; unsigned long long* ghd; unsigned long long* ghq;
    const unsigned* base_ptr;                 // the token's epoch base (device memory; advanced once per token)
    const float* x1;                          // the residual stream when the launch starts (owners read their rows)
    const float* rope_cos; const float* rope_sin; const int* pos_ptr;
    int* err;
    unsigned long long* trace;                // tools/trace_eng.py: per-workgroup stamps
    int ablate;                               // FLM_ABLATE builds only: 1 consumers skip the dots, 2 no scale loads, 4 one fill in flight per loader, 16 no prologues / epilogues
};

// LDS: [ring: nslot x 17 KiB] [ctl: 256 B] [xq: kmax * esz] [xs: kmax / 64 floats] [chain staging: 4 strips (rmsnorm phases)]
struct EngLds { int off_ctl, off_xq, off_xs, off_stage, total; };
__host__ __device__ inline EngLds eng_lds_layout(int nslot, int kmax, int esz, int nmax_norm) {
    EngLds L;
    L.off_ctl = nslot * kEngSlotBytes;
    L.off_xq = L.off_ctl + 256;
    L.off_xs = L.off_xq + ((kmax * esz + 255) & ~255);
    L.off_stage = L.off_xs + (((kmax / kGroup) * 4 + 255) & ~255);
    L.total = L.off_stage + (nmax_norm ? 4 * chain_strip_floats(nmax_norm) * 4 + 64 : 0);
    return L;
}
// ctl words
enum { ECTL_FILL = 0, ECTL_FREE = 16, ECTL_SYNC = 32, ECTL_ABORT = 33, ECTL_XREQ = 34 /* consumer waves that have fetched the launch's first activation */,
       ECTL_GATHER = 35 /* consumer waves sweeping granules right now */, ECTL_RED = 36 };

// how the rows of a phase fall on (CU, consumer), and the numbering of the phase's ring fills.  Consumers w < rr ("long") have ua units =
// SA slots, the others ("short") ua - 1 units = SB slots.  Only fills that exist are numbered -- a consumer that has run out takes no ring
// slot -- and the two classes are MERGED BY PROGRESS: long round s (one fill of every long consumer) has key (2 s + 1) / SA, short round
// s' key (2 s' + 1) / SB, smaller keys first, long first on ties.  So a consumer with twice the rows is fed at twice the rate from the
// start, and everybody ends together (in plain round-robin order the long streams were starved during the first half of a phase and
// alone, at their own dot rate, during the second: FFN13 took 19 us instead of 13).
struct EngGeom {
    int PPUV;                    // pieces per unit (x 2: SWIGLU)
    int ucu;                     // units of this CU
    int ua, rr, SA, SB;          // see above
    int nfill;                   // fills of the phase on this CU
    unsigned inv2SA, inv2SB;     // ceil(2^32 / (2 SA)), ceil(2^32 / (2 SB)): exact quotients for the numerators that occur (< 2^16)
    static __host__ __device__ unsigned quo(unsigned x, unsigned d, unsigned inv) {
#if defined(__HIP_DEVICE_COMPILE__)
        return d > 1 ? __umulhi(x, inv) : x;
#else
        (void)inv; return x / d;
#endif
    }
    __host__ __device__ int units_of(int w) const { return w < rr ? ua : ua - 1; }
    __host__ __device__ int slots_of(int w) const { return w < rr ? SA : SB; }
    // short rounds in front of long round s: key' < key  <=>  (2 s' + 1) SA < (2 s + 1) SB
    __host__ __device__ int short_before(int s) const { const int num = (2 * s + 1) * SB - SA; if (num <= 0) return 0; const int q = (int)quo((unsigned)(num + 2 * SA - 1), (unsigned)(2 * SA), inv2SA); return q < SB ? q : SB; }
    // long rounds in front of short round s': key <= key'  <=>  (2 s + 1) SB <= (2 s' + 1) SA
    __host__ __device__ int long_before(int s) const { const int num = (2 * s + 1) * SA - SB; if (num < 0) return 0; const int q = (int)quo((unsigned)num, (unsigned)(2 * SB), inv2SB) + 1; return q < SA ? q : SA; }
    __host__ __device__ int fill_of(int s, int w) const {              // (s < slots_of(w))
        return w < rr ? rr * s + (kEngConsumers - rr) * short_before(s) + w : (kEngConsumers - rr) * s + rr * long_before(s) + (w - rr);
    }
};
// the part of the geometry that needs no division: from the units of the workgroup
__host__ __device__ inline EngGeom eng_geom_units(int ucu, int ppuv) {
    EngGeom g;
    g.PPUV = ppuv; g.ucu = ucu;
    g.ua = (ucu + kEngConsumers - 1) / kEngConsumers; g.rr = ucu - (g.ua - 1) * kEngConsumers;      // rr in 1..8 when ucu > 0   (kEngConsumers: a power of two)
    if (ucu == 0) { g.ua = 0; g.rr = kEngConsumers; }
    g.SA = (g.ua * ppuv + kEngPieces - 1) / kEngPieces;
    g.SB = g.ua > 0 ? ((g.ua - 1) * ppuv + kEngPieces - 1) / kEngPieces : 0;
    if (g.rr == kEngConsumers) g.SB = g.SA;                            // every consumer has ua units
    g.nfill = kEngConsumers * g.SB + g.rr * (g.SA - g.SB);
    g.inv2SA = 0; g.inv2SB = 0;
    return g;
}
__host__ __device__ inline int eng_ppuv(const EngPhase& P, int esz) { return (P.K * esz / 256) * (P.epi == EPI_SWIGLU ? 2 : 1); }
// host: the phase's per-grid numbers
inline void eng_fill_geom(EngPhase& P, int esz, int ncu) {
    const int U = (P.rows + 3) / 4, ppuv = eng_ppuv(P, esz);
    P.ucu_hi = (U + ncu - 1) / ncu; P.umod = U - (P.ucu_hi - 1) * ncu;                       // workgroups c < umod have ucu_hi units (umod in 1..ncu when U > 0)
    if (U == 0) { P.ucu_hi = 0; P.umod = ncu; }
    P.inv_ppuv = ppuv > 1 ? 0xFFFFFFFFu / (unsigned)ppuv + 1u : 0u;
    for (int v = 0; v < 2; ++v) {
        const int ucu = P.ucu_hi - v; const EngGeom g = eng_geom_units(ucu > 0 ? ucu : 0, ppuv);
        P.inv2SA[v] = g.SA > 0 ? 0xFFFFFFFFu / (unsigned)(2 * g.SA) + 1u : 0u; P.inv2SB[v] = g.SB > 0 ? 0xFFFFFFFFu / (unsigned)(2 * g.SB) + 1u : 0u;
    }
}
// device: workgroup c's geometry of phase P
__device__ __forceinline__ EngGeom eng_geom(const EngPhase& P, int esz, int c) {
    const int v = c < P.umod ? 0 : 1;
    int ucu = P.ucu_hi - v; if (ucu < 0) ucu = 0;
    EngGeom g = eng_geom_units(ucu, eng_ppuv(P, esz));
    g.inv2SA = P.inv2SA[v]; g.inv2SB = P.inv2SB[v];
    return g;
}

__device__ __forceinline__ unsigned eng_lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void eng_lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// one 1 KiB piece: every lane's 16 bytes from (rsrc, voff) to LDS at dst + 16 * lane.  The whole offset sits in the VGPR operand, the
// one the hardware bounds-checks: pieces past the end of the work and rows past the end of the matrix cost no memory access.
// "nt": a weight byte is read once per token.  Written in asm because the compiler would drain vmcnt before every LDS access of the
// loader (the sequence words) if it knew about the DMA.
__device__ __forceinline__ void eng_dma16(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen nt lds" :: "v"(voff), "s"(r), "s"(dst) : "memory");
}
typedef unsigned long long u64;
// a value every lane holds alike, moved to scalar registers (the compiler does not use scalar loads for memory that the kernel may write,
// and an "s" asm operand / a buffer descriptor needs SGPRs)
__device__ __forceinline__ int eng_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class P> __device__ __forceinline__ P* eng_uni(P* p) {
    const u64 v = (u64)(uintptr_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<P*>((uintptr_t)(((u64)hi << 32) | lo));
}
__device__ __forceinline__ void gran_st(u64* p, unsigned epoch, unsigned bits) { __hip_atomic_store(p, ((u64)epoch << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// a bounded wait on an LDS word; false = gave up (this wave timed out, or another wave of the workgroup did)
struct EngWait {
    unsigned* ctl; int* err; unsigned long long* wacc = nullptr;      // wacc: (tracing) where this wave adds up the time it spent in slow waits
    __device__ __forceinline__ bool aborted() const { return eng_lds_ld(ctl + ECTL_ABORT) != 0; }
    __device__ __forceinline__ void abort() const { eng_lds_st(ctl + ECTL_ABORT, 1u); __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ bool until_ge(const unsigned* word, unsigned v) const {
        if (eng_lds_ld(word) >= v) return true;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned n = 1;; ++n) {
            if (n < 8) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(6);     // (a poller shares its SIMD with waves that work: after ~0.3 us it backs off)
            if (eng_lds_ld(word) >= v) { if (wacc && (threadIdx.x & 63) == 0) *wacc += __builtin_amdgcn_s_memrealtime() - t0; return true; }
            if ((n & 63) == 0) {
                if (aborted()) return false;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { abort(); return false; }      // 20 ms of the 100 MHz clock
            }
        }
    }
};

// ------------------------------------------------------------------------------------------
// loader L of kEngLoaders: fills F = L, L + 2, ... of the launch
// ------------------------------------------------------------------------------------------
// a scale dword: lane's 4 bytes to LDS at dst + 4 * lane
__device__ __forceinline__ void eng_dma4(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen nt lds" :: "v"(voff), "s"(r), "s"(dst) : "memory");
}
// the same piece load with the column offset in the instruction's 12-bit immediate: the pieces of a fill that lie in one unit differ by
// multiples of 256 bytes, so one VGPR address serves the fill
// (the hardware adds the instruction offset to the LDS address as well -- LDS address = M0 + offset + 16 * lane --: M0 is set OFF bytes low)
template <int OFF>
__device__ __forceinline__ void eng_dma16i(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:%3 nt lds" :: "v"(voff), "s"(r), "s"(dst - OFF), "i"(OFF) : "memory");
}
template <int QT>
__device__ __forceinline__ void eng_loader(const EngArgs& a, char* lds, const int L) {
    using T = QTraits<QT>;
    const int lane = threadIdx.x & 63, c = blockIdx.x, ncu = gridDim.x;
    constexpr int nslot = kEngSlots;
    unsigned* ctl = reinterpret_cast<unsigned*>(lds + nslot * kEngSlotBytes);
    const EngWait wt{ctl, a.err, a.trace ? a.trace + c * kEngTrace + 256 + 8 * L + 7 : nullptr};
    int F0 = 0;
    // A fill = 8 weight pieces + 2 scale dword loads = 10 vector-memory instructions, kEngDepth fills of this loader in flight (12 per CU, ~100 KiB).
    // The loader is a lone wave that issues an instruction every ~6 cycles and must turn a fill around in ~1 us: everything per fill is
    // incremental scalar arithmetic, the phase descriptors come from the kernel-argument segment (scalar loads), and a fill that lies inside
    // one unit (the rule) needs ONE vector address per matrix -- the column offsets are instruction immediates.  (The first versions spent
    // 1.7 us of instructions per fill: divisions by run-time values, per-piece address arithmetic, per-lane scale decodes.)
    constexpr int kDepth = 2;
    int ps0 = 0, ps1 = 0, ps2 = 0; unsigned pv0 = 0, pv1 = 0, pv2 = 0; int npend = 0;   // fills issued and not yet published, oldest first: (slot, sequence to publish)
    auto publish_oldest = [&]() { if (lane == 0) eng_lds_st(ctl + ECTL_FILL + ps0, pv0); ps0 = ps1; pv0 = pv1; ps1 = ps2; pv1 = pv2; --npend; };
    auto drain = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); while (npend > 0) publish_oldest(); };
    // this loader's fills are F = L, L + 4, ...: their slot F % 14 and sequence F / 14 advance incrementally
    int ownF = L, slot = L % nslot; unsigned seq = (unsigned)(L / nslot);
    if (a.trace && lane == 0) a.trace[c * kEngTrace + 256 + 8 * L] = __builtin_amdgcn_s_memrealtime();
    // The CU's memory pipeline is a FIFO: an activation requested behind a ring of weight fills returns behind them (+4 us on the first
    // prologue, measured).  The first fill waits until the consumers hold the launch's first activation.
    if (a.ph[0].pro < EPRO_XQ && !wt.until_ge(ctl + ECTL_XREQ, kEngConsumers)) return;      // (a pre-quantized activation is a few KB: not worth the wait)
    for (int ph = 0; ph < a.nph; ++ph) {
        const EngPhase& P = a.ph[ph];
        const bool two = P.epi == EPI_SWIGLU;
        const EngGeom G = eng_geom(P, T::kEsz, c);
        const int rowbytes = P.K * T::kEsz, sn = P.K / kGroup, NM = two ? 2 : 1;
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.W), 0, NM * P.rows * rowbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.sW), 0, NM * P.rows * sn * 4, 0x00020000);
        const unsigned lane_w = (unsigned)((lane >> 4) * rowbytes + (lane & 15) * 16);
        // scale loads: lane (p = lane >> 4, r = (lane >> 2) & 3, g = lane & 3) of half h -> group g of row r of piece 4 h + p
        const unsigned lane_s = (unsigned)((((lane >> 2) & 3) * sn + (lane & 3)) * 4);
        const unsigned mat_w = two ? (unsigned)(P.rows * rowbytes) : 0u, mat_s = two ? (unsigned)(P.rows * sn * 4) : 0u;
        const unsigned ustep_w = (unsigned)(4 * ncu * kEngConsumers * rowbytes), ustep_s = (unsigned)(4 * ncu * kEngConsumers * sn * 4);
        int sL = 0, sS = 0, Fi = F0;                                     // next long / short round, next fill number
        while (sL < G.SA || sS < G.SB) {
            const bool lng = sS >= G.SB || (sL < G.SA && (2 * sL + 1) * G.SB <= (2 * sS + 1) * G.SA);   // the merge of EngGeom::fill_of
            const int s = lng ? sL : sS, w0 = lng ? 0 : G.rr, w1 = lng ? G.rr : kEngConsumers;
            if (lng) ++sL; else ++sS;
            // does this loader have a fill in the round?  (its fills are ownF, ownF + 4, ...; the round holds Fi .. Fi + (w1 - w0) - 1)
            const int nround = w1 - w0;
            if (ownF >= Fi + nround) { Fi += nround; continue; }
            const int j0 = s * kEngPieces;
            const int m0 = G.PPUV > 1 ? (int)__umulhi((unsigned)j0, P.inv_ppuv) : j0, rem0 = j0 - m0 * G.PPUV;   // the round's first piece: unit m0 of its stream, piece rem0 of the unit
            const int nu_round = lng ? G.ua : G.ua - 1;                   // units of the round's streams
            // a fill inside one unit that exists: the fast path (rows past the end of a matrix read as zero or as the next matrix's rows: the
            // epilogue drops their results)
            const bool fast = rem0 + kEngPieces <= G.PPUV && m0 < nu_round && !(kAblate && (a.ablate & 32));
            // per-lane scale offsets relative to the stream's unit m0 (fast path): piece 4 h + (lane >> 4)
            unsigned so_rel0 = 0, so_rel1 = 0;
            if (fast) {
                const int r0 = rem0 + (lane >> 4), r1 = r0 + 4;
                so_rel0 = (two ? (unsigned)((r0 & 1) ? mat_s : 0u) + (unsigned)((r0 >> 1) * 16) : (unsigned)(r0 * 16)) + lane_s;
                so_rel1 = (two ? (unsigned)((r1 & 1) ? mat_s : 0u) + (unsigned)((r1 >> 1) * 16) : (unsigned)(r1 * 16)) + lane_s;
            }
            const unsigned cbo = (unsigned)((two ? rem0 >> 1 : rem0) * 256);   // fast path: byte offset of the fill's first column block in a row
            while (ownF < Fi + nround) {
                const int w = w0 + (ownF - Fi), F = ownF;
                if (eng_lds_ld(ctl + ECTL_FREE + slot) < seq) {
                    drain();                                          // nothing to issue anyway: what has landed becomes visible now
                    if (!wt.until_ge(ctl + ECTL_FREE + slot, seq)) return;
                }
                const unsigned dst = (unsigned)(uintptr_t)(lds + slot * kEngSlotBytes);
                // while this CU's consumers sweep granules, one fill in flight per loader: their polls queue behind whatever is requested here
                if (eng_lds_ld(ctl + ECTL_GATHER) != 0) drain();
                const unsigned ub_w = (unsigned)(4 * (c + ncu * w) * rowbytes) + (unsigned)m0 * ustep_w, ub_s = (unsigned)(4 * (c + ncu * w) * sn * 4) + (unsigned)m0 * ustep_s;
                if (fast) {
                    if (!(kAblate && (a.ablate & 2))) { eng_dma4(rS, so_rel0 + ub_s, dst + kEngSlotW); eng_dma4(rS, so_rel1 + ub_s, dst + kEngSlotW + 256); }
                    const unsigned v1 = lane_w + ub_w + cbo;
                    if (two) {
                        const unsigned v3 = v1 + mat_w;
                        eng_dma16i<0>(rW, v1, dst); eng_dma16i<0>(rW, v3, dst + 1024); eng_dma16i<256>(rW, v1, dst + 2048); eng_dma16i<256>(rW, v3, dst + 3072);
                        eng_dma16i<512>(rW, v1, dst + 4096); eng_dma16i<512>(rW, v3, dst + 5120); eng_dma16i<768>(rW, v1, dst + 6144); eng_dma16i<768>(rW, v3, dst + 7168);
                    } else {
                        eng_dma16i<0>(rW, v1, dst); eng_dma16i<256>(rW, v1, dst + 1024); eng_dma16i<512>(rW, v1, dst + 2048); eng_dma16i<768>(rW, v1, dst + 3072);
                        eng_dma16i<1024>(rW, v1, dst + 4096); eng_dma16i<1280>(rW, v1, dst + 5120); eng_dma16i<1536>(rW, v1, dst + 6144); eng_dma16i<1792>(rW, v1, dst + 7168);
                    }
                } else {
                    // the general path: the fill crosses a unit end, runs past the stream's last unit, or units are shorter than a fill
                    const int nu = G.units_of(w);
                    int m = m0, rem = rem0; unsigned ubw = ub_w;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int off = 4 * h + (lane >> 4);
                        int mm, rr;
                        if (G.PPUV >= kEngPieces) { const int t = rem + off; const bool wr = t >= G.PPUV; mm = m + (wr ? 1 : 0); rr = wr ? t - G.PPUV : t; }
                        else { const int t = rem + off; const int q = t / G.PPUV; mm = m + q; rr = t - q * G.PPUV; }
                        const int cb = two ? rr >> 1 : rr;
                        const int row = 4 * (c + ncu * (w + kEngConsumers * mm)) + ((lane >> 2) & 3);
                        const unsigned so = (mm < nu && row < P.rows) ? ub_s + (unsigned)(mm - m) * ustep_s + ((two && (rr & 1)) ? mat_s : 0u) + (unsigned)(cb * 16) + lane_s : 0x80000000u;
                        if (!(kAblate && (a.ablate & 2))) eng_dma4(rS, so, dst + kEngSlotW + h * 256);
                    }
#pragma unroll
                    for (int p = 0; p < kEngPieces; ++p) {
                        const int cb = two ? rem >> 1 : rem;
                        const unsigned base = m < nu ? ubw + ((two && (rem & 1)) ? mat_w : 0u) + (unsigned)(cb * 256) : 0x80000000u;
                        eng_dma16(rW, lane_w + base, dst + p * 1024);
                        if (++rem == G.PPUV) { rem = 0; ++m; ubw += ustep_w; }
                    }
                }
                // kDepth fills in flight: with kDepth - 1 older ones pending, the oldest has landed once at most (kDepth - 1) * 10 loads are outstanding
                if (kAblate && (a.ablate & 4)) drain();
                if (npend == kDepth - 1) {
                    if constexpr (kDepth == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");   // (kDepth - 1) * 10
                    publish_oldest();
                }
                static_assert(kDepth == 2 || kDepth == 3, "the vmcnt immediates above");
                if (npend == 0) { ps0 = slot; pv0 = seq + 1; } else if (npend == 1) { ps1 = slot; pv1 = seq + 1; } else { ps2 = slot; pv2 = seq + 1; }
                ++npend;
                ownF += kEngLoaders; slot += kEngLoaders; if (slot >= nslot) { slot -= nslot; ++seq; }
            }
            Fi += nround;
        }
        F0 += G.nfill;
        if (a.trace && lane == 0 && ph < 6) a.trace[c * kEngTrace + 256 + 8 * L + 1 + ph] = __builtin_amdgcn_s_memrealtime();   // the phase's last fill is issued
    }
    drain();
}

// ------------------------------------------------------------------------------------------
// consumer w of kEngConsumers
// ------------------------------------------------------------------------------------------
struct EngCons {
    char* lds; unsigned* ctl; EngWait wt; EngLds LY;
    int lane, w, c, ncu, nsync, ph;
    bool ok;
    unsigned long long* trace;                // [workgroup][kEngTrace] (100 MHz clock)
    __device__ __forceinline__ void stamp(int k) const { if (trace && lane == 0 && k < 14) trace[c * kEngTrace + 16 * w + k] = __builtin_amdgcn_s_memrealtime(); }
    // meet the other consumer waves (LDS counter; the loaders never take part)
    __device__ __forceinline__ void sync4() {   // ("4": the first version had 4 consumer waves)
        ++nsync;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(ctl + ECTL_SYNC, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!wt.until_ge(ctl + ECTL_SYNC, (unsigned)(kEngConsumers * nsync))) ok = false;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};

// spin bookkeeping of the granule sweeps (global memory): give up after 20 ms
struct EngSpin {
    unsigned long long t0; unsigned n;
    __device__ __forceinline__ EngSpin() : t0(0), n(0) {}
    __device__ __forceinline__ bool fail(const EngWait& wt) {
        if (n == 0) t0 = __builtin_amdgcn_s_memrealtime();
        if ((++n & 15) == 0) {
            if (wt.aborted()) return true;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { wt.abort(); return true; }
        }
        __builtin_amdgcn_s_sleep(1);
        return false;
    }
};

// rmsnorm + quantize of the n values staged in the chain strips (gemv_prologue's PRO_RMSNORM_QUANT, on the consumer waves): wave c < 4 runs the
// reference's strided lane c (sq_chain_spec), then every lane scales and quantizes its 4-element pieces.
template <int QT>
__device__ __forceinline__ void eng_norm_quant(EngCons& E, const int n, const float* norm_w, const bool with_norm, const float4 (&nwp)[4]) {
    using T = QTraits<QT>;
    float* stage = reinterpret_cast<float*>(E.lds + E.LY.off_stage);
    float* red = reinterpret_cast<float*>(E.ctl + ECTL_RED);
    char* xq = E.lds + E.LY.off_xq; float* xs = reinterpret_cast<float*>(E.lds + E.LY.off_xs);
    const int bs = chain_bshift(n), B = 1 << bs, LS = B + 4, CS = 64 * LS;
    const int T4 = E.w * 64 + E.lane;                                  // lane index over the consumer waves: owns elements 4 T4 + 4 kEngCL i
    float r = 1.0f;
    if (with_norm) {
        if (E.w < 4) {
            __builtin_amdgcn_s_setprio(3);                             // the whole CU waits for these four waves
            const float l = sq_chain_spec(stage + E.w * CS, bs);       // simd::square_sum's lane w (x86_simd.cpp:942-960)
            __builtin_amdgcn_s_setprio(0);
            if (E.lane == 0) red[E.w] = l;
        }
        E.sync4();
        if (E.w == 1) E.stamp(3 + 3 * E.ph);                          // (tracing) consumer 1: the chains are done
        const float ss = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, red[0]), red[1]), red[2]), red[3]);
        r = rms_scale(ss, n);
    }
    const int rounds = (n + 4 * kEngCL - 1) / (4 * kEngCL);
#pragma unroll 4
    for (int i = 0; i < rounds; ++i) {
        const int e = 4 * T4 + 4 * kEngCL * i, k = T4 + kEngCL * i;             // chain element k of every strip = x[4 k + c]
        const bool act = e < n;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) {
            const int o = (k >> bs) * LS + (k & (B - 1));
            v = make_float4(stage[o], stage[CS + o], stage[2 * CS + o], stage[3 * CS + o]);
            if (with_norm) {   // multiply_avx256 (x86_simd.cpp:1360-1372): (x*w)*r
                const float4 wv = i < 4 ? nwp[i < 4 ? i : 0] : *reinterpret_cast<const float4*>(norm_w + e);   // (the first 4 rounds were fetched before the vector arrived)
                v.x = __fmul_rn(__fmul_rn(v.x, wv.x), r); v.y = __fmul_rn(__fmul_rn(v.y, wv.y), r);
                v.z = __fmul_rn(__fmul_rn(v.z, wv.z), r); v.w = __fmul_rn(__fmul_rn(v.w, wv.w), r);
            }
        }
        const float mx = row16_max(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        const float sc = __fdiv_rn(mx, T::kF);                         // quant::quantize (quant_operators.cpp:26-47)
        if (act) {
            const int q0 = quant_elem(v.x, sc), q1 = quant_elem(v.y, sc), q2 = quant_elem(v.z, sc), q3 = quant_elem(v.w, sc);
            if constexpr (QT == QT_INT8) {
                *reinterpret_cast<uint32_t*>(xq + e) = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
            } else {
                uint2 pk; pk.x = (uint32_t)(q0 & 0xffff) | ((uint32_t)(q1 & 0xffff) << 16); pk.y = (uint32_t)(q2 & 0xffff) | ((uint32_t)(q3 & 0xffff) << 16);
                *reinterpret_cast<uint2*>(xq + (size_t)e * 2) = pk;
            }
            if ((E.lane & 15) == 0) xs[e / kGroup] = sc;
        }
    }
    E.sync4();
}

// stage 4 consecutive values (elements e .. e + 3, e = 4 k) into the chain strips
__device__ __forceinline__ void eng_stage4(float* stage, int k, int bs, const float4& v) {
    const int B = 1 << bs, LS = B + 4, CS = 64 * LS, o = (k >> bs) * LS + (k & (B - 1));
    stage[o] = v.x; stage[CS + o] = v.y; stage[2 * CS + o] = v.z; stage[3 * CS + o] = v.w;
}

template <int QT>
__device__ __forceinline__ void eng_prologue(EngCons& E, const EngArgs& a, const EngPhase& P, const unsigned epoch_in) {
    using T = QTraits<QT>;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const int n = P.K, T4 = E.w * 64 + E.lane;
    float* stage = reinterpret_cast<float*>(E.lds + E.LY.off_stage);
    char* xq = E.lds + E.LY.off_xq; float* xs = reinterpret_cast<float*>(E.lds + E.LY.off_xs);
    // the activation buffer is rewritten below: every consumer wave must be done with the previous phase's dots (a wave that ran out of
    // rows early would otherwise overwrite what the others still read)
    E.sync4();
    // (the loaders' first fill waits for ECTL_XREQ: the launch's first activation is fetched through an empty memory pipeline)
    auto xreq = [&]() { if (E.lane == 0) __hip_atomic_fetch_add(E.ctl + ECTL_XREQ, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto gather = [&](int d) { if (E.lane == 0) __hip_atomic_fetch_add(E.ctl + ECTL_GATHER, (unsigned)d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    if (P.pro == EPRO_XQ) {
        const int nb16 = n * T::kEsz / 16;
        int4 v0 = make_int4(0, 0, 0, 0); float s0 = 0.f;
        if (T4 < nb16) v0 = reinterpret_cast<const int4*>(P.xq)[T4];
        if (T4 < n / kGroup) s0 = P.xs[T4];
        asm volatile("" :: "v"(v0.x), "v"(s0));                        // landed
        xreq();
        if (T4 < nb16) reinterpret_cast<int4*>(xq)[T4] = v0;
        if (T4 < n / kGroup) xs[T4] = s0;
        for (int i = T4 + kEngCL; i < nb16; i += kEngCL) reinterpret_cast<int4*>(xq)[i] = reinterpret_cast<const int4*>(P.xq)[i];
        for (int g = T4 + kEngCL; g < n / kGroup; g += kEngCL) xs[g] = P.xs[g];
        E.sync4();
        return;
    }
    if (P.pro == EPRO_GRAN_HD) {
        // the quantized hd: per 64-group 64 * esz / 4 dwords + 1 scale, each an 8-byte granule {value, tag}
        constexpr int DPG = 16 * T::kEsz, GPG = DPG + 1;
        const int total = (n / kGroup) * GPG;
        gather(1);
        for (int t0 = 0; t0 < total; t0 += kEngCL * 8) {
            u64 g[8];
            EngSpin sp;
            while (true) {
                bool good = true;
#pragma unroll
                for (int k = 0; k < 8; ++k) { const int t = t0 + T4 + kEngCL * k; g[k] = t < total ? __hip_atomic_load(a.ghq + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (u64)epoch_in << 32; }
#pragma unroll
                for (int k = 0; k < 8; ++k) good = good && (unsigned)(g[k] >> 32) == epoch_in;
                if (__all(good)) break;
                if (sp.fail(E.wt)) { E.ok = false; break; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int t = t0 + T4 + kEngCL * k;
                if (t < total) { const int grp = t / GPG, kk = t - grp * GPG; if (kk < DPG) reinterpret_cast<unsigned*>(xq)[grp * DPG + kk] = (unsigned)g[k]; else reinterpret_cast<unsigned*>(xs)[grp] = (unsigned)g[k]; }
            }
        }
        gather(-1);
        E.sync4();
        return;
    }
    // fp32 sources: stage the vector in the chain strips (zero-padded), then (rmsnorm,) quantize
    // the norm weights first: requested behind the loaders' fills they would come back microseconds later, on the critical path
    float4 nwp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int e = 4 * T4 + 4 * kEngCL * i; nwp[i] = (P.pro != EPRO_X_Q && e < n) ? *reinterpret_cast<const float4*>(P.norm_w + e) : make_float4(0.f, 0.f, 0.f, 0.f); }
    const int bs = chain_bshift(n), slots = 64 << bs;                  // chain elements per strip incl. padding
    const int rounds = (slots + kEngCL - 1) / kEngCL;
    if (P.pro == EPRO_GRAN_RMS) {
        // the residual stream: n granules {value, tag}; a lane's 4 elements = 32 bytes = two 16-byte coherent loads (granules are written by
        // single 8-byte stores; 16-byte halves arrive untorn)
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(a.gx1, 0, n * 8, 0x00020000);
        gather(1);
        for (int i0 = 0; i0 < rounds; i0 += 4) {
            v4u g[4][2];
            EngSpin sp;
            while (true) {
                bool good = true;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = T4 + kEngCL * (i0 + i);                 // elements 4k .. 4k+3 = granules 4k .. 4k+3
                    if (4 * k < n) {
                        g[i][0] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rg, k * 32, 0, kAuxCoherent));
                        g[i][1] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rg, k * 32 + 16, 0, kAuxCoherent));
                    } else { g[i][0] = v4u{0u, epoch_in, 0u, epoch_in}; g[i][1] = g[i][0]; }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) good = good && g[i][0].y == epoch_in && g[i][0].w == epoch_in && g[i][1].y == epoch_in && g[i][1].w == epoch_in;
                if (__all(good)) break;
                if (sp.fail(E.wt)) { E.ok = false; break; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = T4 + kEngCL * (i0 + i);
                if (k < slots) eng_stage4(stage, k, bs, make_float4(__uint_as_float(g[i][0].x), __uint_as_float(g[i][0].z), __uint_as_float(g[i][1].x), __uint_as_float(g[i][1].z)));
            }
        }
        gather(-1);
    } else {
        for (int i0 = 0; i0 < rounds; i0 += 4) {
            float4 xv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int k = T4 + kEngCL * (i0 + i); xv[i] = 4 * k < n ? *reinterpret_cast<const float4*>(P.x + 4 * k) : make_float4(0.f, 0.f, 0.f, 0.f); }
            if (i0 == 0) { asm volatile("" :: "v"(xv[0].x), "v"(xv[1].x), "v"(xv[2].x), "v"(xv[3].x)); xreq(); }   // landed
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int k = T4 + kEngCL * (i0 + i); if (k < slots) eng_stage4(stage, k, bs, xv[i]); }
        }
    }
    if (E.w == 0) E.stamp(3 + 3 * E.ph);                              // (tracing) consumer 0: its part of the vector is staged
    E.sync4();
    if (E.w == 2) E.stamp(3 + 3 * E.ph);                              // (tracing) consumer 2: everybody's part is staged
    eng_norm_quant<QT>(E, n, P.norm_w, P.pro != EPRO_X_Q, nwp);
}

// the reference's chain over a piece's 4 groups: acc = fma(sW[g] * sX[g], float(dot_g), acc), g ascending (quant_operators.cpp:274-276).  fd: this
// lane's group dot; group g's dot sits in lane 4 g of the DPP row and is fetched by the DPP operand of v_fmac (a fused multiply-add; the
// compiler's own choice for __fmaf_rn): 4 dependent instructions per piece.  s_nop 1: the two wait states between a VALU write of fd and its
// DPP read, which the compiler does not insert for an asm statement.
__device__ __forceinline__ void eng_chain4(float& acc, const float fd, const float4& sg) {
    // (four statements, so that the scheduler can put another piece's dots between the dependent FMAs)
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(fd), "v"(sg.x));
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(fd), "v"(sg.y));
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(fd), "v"(sg.z));
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(fd), "v"(sg.w));
}
// `n` consecutive pieces (TWO: pairs {W1 piece, W3 piece}) of ONE unit, starting at piece address wq / scale records sq / activation chunk xa /
// activation scales xsa: no bookkeeping inside, so that the unrolled body is LDS reads at immediate offsets, dots and the chain
template <bool TWO>
__device__ __forceinline__ void eng_run(const char* wq, const char* sq, const char* xa, const char* xsa, const int n, float& acc, float& acc3) {
    // Two register sets, A and B: while one piece's dots and chain run, the other's six LDS reads are in flight (a lone wave per SIMD
    // has nobody to hide the ~130-cycle LDS latency behind: the plain loop spent half its time in s_waitcnt lgkmcnt(0)).  The read-ahead
    // runs up to three pieces past the run: inside the ring or the regions behind it, never consumed.
    struct Ops { v4i x, w1, w3; float4 sx, s1, s3; };
    auto ld = [&](Ops& o, int i) {
        o.x = *reinterpret_cast<const v4i*>(xa + i * 256);
        o.sx = *reinterpret_cast<const float4*>(xsa + i * 16);
        if constexpr (TWO) {
            o.w1 = *reinterpret_cast<const v4i*>(wq + (2 * i) * 1024); o.w3 = *reinterpret_cast<const v4i*>(wq + (2 * i + 1) * 1024);
            o.s1 = *reinterpret_cast<const float4*>(sq + (2 * i) * 64); o.s3 = *reinterpret_cast<const float4*>(sq + (2 * i + 1) * 64);
        } else {
            o.w1 = *reinterpret_cast<const v4i*>(wq + i * 1024);
            o.s1 = *reinterpret_cast<const float4*>(sq + i * 64);
        }
    };
    struct Dots { float f1, f3; float4 g1, g3; };
    auto dots = [&](const Ops& o) {
        Dots d;
        d.f1 = (float)quad_sum(dot16_i8(o.w1, o.x, 0));                                    // exact int32 group dot -> fp32, as "s * dot" does
        d.g1 = make_float4(__fmul_rn(o.s1.x, o.sx.x), __fmul_rn(o.s1.y, o.sx.y), __fmul_rn(o.s1.z, o.sx.z), __fmul_rn(o.s1.w, o.sx.w));   // s = sW * sX
        if constexpr (TWO) {
            d.f3 = (float)quad_sum(dot16_i8(o.w3, o.x, 0));
            d.g3 = make_float4(__fmul_rn(o.s3.x, o.sx.x), __fmul_rn(o.s3.y, o.sx.y), __fmul_rn(o.s3.z, o.sx.z), __fmul_rn(o.s3.w, o.sx.w));
        }
        return d;
    };
    auto chain = [&](const Dots& d) { eng_chain4(acc, d.f1, d.g1); if constexpr (TWO) eng_chain4(acc3, d.f3, d.g3); };
    // piece i's chain (dependent FMAs) is issued next to piece i + 1's dots (independent), piece i + 2's and i + 3's reads are in flight
    Ops A, B;
    ld(A, 0); ld(B, 1);
    Dots dA = dots(A);
    int i = 0;
#pragma unroll 1
    for (; i + 2 <= n; i += 2) {
        ld(A, i + 2);
        const Dots dB = dots(B);
        chain(dA);
        ld(B, i + 3);
        dA = dots(A);
        chain(dB);
    }
    if (i < n) chain(dA);
}

template <int QT>
__device__ __forceinline__ void eng_consumer(const EngArgs& a, char* lds, const int w) {
    using T = QTraits<QT>;
    static_assert(QT == QT_INT8, "the engine's lane layout is written for 1-byte elements (4 quant groups per 256-byte piece row)");
    EngCons E;
    E.lds = lds; E.lane = threadIdx.x & 63; E.w = w; E.c = blockIdx.x; E.ncu = gridDim.x; E.nsync = 0; E.ok = true; E.trace = a.trace;
    E.stamp(0);
    constexpr int nslot = kEngSlots;
    const int lane = E.lane, c = E.c, ncu = E.ncu;
    E.ctl = reinterpret_cast<unsigned*>(lds + nslot * kEngSlotBytes);
    E.wt = EngWait{E.ctl, a.err, a.trace ? a.trace + E.c * kEngTrace + 16 * w + 15 : nullptr};
    {   // the LDS layout depends on the largest K / the largest normalised vector of the launch
        int kmax = 0, nnorm = 0;
        for (int ph = 0; ph < a.nph; ++ph) { const EngPhase& P = a.ph[ph]; if (P.K > kmax) kmax = P.K; if (P.pro != EPRO_XQ && P.pro != EPRO_GRAN_HD && P.K > nnorm) nnorm = P.K; }
        E.LY = eng_lds_layout(nslot, kmax, T::kEsz, nnorm);
    }
    const unsigned base = (unsigned)eng_uni((int)*a.base_ptr);
    const int pos = eng_uni(*a.pos_ptr);
    // the residual rows this lane owns (lanes 0..3 of a consumer: rows 4u .. 4u+3 of its units in the dim-row phases)
    float x1own[kEngMaxOwn];
#pragma unroll
    for (int m = 0; m < kEngMaxOwn; ++m) x1own[m] = 0.f;
    for (int ph = 0; ph < a.nph; ++ph) {
        const EngPhase& P = a.ph[ph];
        if (P.epi == EPI_RESIDUAL) {
#pragma unroll
            for (int m = 0; m < kEngMaxOwn; ++m) { const int row = 4 * (c + ncu * (w + kEngConsumers * m)) + lane; if (lane < 4 && row < P.rows) x1own[m] = a.x1[row]; }
            break;
        }
    }
    const char* xq = lds + E.LY.off_xq; const float* xs = reinterpret_cast<const float*>(lds + E.LY.off_xs);
    int F0 = 0;
    unsigned long long tacc_fill = 0, tacc_run = 0, tacc_epi = 0, tacc_slots = 0;   // (tracing) 100 MHz ticks waiting for fills / in the dots / in epilogues; slots
    for (int ph = 0; ph < a.nph && E.ok; ++ph) {
        const EngPhase& P = a.ph[ph];
        const unsigned epoch = base + (unsigned)P.index + 1u;           // of this phase's results; the previous phase's: epoch - 1
        const bool two = P.epi == EPI_SWIGLU;
        const EngGeom G = eng_geom(P, T::kEsz, c);
        const int nu = G.units_of(w), ns = G.slots_of(w);
        if (kAblate && (a.ablate & 16)) { if (ph == 0 && lane == 0) __hip_atomic_fetch_add(E.ctl + ECTL_XREQ, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // (pure streaming: no prologue)
        else { E.ph = ph; eng_prologue<QT>(E, a, P, epoch - 1u); }
        if (!E.ok) break;
        E.stamp(1 + 3 * ph);
        float acc = 0.f, acc3 = 0.f;
        int m = 0, rem = 0;
        for (int s = 0; s < ns; ++s) {
            const int F = F0 + G.fill_of(s, w), slot = F % nslot;
            const unsigned seq = (unsigned)(F / nslot) + 1u;
            const unsigned long long tf0 = E.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
            if (!E.wt.until_ge(E.ctl + ECTL_FILL + slot, seq)) { E.ok = false; break; }
            if (E.trace) { tacc_fill += __builtin_amdgcn_s_memrealtime() - tf0; ++tacc_slots; }
            {
                const char* sl = lds + slot * kEngSlotBytes;
                const char* wp = sl + lane * 16;
                const char* sp = sl + kEngSlotW + (lane >> 4) * 16;     // the 16-byte scale record of this lane's row: + 64 per piece
                int p = 0;
                while (p < kEngPieces && m < nu) {                      // (wave-uniform) runs of pieces of one unit
                    int run = G.PPUV - rem; if (run > kEngPieces - p) run = kEngPieces - p;
                    const int cb0 = two ? rem >> 1 : rem;
                    const char* xa = xq + cb0 * 256 + (lane & 15) * 16;
                    const char* xsa = reinterpret_cast<const char*>(xs) + cb0 * 16;
                    const unsigned long long tr0 = E.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
                    if (kAblate && (a.ablate & 1)) {}
                    else if (two) eng_run<true>(wp + p * 1024, sp + p * 64, xa, xsa, run >> 1, acc, acc3);
                    else eng_run<false>(wp + p * 1024, sp + p * 64, xa, xsa, run, acc, acc3);
                    if (E.trace) tacc_run += ((__builtin_amdgcn_s_memrealtime() - tr0) << 16) + (unsigned)run;   // (tracing) time in the dots, pieces done
                    p += run; rem += run;
                    const unsigned long long te0 = E.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
                    if (rem == G.PPUV && kAblate && (a.ablate & 16)) { acc = 0.f; acc3 = 0.f; rem = 0; ++m; }
                    else if (rem == G.PPUV) {
                            // ---- the unit's 4 rows are complete: row r's value sits in all lanes of DPP row r
                            const int u = c + ncu * (w + kEngConsumers * m), row0 = 4 * u;
                            const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 0)), a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 16));
                            const float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 32)), a3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 48));
                            const float v = lane == 0 ? a0 : lane == 1 ? a1 : lane == 2 ? a2 : a3;
                            const int row = row0 + lane;
                            const bool rv = lane < 4 && row < P.rows;
                            if (P.epi == EPI_STORE) {
                                if (rv) P.out[row] = v;
                            } else if (P.epi == EPI_RESIDUAL) {
                                float nv = 0.f;
#pragma unroll
                                for (int mm = 0; mm < kEngMaxOwn; ++mm) if (mm == m) { nv = __fadd_rn(x1own[mm], v); x1own[mm] = nv; }   // o.add(tmp, offset) transformer.cpp:465,493
                                if (rv) { P.out[row] = nv; if (P.gran_out) gran_st(a.gx1 + row, epoch, __float_as_uint(nv)); }
                            } else if (P.epi == EPI_SWIGLU) {
                                const float b0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc3), 0)), b1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc3), 16));
                                const float b2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc3), 32)), b3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc3), 48));
                                const float v3 = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
                                // o1.swiglu(o3) (transformer.cpp:481): here only when the results stay in memory; inside a launch the owner of the row's quant group
                                // evaluates it for 64 rows at once (4 lanes of a lone wave in double precision cost ~1 us per unit on this wave's critical path)
                                if (rv) { if (P.gran_out) { gran_st(a.ghd + 2 * row, epoch, __float_as_uint(v)); gran_st(a.ghd + 2 * row + 1, epoch, __float_as_uint(v3)); } else P.out[row] = swiglu_elem(v, v3); }
                            } else {   // EPI_ROPE_KV: rows (2i, 2i+1) of [Wq; Wk; Wv]: RoPE on q and k, k and v appended to the cache
                                const float x0 = lane == 0 ? a0 : a2, x1 = lane == 0 ? a1 : a3;
                                const int prow = row0 + lane;                                  // lanes 0 and 2: the pair's first row
                                if ((lane == 0 || lane == 2) && prow < P.rows) {
                                    const int hs = P.hs;
                                    if (prow < P.dim + P.kv_dim) {
                                        const int rr = prow < P.dim ? prow : prow - P.dim;
                                        const int h = rr / hs, d = rr - h * hs;
                                        const float rc = a.rope_cos[(size_t)pos * (hs / 2) + d / 2], rs = a.rope_sin[(size_t)pos * (hs / 2) + d / 2];
                                        float o0, o1;
                                        rope_pair(x0, x1, rc, rs, o0, o1);
                                        if (prow < P.dim) { P.out[prow] = o0; P.out[prow + 1] = o1; }
                                        else { float* kp = P.kcache + ((size_t)h * P.max_seq + pos) * hs + d; kp[0] = o0; kp[1] = o1; }
                                    } else {
                                        const int rr = prow - P.dim - P.kv_dim;
                                        const int h = rr / hs, d = rr - h * hs;
                                        float* vp = P.vcache + ((size_t)h * P.max_seq + pos) * hs + d; vp[0] = x0; vp[1] = x1;
                                    }
                                }
                            }
                            acc = 0.f; acc3 = 0.f; rem = 0; ++m;
                            if (E.trace) tacc_epi += __builtin_amdgcn_s_memrealtime() - te0;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave's reads of the slot have returned
            if (lane == 0) eng_lds_st(E.ctl + ECTL_FREE + slot, seq);
        }
        F0 += G.nfill;
        if (!E.ok) break;
        E.stamp(2 + 3 * ph);
        if (P.epi == EPI_SWIGLU && P.gran_out && w == kEngConsumers - 1 && !(kAblate && (a.ablate & 16))) {
            // hop 1 of the hd hand-off: this CU owns the quant groups g = c, c + nCU, ...: gather the group's 64 pairs of chain values (lane = row),
            // SwiGLU, qh.quantize(hd) (transformer.cpp:149; quant_operators.cpp:26-47: max order-free, then the element step), publish 16 dwords + scale
            constexpr int DPG = 16 * T::kEsz, GPG = DPG + 1;
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.ghd, 0, P.rows * 16, 0x00020000);
            for (int g = c; g < P.rows / kGroup; g += ncu) {
                v4u x;                                                  // {W1 dot chain, tag, W3 dot chain, tag} of row 64 g + lane
                EngSpin sp;
                while (true) {
                    x = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rh, (g * kGroup + lane) * 16, 0, kAuxCoherent));
                    if (__all(x.y == epoch && x.w == epoch)) break;
                    if (sp.fail(E.wt)) { E.ok = false; break; }
                }
                if (!E.ok) break;
                const float hv = swiglu_elem(__uint_as_float(x.x), __uint_as_float(x.z));      // o1.swiglu(o3) transformer.cpp:481
                P.out[g * kGroup + lane] = hv;
                const float mx = wave_max(fabsf(hv));
                const float sc = __fdiv_rn(mx, T::kF);
                const int q = quant_elem(hv, sc);
                const unsigned pk = (unsigned)quad_sum((q & 0xff) << (8 * (lane & 3)));       // the quad's 4 bytes (disjoint bits: the sum is the OR)
                if ((lane & 3) == 0) gran_st(a.ghq + g * GPG + (lane >> 2), epoch, pk);
                if (lane == 0) gran_st(a.ghq + g * GPG + DPG, epoch, __float_as_uint(sc));
            }
            E.stamp(3 + 3 * ph);
        }
    }
    if (E.trace && lane == 0) { unsigned long long* t = E.trace + c * kEngTrace; t[16 * w + 14] = tacc_run; t[320 + 4 * w] = tacc_fill; t[320 + 4 * w + 1] = tacc_run >> 16; t[320 + 4 * w + 2] = tacc_epi; t[320 + 4 * w + 3] = tacc_slots; }
}

// grid = CUs (every workgroup resident: the consumers of all CUs wait for each other's granules), block = kEngBlock
template <int QT>
__global__ void __launch_bounds__(kEngBlock) k_engine(const EngArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    unsigned* ctl = reinterpret_cast<unsigned*>(lds + kEngSlots * kEngSlotBytes);
    if (threadIdx.x < 64) ctl[threadIdx.x] = 0;
    __syncthreads();                                                     // the only workgroup barrier of the kernel
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < kEngLoaders) eng_loader<QT>(a, lds, wave);
    else eng_consumer<QT>(a, lds, wave - kEngLoaders);
}

} // namespace flm
