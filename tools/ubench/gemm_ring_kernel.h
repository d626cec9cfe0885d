// gemm_ring_kernel.h -- EXPERIMENT (round 3, not part of the product library): the batched prompt path's int8 GEMM as loader waves + consumer
// waves around an LDS ring, the decode engine's scheme.  Bit-identical to k_gemm_q8_mfma (it passed tests/test_gpu_configs.py's tile-kernel
// shapes while it was wired into launch_gemm) and NOT faster: profiles/r03_prefill_gemm_ring.txt has the measurements and what they say about
// the tile kernel's own bound.  Kept with its micro-benchmark (tools/ubench/gemm_ring.hip) so that the numbers can be reproduced.
#pragma once
#include "flm_kernels.h"
#pragma clang fp contract(off)

namespace flm {

// ------------------------------------------------------------------------------------------
// k_gemm_q8_ring<EPI, WT, WR, NB>: quant::matmul (quant_operators.cpp:252-284, 571-591) for a batch of token rows, int8.
//
// Why a second GEMM: the tile kernel (k_gemm_q8_mfma) stages its operands through registers into LDS and meets on a barrier per stage;
// its counters (profiles/r02_prefill_gemm_pmc.txt) show a wave spending a stage in sequence -- loads, ds_writes, barrier, LDS reads,
// MFMAs, a 48-96 instruction fp32 chain at one VALU issue per ~6.5 cycles -- with too few waves to cover it: 12-20 % of the int8 matrix peak.
// Here the two jobs are split between waves (the decode engine's scheme, flm_engine.h):
//   * ONE 1024-thread workgroup per CU, persistent over the blocks dealt to the CU;
//   * kGrLoaders = 4 loader waves move operands L2 -> LDS with LDS-DMA (buffer_load_dwordx4 ... lds: no registers, no ds_write, no VALU).
//     One DMA instruction = one MFMA operand fragment: lane (r = lane & 31, h = lane >> 5) fetches the 16 bytes 32 kk + 16 h .. of the
//     group in row r, and the hardware lays the wave's 64 pieces down in lane order -- exactly the order in which the consumers read them
//     back with one conflict-free ds_read_b128 (a fragment = 32 rows x 32 k-bytes = 1 KiB, the operand of one v_mfma_i32_32x32x32_i8).
//     The scales of the stage's tokens and rows follow as dword DMAs from the group-major copies (XsT, sWT);
//   * up to 12 consumer waves, wave (wt, wr) owning 32 tokens x NB fragments of 32 weight rows: per quant group it reads its fragments,
//     runs 2 NB MFMAs (the group's int32 dots, exact) and the reference's chain step acc = fma(sW * sX, float(dot), acc) on its 16 NB
//     results per lane -- groups ascending, so the result is the tile kernel's and the GEMV's, bit for bit;
//   * a STAGE = one quant group of the whole block (WT x 2 KiB of tokens, WR NB x 2 KiB of weight rows, their scales) = one ring slot; no
//     s_barrier anywhere: loader L publishes fillw[slot][L] = round once its pieces of the stage have landed (its own vmcnt), consumer w
//     publishes prog[w] = stages read; a loader refills a slot when every consumer has read it.  The loaders run ahead over block
//     boundaries, so a block's epilogue (stores, SwiGLU, RoPE) overlaps the next block's first stages.
// Block = 128 tokens x (32 NB WR) rows; blocks are dealt to the CUs so that the blocks sharing weight rows run on one XCD (one L2).
// ------------------------------------------------------------------------------------------
constexpr int kGrLoaders = 4, kGrBlock = 1024, kGrCtl = 1024, kGrLdsMax = 160 * 1024, kGrMaxSlots = 12;
template <int WT, int WR, int NB>
struct GrTile {
    static constexpr int NC = WT * WR, NF = WR * NB, TT = 32 * WT, TR = 32 * NF;
    static constexpr int kOpsA = 2 * WT, kOpsB = 2 * NF, kOpsSx = (TT + 63) / 64, kOpsSw = (TR + 63) / 64, kOps = kOpsA + kOpsB + kOpsSx + kOpsSw;
    // (a scale DMA writes all 64 lanes' dwords -- zeros for the lanes past the block: the scale areas are whole 256-byte runs)
    static constexpr int kOffB = WT * 2048, kOffSx = kOffB + NF * 2048, kOffSw = kOffSx + kOpsSx * 256, kSlot = kOffSw + kOpsSw * 256;
    static constexpr int NS0 = (kGrLdsMax - kGrCtl) / kSlot, NS = NS0 < kGrMaxSlots ? NS0 : kGrMaxSlots;
    static constexpr int kLds = kGrCtl + NS * kSlot;
    static_assert(NC + kGrLoaders <= 16 && NC <= 12 && NS >= 3, "one workgroup: loaders + consumers; a ring");
    static_assert(kSlot % 16 == 0, "slots keep 16-byte alignment");
};
// control words at the start of the LDS: pub[4] (stages of the launch whose pieces loader L has seen land; one 16-byte line), prog[16] from kGrOffProg, abort
constexpr int kGrOffProg = kGrMaxSlots * 16, kGrOffAbort = kGrOffProg + 64;
static_assert(kGrOffAbort + 4 <= kGrCtl, "control words fit");

// one fragment / one run of scales: lane's 16 (4) bytes from (rsrc, voff + soff) to LDS at dst + 16 (4) * lane.  Default cache policy: every
// CU re-reads the token rows, and the CUs of an XCD share weight rows -- these loads are meant to hit in L2.
__device__ __forceinline__ void gr_dma16(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(r), "s"(soff), "s"(dst) : "memory");
}
__device__ __forceinline__ void gr_dma4(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds" :: "v"(voff), "s"(r), "s"(soff), "s"(dst) : "memory");
}
template <int N> __device__ __forceinline__ void gr_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

struct GrSync {
    unsigned* ctl; int* err;
    __device__ __forceinline__ bool aborted() const { return eng_lds_ld(ctl + kGrOffAbort / 4) != 0; }
    __device__ __forceinline__ void abort() const { eng_lds_st(ctl + kGrOffAbort / 4, 1u); if (err) __hip_atomic_store(err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};

// which block does workgroup wg run in round k?  Blocks are numbered token block fastest; round k covers blocks [k nwg, (k + 1) nwg); inside a
// round the XCDs (workgroup id mod 8) take contiguous runs, so the token blocks of one row block -- the blocks that share weight rows -- meet in one L2.
__device__ __forceinline__ int gr_block_of(int wg, int nwg, int k) {
    const int per = nwg >> 3, rem = nwg & 7, xcd = wg & 7;
    return k * nwg + xcd * per + (xcd < rem ? xcd : rem) + (wg >> 3);
}

template <int EPI, int WT, int WR, int NB, int L>
__device__ __forceinline__ void gr_loader(const GemmArgs& a, char* lds, int* err, unsigned long long* trace, const int ablate) {
    using G = GrTile<WT, WR, NB>;
    constexpr bool TWO = EPI == EPI_SWIGLU;
    constexpr int TRH = TWO ? G::TR / 2 : G::TR;
    constexpr int kMine = (G::kOps - L + kGrLoaders - 1) / kGrLoaders;        // ops L, L + 4, ...
    constexpr unsigned kOOB = 0x80000000u;
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    unsigned* const ctl = reinterpret_cast<unsigned*>(lds);
    const GrSync sy{ctl, err};
    const unsigned rowbytes = (unsigned)a.n, rows_tot = (TWO ? 2u : 1u) * (unsigned)a.rows;
    const int sn = a.n / kGroup, ntb = (a.B + G::TT - 1) / G::TT, nrb = (a.rows + TRH - 1) / TRH, nblk = ntb * nrb;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, (int)(rows_tot * rowbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.Xq), 0, (int)((unsigned)a.B * rowbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rWs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.sWT), 0, (int)(rows_tot * sn * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rXs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.XsT), 0, (int)((unsigned)a.B * sn * 4), 0x00020000);
    const unsigned ring = (unsigned)(uintptr_t)(lds + kGrCtl);       // LDS byte address of slot 0 (the low 32 bits of a __shared__ pointer)
    const unsigned step_sx = (unsigned)a.B * 4, step_sw = rows_tot * 4;
    if (kAblate && (ablate & 16)) return;
    unsigned gs = 0;                                                   // stages issued by this loader = the launch's stage counter
    int pend_slot = -1; unsigned pend_round = 0;
    unsigned long long tr_t0 = 0, tr_wait = 0, tr_w0 = 0; unsigned tr_nwait = 0;
    if (kAblate && trace) tr_t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0;; ++k) {
        const int blk = gr_block_of(blockIdx.x, gridDim.x, k);
        if (k * (int)gridDim.x >= nblk) break;
        const bool live = blk < nblk;                                  // (a round's tail: nothing to do, nothing published, the consumers skip it too)
        if (!live) continue;
        const int r0 = (blk / ntb) * TRH, b0 = (blk % ntb) * G::TT;
        // tile row tr (fragment tr / 32) -> row of W, or -1
        auto wrow = [&](int tr) -> int {
            if (tr >= G::TR) return -1;
            const int f = tr >> 5, r = TWO ? r0 + (f >> 1) * 32 + (tr & 31) : r0 + tr;
            return r < a.rows ? (TWO ? (f & 1) * a.rows + r : r) : -1;
        };
        unsigned voff[kMine];
#pragma unroll
        for (int i = 0; i < kMine; ++i) {
            const int o = L + kGrLoaders * i;
            if (o < G::kOpsA) { const int t = b0 + (o >> 1) * 32 + l31; voff[i] = t < a.B ? (unsigned)t * rowbytes + (o & 1) * 32 + h * 16 : kOOB; }
            else if (o < G::kOpsA + G::kOpsB) { const int q = o - G::kOpsA, r = wrow((q >> 1) * 32 + l31); voff[i] = r >= 0 ? (unsigned)r * rowbytes + (q & 1) * 32 + h * 16 : kOOB; }
            else if (o < G::kOpsA + G::kOpsB + G::kOpsSx) { const int q = o - G::kOpsA - G::kOpsB, t = q * 64 + lane; voff[i] = (t < G::TT && b0 + t < a.B) ? (unsigned)(b0 + t) * 4 : kOOB; }
            else { const int q = o - G::kOpsA - G::kOpsB - G::kOpsSx, r = wrow(q * 64 + lane); voff[i] = r >= 0 ? (unsigned)r * 4 : kOOB; }
        }
        unsigned so_k = 0, so_sx = 0, so_sw = 0;
        for (int g = 0; g < sn; ++g, ++gs) {
            const unsigned slot = gs % G::NS, round = gs / G::NS;
            if (gs >= (unsigned)G::NS) {                               // the slot's previous stage gs - NS must have been read by every consumer
                const unsigned need = gs - G::NS + 1;
                unsigned long long t0 = 0;
                for (unsigned n = 0;; ++n) {
                    const unsigned p = lane < G::NC ? eng_lds_ld(ctl + kGrOffProg / 4 + lane) : 0xffffffffu;
                    if (__all(p >= need)) { if (kAblate && trace && n) { tr_wait += __builtin_amdgcn_s_memtime() - tr_w0; ++tr_nwait; } break; }
                    if (kAblate && trace && n == 0) tr_w0 = __builtin_amdgcn_s_memtime();
                    if (n == 0) {                                      // the ring is full: what this loader has in flight is all it can do -- see it land and publish it
                        t0 = __builtin_amdgcn_s_memrealtime();
                        if (pend_slot >= 0) { gr_vmcnt<0>(); if (lane == 0) eng_lds_st(ctl + L, gs); pend_slot = -1; }
                    }
                    __builtin_amdgcn_s_sleep(1);
                    if ((n & 63) == 63) { if (sy.aborted()) return; if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { sy.abort(); return; } }
                }
            }
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(ring + slot * G::kSlot));
            const bool stamp = kAblate && trace && blockIdx.x == 0 && L == 0 && gs < 64 && lane == 0;
            if (stamp) trace[256 * 16 * 4 + gs * 8 + 0] = __builtin_amdgcn_s_memtime();
            if (!(kAblate && (ablate & 8)))
#pragma unroll
            for (int i = 0; i < kMine; ++i) {
                const int o = L + kGrLoaders * i;
                if (o < G::kOpsA) gr_dma16(rX, voff[i], so_k, dst + (o >> 1) * 2048 + (o & 1) * 1024);
                else if (o < G::kOpsA + G::kOpsB) { const int q = o - G::kOpsA; gr_dma16(rW, voff[i], so_k, dst + G::kOffB + (q >> 1) * 2048 + (q & 1) * 1024); }
                else if (o < G::kOpsA + G::kOpsB + G::kOpsSx) { const int q = o - G::kOpsA - G::kOpsB; gr_dma4(rXs, voff[i], so_sx, dst + G::kOffSx + q * 256); }
                else { const int q = o - G::kOpsA - G::kOpsB - G::kOpsSx; gr_dma4(rWs, voff[i], so_sw, dst + G::kOffSw + q * 256); }
            }
            so_k += kGroup; so_sx += step_sx; so_sw += step_sw;
            if (stamp) trace[256 * 16 * 4 + gs * 8 + 1] = __builtin_amdgcn_s_memtime();
            if (pend_slot >= 0) {                                      // the stage before this one has landed once only this stage's pieces are outstanding
                gr_vmcnt<kMine>();
                if (lane == 0) eng_lds_st(ctl + L, gs);                // stages 0 .. gs - 1 of this loader have landed
            }
            if (stamp) trace[256 * 16 * 4 + gs * 8 + 2] = __builtin_amdgcn_s_memtime();
            pend_slot = (int)slot; pend_round = round;
        }
    }
    if (pend_slot >= 0) { gr_vmcnt<0>(); if (lane == 0) eng_lds_st(ctl + L, gs); }
    if (kAblate && trace && lane == 0) { unsigned long long* t = trace + ((size_t)blockIdx.x * 16 + L) * 4; t[0] = tr_t0; t[1] = __builtin_amdgcn_s_memtime(); t[2] = tr_wait; t[3] = tr_nwait; }
}

// A consumer wave issues an instruction every ~6.5 cycles whatever its kind (tools/ubench/valu_rate.hip; three consumers share a SIMD), so a
// stage costs what its instruction count costs -- the first version of this loop (one v_cvt, v_mul, v_fma per result, a vector-condition
// poll, every LDS wait exposed) ran 2000-3000 cycles per stage, no faster than the tile kernel.  Hence:
//   * the int32 -> fp32 conversion rides on the MFMA: the accumulator starts at 0x4B400000, the bit pattern of 1.5 * 2^23, so a result's
//     bits ARE the float 12582912 + dot (|dot| <= 64 * 128 * 128 < 2^22 keeps it inside the binade), and float(dot) = that - 12582912
//     exactly -- a subtraction, which unlike the conversion has a packed form.  Per 32 results: 8 v_pk_add_f32 (the conversion), 8 v_pk_mul_f32
//     (sW * sX, rounded once each as in the reference), 8 v_pk_fma_f32 (the chain step) instead of 48 scalar instructions; every half of a
//     packed operation is the IEEE operation, so the bits are those of quant_operators.cpp:274;
//   * the stage loop is software-pipelined by hand: the second fragment's operand read flies under the first fragment's MFMAs and chain; the
//     next stage's token and first-fragment operands are requested (behind a scalar-branch poll of the four fill words) before the last
//     chain of this stage, its scales behind it.
template <int EPI, int WT, int WR, int NB>
__device__ __forceinline__ void gr_consumer(const GemmArgs& a, char* lds, const int w, int* err, unsigned long long* trace, const int ablate) {
    using G = GrTile<WT, WR, NB>;
    constexpr bool TWO = EPI == EPI_SWIGLU;
    static_assert(!TWO || NB == 2, "gate and up are the two B fragments of a wave");
    static_assert(NB == 1 || NB == 2, "one or two B fragments per wave");
    constexpr int TRH = TWO ? G::TR / 2 : G::TR;
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wt = w % WT, wr = w / WT;
    unsigned* const ctl = reinterpret_cast<unsigned*>(lds);
    const GrSync sy{ctl, err};
    const int sn = a.n / kGroup, ntb = (a.B + G::TT - 1) / G::TT, nrb = (a.rows + TRH - 1) / TRH, nblk = ntb * nrb;
    int nmine = 0;                                                     // blocks of this workgroup
    for (int k = 0; k * (int)gridDim.x < nblk; ++k) nmine += gr_block_of(blockIdx.x, gridDim.x, k) < nblk;
    const unsigned total = (unsigned)nmine * (unsigned)sn;             // stages of this workgroup
    if (total == 0) return;
    const char* const ring = lds + kGrCtl;
    const unsigned offA = (unsigned)(wt * 2048 + lane * 16), offB = (unsigned)(G::kOffB + wr * NB * 2048 + lane * 16);
    const unsigned offsx = (unsigned)(G::kOffSx + (wt * 32 + 4 * h) * 4), offsw = (unsigned)(G::kOffSw + (wr * NB * 32 + l31) * 4);
    constexpr int kMagic = 0x4B400000;                                 // bits of 12582912.0f = 1.5 * 2^23
    const v16i Kc = {kMagic, kMagic, kMagic, kMagic, kMagic, kMagic, kMagic, kMagic, kMagic, kMagic, kMagic, kMagic, kMagic, kMagic, kMagic, kMagic};
    const f2 negK = {-12582912.f, -12582912.f};
    unsigned long long tr_t0 = 0, tr_wait = 0, tr_w0 = 0; unsigned tr_nwait = 0;
    if (kAblate && trace) tr_t0 = __builtin_amdgcn_s_memtime();
    if (kAblate && trace && blockIdx.x == 0 && w == 0 && lane == 0) { trace[256 * 16 * 4 + 64 * 8 + 48] = __builtin_amdgcn_s_memtime(); trace[256 * 16 * 4 + 64 * 8 + 49] = __builtin_amdgcn_s_memrealtime(); }

    // wait until all four loaders' pieces of stage gs have landed: pub[L] counts loader L's landed stages; the minimum is cached, so the words are
    // read again only when the consumer has caught up with what it last saw (scalar branch: the words are the same for every lane)
    unsigned avail = 0;
    auto poll = [&](unsigned gs) -> bool {
        if (kAblate && (ablate & 16)) return true;
        if (avail > gs) return true;
        const unsigned fw = (unsigned)(uintptr_t)ctl;
        unsigned long long t0 = 0;
        for (unsigned n = 0;; ++n) {
            v4u f;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(fw) : "memory");
            const unsigned m0 = f.x < f.y ? f.x : f.y, m1 = f.z < f.w ? f.z : f.w, m = m0 < m1 ? m0 : m1;
            avail = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
            if (avail > gs) { if (kAblate && trace && n) { tr_wait += __builtin_amdgcn_s_memtime() - tr_w0; ++tr_nwait; } return true; }
            if (kAblate && trace && n == 0) tr_w0 = __builtin_amdgcn_s_memtime();
            if (n == 0) t0 = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_sleep(1);
            if ((n & 63) == 63) { if (sy.aborted()) return false; if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { sy.abort(); return false; } }
        }
    };
    // one fragment's 32 results: acc = fma(sW * sX, float(dot), acc), two results per instruction
    auto chain = [&](f2 (&ac)[8], const v16i& d, const float swj, const float4 (&sx)[4]) {
        const f2 swv = {swj, swj};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f2 s01 = swv * f2{sx[q].x, sx[q].y}, s23 = swv * f2{sx[q].z, sx[q].w};
            const f2 f01 = f2{__int_as_float(d[4 * q + 0]), __int_as_float(d[4 * q + 1])} + negK, f23 = f2{__int_as_float(d[4 * q + 2]), __int_as_float(d[4 * q + 3])} + negK;
            ac[2 * q] = __builtin_elementwise_fma(s01, f01, ac[2 * q]);            // quant_operators.cpp:274
            ac[2 * q + 1] = __builtin_elementwise_fma(s23, f23, ac[2 * q + 1]);
        }
    };

    f2 acc[NB][8];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f2{0.f, 0.f};
    // stage 0's operands
    if (!poll(0)) return;
    v4i a0, a1, b0v[2]; float4 sx[4]; float sw[NB];
    {
        const char* base = ring;
        a0 = *reinterpret_cast<const v4i*>(base + offA); a1 = *reinterpret_cast<const v4i*>(base + offA + 1024);
        b0v[0] = *reinterpret_cast<const v4i*>(base + offB); b0v[1] = *reinterpret_cast<const v4i*>(base + offB + 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q) sx[q] = *reinterpret_cast<const float4*>(base + offsx + q * 32);
#pragma unroll
        for (int j = 0; j < NB; ++j) sw[j] = *reinterpret_cast<const float*>(base + offsw + j * 128);
    }
    int k = 0, g = 0;
    while (gr_block_of(blockIdx.x, gridDim.x, k) >= nblk) ++k;        // (a workgroup's live blocks: rounds k with a block index inside the problem)
    for (unsigned gs = 0; gs < total; ++gs) {
        const unsigned slot = gs % G::NS;
        const char* base = ring + slot * G::kSlot;
        const bool more = gs + 1 < total;
        const bool stamp = kAblate && trace && blockIdx.x == 0 && w == 0 && gs < 64 && lane == 0;
        if (stamp) trace[256 * 16 * 4 + gs * 8 + 4] = __builtin_amdgcn_s_memtime();
        if (kAblate && trace && blockIdx.x == 0 && lane == 0 && (gs & 15) == 0 && gs < 64) trace[256 * 16 * 4 + 64 * 8 + w * 4 + (gs >> 4)] = __builtin_amdgcn_s_memtime();
        v4i b1v[2];
        if constexpr (NB == 2) { b1v[0] = *reinterpret_cast<const v4i*>(base + offB + 2048); b1v[1] = *reinterpret_cast<const v4i*>(base + offB + 2048 + 1024); }
        v16i d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0v[0], Kc, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0v[1], d0, 0, 0, 0);
        v16i d1 = d0;
        if constexpr (NB == 2) {
            if (!(kAblate && (ablate & 1))) chain(acc[0], d0, sw[0], sx);
            d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1v[0], Kc, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1v[1], d1, 0, 0, 0);
        }
        // every LDS read of this slot has been issued (the operands were read a stage ago, b1v above): the slot may be refilled once they
        // have been served -- LDS operations of a wave are served in order, so the store behind them needs no wait
        asm volatile("" ::: "memory");
        if (lane == 0 && !(kAblate && (ablate & 16))) eng_lds_st(ctl + kGrOffProg / 4 + w, gs + 1);
        asm volatile("" ::: "memory");
        const char* nbase = ring + ((gs + 1) % G::NS) * G::kSlot;
        if (more) {
            if (!poll(gs + 1)) return;
            a0 = *reinterpret_cast<const v4i*>(nbase + offA); a1 = *reinterpret_cast<const v4i*>(nbase + offA + 1024);
            b0v[0] = *reinterpret_cast<const v4i*>(nbase + offB); b0v[1] = *reinterpret_cast<const v4i*>(nbase + offB + 1024);
        }
        if (!(kAblate && (ablate & 1))) chain(acc[NB - 1], d1, sw[NB - 1], sx);
        if (more) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sx[q] = *reinterpret_cast<const float4*>(nbase + offsx + q * 32);
#pragma unroll
            for (int j = 0; j < NB; ++j) sw[j] = *reinterpret_cast<const float*>(nbase + offsw + j * 128);
        }
        if (++g == sn) {                                               // the block is complete
            const int blk = gr_block_of(blockIdx.x, gridDim.x, k);
            const int r0 = (blk / ntb) * TRH, b0 = (blk % ntb) * G::TT;
            float out[NB][16];
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) { out[j][2 * i] = acc[j][i].x; out[j][2 * i + 1] = acc[j][i].y; acc[j][i] = f2{0.f, 0.f}; }
            gemm_epilogue<EPI, NB>(a, out, r0 + wr * 32 * (TWO ? 1 : NB) + l31, b0 + wt * 32, lane);
            g = 0; ++k;
            while (more && gr_block_of(blockIdx.x, gridDim.x, k) >= nblk) ++k;
        }
    }
    if (kAblate && trace && blockIdx.x == 0 && w == 0 && lane == 0) { trace[256 * 16 * 4 + 64 * 8 + 50] = __builtin_amdgcn_s_memtime(); trace[256 * 16 * 4 + 64 * 8 + 51] = __builtin_amdgcn_s_memrealtime(); }
    if (kAblate && trace && lane == 0) { unsigned long long* t = trace + ((size_t)blockIdx.x * 16 + kGrLoaders + w) * 4; t[0] = tr_t0; t[1] = __builtin_amdgcn_s_memtime(); t[2] = tr_wait; t[3] = tr_nwait; }
}

// NB == 1: one fragment per wave, pipelined ACROSS stages -- step i issues the MFMAs of stage i + 1 into one result buffer, requests the
// operands of stage i + 2, runs the chain of stage i from the other buffer, then requests the scales of stage i + 1: every LDS read and every
// MFMA result has a chain's worth of instructions between its issue and its use.  A wave reads from two slots at a time (operands one stage
// ahead of scales).
// Measured on the way (tools/ubench/chainrate.hip, tools/ubench/gemm_ring.hip): a fragment's chain step in the plain form (16 x v_cvt_f32_i32,
// v_mul_f32, v_fma_f32) costs a SIMD 43 cycles with four waves on it, 58 with three; the packed form on magic-biased MFMA results (accumulator
// preset to 0x4B400000, float(dot) = bits - 12582912 as a v_pk_add_f32; 24 instead of 48 instructions) costs 53 / 61 -- packed fp32 is no faster
// per result here, and its constant accumulator tuple costs 16 registers -- so the chain is the plain one.  LDS reads, waits and the progress
// word are asm: their order of issue IS the protocol, the waits are counted by hand (LDS operations return in order), and the steady-state step
// is one straight path (no register copies where branches would meet).
template <int EPI, int WT, int WR>
__device__ __forceinline__ void gr_consumer1(const GemmArgs& a, char* lds, const int w, int* err, unsigned long long* trace, const int ablate) {
    using G = GrTile<WT, WR, 1>;
    static_assert(EPI != EPI_SWIGLU, "gate and up need the two-fragment consumer");
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wt = w % WT, wr = w / WT;
    unsigned* const ctl = reinterpret_cast<unsigned*>(lds);
    const GrSync sy{ctl, err};
    const int sn = a.n / kGroup, ntb = (a.B + G::TT - 1) / G::TT, nrb = (a.rows + G::TR - 1) / G::TR, nblk = ntb * nrb;
    int nmine = 0;
    for (int k = 0; k * (int)gridDim.x < nblk; ++k) nmine += gr_block_of(blockIdx.x, gridDim.x, k) < nblk;
    const unsigned total = (unsigned)nmine * (unsigned)sn;
    if (total == 0) return;
    const unsigned ring = (unsigned)(uintptr_t)(lds + kGrCtl);
    const unsigned offA = ring + (unsigned)(wt * 2048 + lane * 16), offBv = ring + (unsigned)(G::kOffB + wr * 2048 + lane * 16);
    const unsigned offsx = ring + (unsigned)(G::kOffSx + (wt * 32 + 4 * h) * 4), offsw = ring + (unsigned)(G::kOffSw + (wr * 32 + l31) * 4);
    const unsigned progaddr = (unsigned)(uintptr_t)(ctl + kGrOffProg / 4 + w);
    unsigned long long tr_t0 = 0, tr_wait = 0, tr_w0 = 0; unsigned tr_nwait = 0;
    if (kAblate && trace) tr_t0 = __builtin_amdgcn_s_memtime();
    bool ok = true;
    unsigned avail = 0;                       // stages all four loaders have published (cached minimum of pub[0..3])
    auto poll = [&](unsigned stage) {
        if (kAblate && (ablate & 16)) return;
        if (avail > stage) return;
        const unsigned fw = (unsigned)(uintptr_t)ctl;
        unsigned long long t0 = 0;
        for (unsigned n = 0;; ++n) {
            v4u f;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(fw) : "memory");
            const unsigned m0 = f.x < f.y ? f.x : f.y, m1 = f.z < f.w ? f.z : f.w, m = m0 < m1 ? m0 : m1;
            avail = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
            if (avail > stage) { if (kAblate && trace && n) { tr_wait += __builtin_amdgcn_s_memtime() - tr_w0; ++tr_nwait; } return; }
            if (kAblate && trace && n == 0) tr_w0 = __builtin_amdgcn_s_memtime();
            if (n == 0) t0 = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_sleep(1);
            if ((n & 63) == 63) { if (sy.aborted()) { ok = false; return; } if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { sy.abort(); ok = false; return; } }
        }
    };
    v4i a0, a1, b0, b1; v4i sxA[4], sxB[4]; int swA, swB;             // (two sets of scales: a stage's are requested a whole step before its chain)
#define FLM_GR_REQ_AB(so) do { const unsigned va_ = offA + (so), vb_ = offBv + (so); \
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:1024" \
                     : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1) : "v"(va_), "v"(vb_) : "memory"); } while (0)
#define FLM_GR_REQ_S(so, sx, sw) do { const unsigned vx_ = offsx + (so), vw_ = offsw + (so); \
        asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:32\n\tds_read_b128 %2, %5 offset:64\n\tds_read_b128 %3, %5 offset:96\n\tds_read_b32 %4, %6" \
                     : "=&v"(sx[0]), "=&v"(sx[1]), "=&v"(sx[2]), "=&v"(sx[3]), "=&v"(sw) : "v"(vx_), "v"(vw_) : "memory"); } while (0)
    // (the waits name the registers they cover: what consumes them is ordered behind the wait by the data dependency)
#define FLM_GR_WAIT_AB(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) :: "memory")
#define FLM_GR_WAIT_S(n, sx, sw) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(sx[0]), "+v"(sx[1]), "+v"(sx[2]), "+v"(sx[3]), "+v"(sw) :: "memory")
#define FLM_GR_MFMA(d) do { const v16i z_ = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, z_, 0, 0, 0); d = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, d, 0, 0, 0); } while (0)
    float acc[1][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = 0.f;
    auto chain = [&](const v16i& d, const v4i (&sxq)[4], const int swi) {
        const float swj = __int_as_float(swi);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[0][4 * q + 0] = __fmaf_rn(__fmul_rn(swj, __int_as_float(sxq[q].x)), (float)d[4 * q + 0], acc[0][4 * q + 0]);   // quant_operators.cpp:274
            acc[0][4 * q + 1] = __fmaf_rn(__fmul_rn(swj, __int_as_float(sxq[q].y)), (float)d[4 * q + 1], acc[0][4 * q + 1]);
            acc[0][4 * q + 2] = __fmaf_rn(__fmul_rn(swj, __int_as_float(sxq[q].z)), (float)d[4 * q + 2], acc[0][4 * q + 2]);
            acc[0][4 * q + 3] = __fmaf_rn(__fmul_rn(swj, __int_as_float(sxq[q].w)), (float)d[4 * q + 3], acc[0][4 * q + 3]);
        }
    };
    int k = 0, g = 0;
    while (gr_block_of(blockIdx.x, gridDim.x, k) >= nblk) ++k;
    // slot bookkeeping without divisions: stage i lives in slot i mod NS, round i / NS
    unsigned s_ab = 0, r_ab = 1, so_ab = 0;   // slot / needed fill value / byte offset of the stage whose OPERANDS are requested next
    unsigned so_s = 0;                        // byte offset of the slot whose SCALES are requested next
    auto adv_ab = [&]() { so_ab += G::kSlot; if (++s_ab == (unsigned)G::NS) { s_ab = 0; so_ab = 0; ++r_ab; } };
    auto adv_s = [&]() { so_s += G::kSlot; if (so_s == (unsigned)(G::NS * G::kSlot)) so_s = 0; };
    auto block_done = [&](const unsigned i) {
        const int blk = gr_block_of(blockIdx.x, gridDim.x, k);
        const int r0 = (blk / ntb) * G::TR, b0r = (blk % ntb) * G::TT;
        gemm_epilogue<EPI, 1>(a, acc, r0 + wr * 32 + l31, b0r + wt * 32, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][q] = 0.f;
        g = 0; ++k;
        while (i + 1 < total && gr_block_of(blockIdx.x, gridDim.x, k) >= nblk) ++k;
    };
    // prologue: operands of stage 0, its MFMAs, operands of stage 1 (a repeat of stage 0's if there is none: the request pattern never changes), scales of stage 0
    poll(0); if (!ok) return;
    FLM_GR_REQ_AB(0u); adv_ab();
    FLM_GR_WAIT_AB(0);
    v16i dA, dB2;
    FLM_GR_MFMA(dA);
    dB2 = dA;
    asm volatile("" :: "v"(dA) : "memory");
    if (total > 1) { poll(1); if (!ok) return; }
    FLM_GR_REQ_AB(total > 1 ? so_ab : 0u); if (total > 1) adv_ab();
    FLM_GR_REQ_S(0u, sxA, swA);
    // invariant at the top of step i: outstanding LDS reads = the 4 operand reads of stage i + 1, then the 5 scale reads of stage i.
    // step i: MFMAs of stage i + 1 -> dNext; operands of stage i + 2 (behind their poll) and scales of stage i + 1 requested; chain of stage i
    // from dCur and the scales requested a step ago; the progress word.  Past the end the requests repeat the last slots read (never consumed)
    // and the MFMAs run on them (never used).
#define FLM_GR_STEP(i, dCur, dNext, sxCur, swCur, sxNext, swNext) do { \
        if (kAblate && trace && blockIdx.x == 0 && lane == 0 && ((i) & 15) == 0 && (i) < 64) trace[256 * 16 * 4 + 64 * 8 + w * 4 + ((i) >> 4)] = __builtin_amdgcn_s_memtime(); \
        FLM_GR_WAIT_AB(5); \
        FLM_GR_MFMA(dNext); \
        asm volatile("" :: "v"(dNext) : "memory"); \
        if ((i) + 2 < total) { poll((i) + 2); if (!ok) return; } \
        FLM_GR_REQ_AB(so_ab); if ((i) + 2 < total) adv_ab(); \
        if ((i) + 1 < total) adv_s(); \
        FLM_GR_REQ_S(so_s, sxNext, swNext); \
        FLM_GR_WAIT_S(9, sxCur, swCur); \
        if (!(kAblate && (ablate & 1))) chain(dCur, sxCur, swCur); \
        if (!(kAblate && (ablate & 16))) asm volatile("ds_write_b32 %0, %1" :: "v"(progaddr), "v"((i) + 1) : "memory"); \
        if (++g == sn) block_done(i); \
    } while (0)
    for (unsigned i = 0; i < total; i += 2) {
        FLM_GR_STEP(i, dA, dB2, sxA, swA, sxB, swB);
        if (i + 1 < total) FLM_GR_STEP(i + 1, dB2, dA, sxB, swB, sxA, swA);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (kAblate && trace && lane == 0) { unsigned long long* t = trace + ((size_t)blockIdx.x * 16 + kGrLoaders + w) * 4; t[0] = tr_t0; t[1] = __builtin_amdgcn_s_memtime(); t[2] = tr_wait; t[3] = tr_nwait; }
#undef FLM_GR_STEP
#undef FLM_GR_MFMA
#undef FLM_GR_WAIT_AB
#undef FLM_GR_WAIT_S
#undef FLM_GR_REQ_AB
#undef FLM_GR_REQ_S
}

template <int EPI, int WT, int WR, int NB>
__global__ void __launch_bounds__(kGrBlock) k_gemm_q8_ring(const GemmArgs a, int* err, unsigned long long* trace, int ablate) {   // trace, ablate: FLM_ABLATE builds (tools/ubench/gemm_ring.hip)
    using G = GrTile<WT, WR, NB>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x < kGrCtl / 4) reinterpret_cast<unsigned*>(lds)[threadIdx.x] = 0u;
    __syncthreads();                                                   // the only barrier: the control words start at zero
    if (wave < kGrLoaders) {
        switch (wave) {
        case 0: gr_loader<EPI, WT, WR, NB, 0>(a, lds, err, trace, ablate); break;
        case 1: gr_loader<EPI, WT, WR, NB, 1>(a, lds, err, trace, ablate); break;
        case 2: gr_loader<EPI, WT, WR, NB, 2>(a, lds, err, trace, ablate); break;
        default: gr_loader<EPI, WT, WR, NB, 3>(a, lds, err, trace, ablate); break;
        }
    } else if (wave - kGrLoaders < G::NC) {
        if constexpr (NB == 1) gr_consumer1<EPI, WT, WR>(a, lds, wave - kGrLoaders, err, trace, ablate);
        else gr_consumer<EPI, WT, WR, NB>(a, lds, wave - kGrLoaders, err, trace, ablate);
    }
}

}  // namespace flm
