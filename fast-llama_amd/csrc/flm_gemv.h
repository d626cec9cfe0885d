// flm_gemv.h -- the group-quantized GEMV (k_gemv): argument block, LDS layout, the rmsnorm chain, activation prologues, GemvCtx.
// Part of flm_kernels.h (hand-written gfx950 / CDNA4 kernels of the fast-llama per-token hot path); include that header.
#pragma once
#include "flm_math.h"
// (bit-exactness hygiene: see flm_math.h -- no implicit FMA contraction in any of these headers)
#pragma clang fp contract(off)

namespace flm {

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
    int ctr_off;                                // LDS byte offset of the two step counters; 0: the layout's own
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
    // tensor parallel, peer-to-peer: every result is also stored into the same place of every peer rank's buffer (mapped over
    // xGMI; system-scope stores), so that after the launch plus one flag round every rank holds the whole vector.
    // out_peer[i] = peer i's `out` pointer (already offset like `out`); for EPI_ROPE_KV nothing is exchanged (q/k/v stay local).
    float* out_peer[7]; int n_peer;
    // the same vector as data-tagged granules (granule_t below; k_layers' granule hand-offs: GemvCtx::gron): local / every peer's, offset like `out`; the tag is the launch's (run time)
    unsigned long long* gout; unsigned long long* gout_peer[7];
    unsigned long long* gk; unsigned long long* gv;   // EPI_ROPE_KV: this token's K / V cache row [kv_dim] as granules beside the cache rows (gout: q)
    // ... and the flag round of that exchange folded into the CONSUMING launch (XchgFold below; world == 0: not used)
    struct XchgFold {
        unsigned* local_flags; unsigned* peer_flags[8];         // flag lines [slot][rank], 64 bytes each, in every rank's exchange buffer
        const unsigned* base; unsigned add;                     // epoch of the exchange this launch consumes = *base (the token's, advanced by k_embed) + add
        int rank, world, slot; int* err;
    } xf;
    // debugging taps used by the op-level exports (may be null)
    void* dbg_xq; float* dbg_xs; float* dbg_xn;
    unsigned long long* trace;                  // FLM_ABLATE builds: per-workgroup timeline [grid][8] (s_memtime), else unused
    int ablate;                                 // FLM_ABLATE builds only (every test of it sits behind kAblate; product builds ignore it): 1 no group chain, 2 no rmsnorm chain, 4 no weight loads, 8 no dots, 16 return immediately, 32 return after prologue
};

// result store: local (write-through, agent scope) + every peer (system scope)
__device__ __forceinline__ void st_result(const GemvArgs& a, unsigned row, float v) {
    st_agent(a.out + row, v);
    for (int i = 0; i < a.n_peer; ++i) __hip_atomic_store(a.out_peer[i] + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#ifndef FLM_ABLATE
#define FLM_ABLATE 0          // build with -DFLM_ABLATE=1 to compile the perf-exploration switches of GemvArgs::ablate into the hot loop
#endif
constexpr bool kAblate = FLM_ABLATE != 0;
constexpr int kStepBlk = 4;            // H: 1 KiB wave loads per step; two steps (register sets) in flight: 8 KiB/wave, 128 KiB/CU

// staging geometry of the speculative chain: B = elements per lane = the power of two >= max(4, ceil(n/4 / 64))
__host__ __device__ inline int chain_bshift(int n) { int s = 2; while ((64 << s) < n / 4) ++s; return s; }
__host__ __device__ inline int chain_strip_floats(int n) { return 64 * ((1 << chain_bshift(n)) + 4); }
// LDS layout: [xq : n*esz] [xs : n/64 floats, padded to 16 B] [red : 16 floats] [scratch]
// scratch = max( rmsnorm staging: 4 chain strips of [64 lanes][B + 4] floats (sq_chain_spec) ,
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
    const int chain = 4 * chain_strip_floats(n) * 4 + 64;                       // the 4 strided lanes of square_sum, each a lane-major strip
    if (norm && chain > scratch) scratch = chain;
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

// The same chain -- l <- fma(x_k, x_k, l), k ascending, bit for bit -- evaluated by a WHOLE WAVE in a handful of rounds
// instead of n dependent steps (tools/chain_emul.c is the build-host emulation of this function, fuzzed against the
// sequential chain; tests/test_gpu_ops.py::test_square_sum_speculative_is_bit_exact checks the kernel).
// The terms are non-negative, so l only grows, and while l stays inside one binade [A, 2A), A = 2^E, every step rounds
// l + x^2 to a multiple of u = ulp(A): the increment fl(l + x^2) - l does not depend on l (exact ties aside), it is
// t = fma(x, x, A) - A.  Lane L owns B consecutive elements (LDS strip, lane-major, row stride B + 4 floats):
//   1. approximate prefix of sum x^2 (fp32 wave scan) -> the binade E_L the lane expects to start in;
//   2. T_L = sum of its increments against 2^E_L (multiples of u: exact in any order), tie flag (x^2 exactly half way);
//   3. exact exclusive prefix S_L of the T's in fp64 (every T is a multiple of 2^(Emin-23), every partial sum is below
//      2^(Emax+2): exact while Emax - Emin <= 26, else the plain chain runs);
//   4. rounds: given an exact (base lane, base value), lane t would start at base + (S_t - S_base); it is CONSISTENT if
//      that start lies in its expected binade, start + T_t stays below 2^(E_t+1) and it saw no tie.  The first
//      inconsistent lane f has an exact start (all lanes before it are consistent): its B real steps give the next base.
// l crosses a binade ~log2(n) times, mostly early: the first H lanes' elements simply run in order, then ~4 rounds.
__device__ __forceinline__ double wave_scan_incl_f64(double v) {
#define FLM_SCAN64(ctrl, rmask, bc) { const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xF, bc), hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xF, bc); v = __dadd_rn(v, __hiloint2double(hi_, lo_)); }
    FLM_SCAN64(0x111 /* row_shr:1 */, 0xF, true) FLM_SCAN64(0x112 /* row_shr:2 */, 0xF, true)
    FLM_SCAN64(0x114 /* row_shr:4 */, 0xF, true) FLM_SCAN64(0x118 /* row_shr:8 */, 0xF, true)
    FLM_SCAN64(0x142 /* row_bcast:15 */, 0xA, false) FLM_SCAN64(0x143 /* row_bcast:31 */, 0xC, false)
#undef FLM_SCAN64
    return v;
}
__device__ __forceinline__ float wave_scan_incl(float v) {
#define FLM_SCAN_STEP(ctrl, rmask, bc) v = __fadd_rn(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, bc)));
    FLM_SCAN_STEP(0x111 /* row_shr:1 */, 0xF, true) FLM_SCAN_STEP(0x112 /* row_shr:2 */, 0xF, true)
    FLM_SCAN_STEP(0x114 /* row_shr:4 */, 0xF, true) FLM_SCAN_STEP(0x118 /* row_shr:8 */, 0xF, true)
    FLM_SCAN_STEP(0x142 /* row_bcast:15 */, 0xA, false) FLM_SCAN_STEP(0x143 /* row_bcast:31 */, 0xC, false)
#undef FLM_SCAN_STEP
    return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
constexpr int kChainHead = 4;          // lanes whose elements run in order before the speculative part
// p: one chain's strip [64 lanes][B + 4] floats, zero-padded past the data.  Returns the chain value in every lane.
// BVR > 0: B = 4 * BVR is a compile-time constant and the lane's elements stay in registers (B <= 16: every model width up to
// 4096); BVR == 0: any B, elements re-read from LDS in every pass.
#ifdef FLM_TRACE_PRO_RT
#define FLM_CHAIN_STAMP(k) if (tr && lane == 0) tr[k] = __builtin_amdgcn_s_memtime();
#else
#define FLM_CHAIN_STAMP(k)
#endif
// OP 0: l <- fma(x, x, l) (the rmsnorm sum of squares); OP 1: l <- l + x, x >= 0 (any chain of non-negative terms: the argument above holds verbatim; round 5 evaluated the
// long-context softmax sum this way -- exact, and slower than the lone lane there: tools/experiments/r5_handoffs/spec_softmax_sum.*).  PAD: floats between two lanes' runs of B elements.
template <int OP> __device__ __forceinline__ float chain_step(float l, float x) { if constexpr (OP == 0) return __fmaf_rn(x, x, l); else return __fadd_rn(l, x); }
template <int BVR, int OP = 0, int PAD = 4>
__device__ __forceinline__ float chain_spec_t(const float* p, const int bshift, int* rounds_out, unsigned long long* tr = nullptr, const int Brt = 0) {
    const int lane = threadIdx.x & 63, B = BVR ? 4 * BVR : (Brt ? Brt : (1 << bshift)), BV = B >> 2, LS = B + PAD;      // (Brt: any multiple of 4, LDS-fed form)
    const float4* pl = reinterpret_cast<const float4*>(p + lane * LS);
    float4 xr[BVR ? BVR : 1];
    if constexpr (BVR > 0) {
#pragma unroll
        for (int q = 0; q < BVR; ++q) xr[q] = pl[q];
    }
    auto elem = [&](int q) -> float4 { if constexpr (BVR > 0) return xr[q]; else return pl[q]; };
    FLM_CHAIN_STAMP(0)
#define FLM_SQ4(l, v) l = chain_step<OP>(l, v.x); l = chain_step<OP>(l, v.y); l = chain_step<OP>(l, v.z); l = chain_step<OP>(l, v.w);
    // (A lone wave issues one instruction every ~8 shader clocks whatever its kind -- measured in situ: ~900 instructions took 3.3 us -- so what counts below is
    //  the NUMBER of instructions on this wave's path, not their latencies.)
    // 1. approximate per-lane sums; all-zero lanes (every element +-0: pass-through whatever l is)
    float s = 0.f; unsigned orb = 0u;
#pragma unroll
    for (int q = 0; q < BV; ++q) { const float4 v = elem(q); FLM_SQ4(s, v) orb |= __float_as_uint(v.x) | __float_as_uint(v.y); orb |= __float_as_uint(v.z) | __float_as_uint(v.w); }
    const bool allzero = (orb & 0x7fffffffu) == 0u;
    const float P = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(wave_scan_incl(s)), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
    FLM_CHAIN_STAMP(1)
    // the head: lanes [0, kChainHead) in order.  Every lane advances ITS elements from whatever it holds, then takes its left neighbour's result
    // (wave_shr:1): after round k lane k holds the chain through lanes 0..k -- 17 instructions per head lane, no broadcast of the elements.
    float hv;
    {
        float l = 0.f;
#pragma unroll
        for (int L = 0; L < kChainHead; ++L) {
#pragma unroll
            for (int q = 0; q < BV; ++q) { const float4 v = elem(q); FLM_SQ4(l, v) }
            if (L + 1 < kChainHead) l = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(l), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
        }
        hv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l), kChainHead - 1));
    }
    FLM_CHAIN_STAMP(2)
    // 2. the lane's increments against the binade it expects to start in, T = sum of RN_u(x^2), as the chain itself run from A = 2^E: while a chain stays inside
    //    [A, 2A) every step adds RN_u(x^2) whatever multiple of u it started from (exact ties aside), so chain(A) - A is that sum; if it leaves the binade,
    //    T >= A and the lane fails the "stays below 2A" test of the rounds anyway.  Ties: the same chain from A + u has the other parity -- without a tie
    //    the two end exactly u apart, a tie moves them to 0 or 2u apart (and later ties keep them there).  34 instructions instead of 96.
    const unsigned eb = __float_as_uint(P) & 0x7f800000u;
    bool valid = lane >= kChainHead && eb >= (27u << 23) && eb <= (250u << 23);      // P finite and normal, room for u and 2A
    const unsigned ebv = valid ? eb : 0x3f800000u;
    const float A = __uint_as_float(ebv), u1 = __uint_as_float(ebv - (23u << 23)), top = __fadd_rn(A, A);
    float ca = A, cb = __fadd_rn(A, u1);
#pragma unroll
    for (int q = 0; q < BV; ++q) { const float4 v = elem(q); FLM_SQ4(ca, v) FLM_SQ4(cb, v) }
    float T = __fsub_rn(ca, A);
    const bool tie = __fsub_rn(cb, ca) != u1;
    if (!(T < INFINITY)) valid = false;                         // (inf / NaN never enters the prefix)
    if (!valid) T = 0.f;
    FLM_CHAIN_STAMP(3)
    // 3. fp64 exactness of the prefix
    const unsigned long long live = __ballot(valid && !allzero);
    bool plain = false;
    if (live) {
        const unsigned e0 = (unsigned)__builtin_amdgcn_readlane((int)eb, __ffsll((long long)live) - 1), e1 = (unsigned)__builtin_amdgcn_readlane((int)eb, 63 - __clzll((long long)live));
        plain = e1 < e0 || ((e1 - e0) >> 23) > 26;
    }
    if (plain) {                                                // extreme dynamic range: the plain chain over the strip
        float l = 0.f;
        for (int L = 0; L < 64; ++L) { const float4* r = reinterpret_cast<const float4*>(p + L * LS); for (int q = 0; q < BV; ++q) { const float4 v = r[q]; FLM_SQ4(l, v) } }
        if (rounds_out) *rounds_out = -1;
        return l;
    }
    const double Td = (double)T, Si = wave_scan_incl_f64(Td), S = __dsub_rn(Si, Td);
    FLM_CHAIN_STAMP(4)
    // 4. rounds.  The lane tests as wave masks (a v_cmp writes its mask straight into scalar registers; the rest is scalar logic): ~40 instructions per round.
    //    (No test that the presumed start is exactly an fp32 value: the first inconsistent lane's start is exact by induction -- all lanes before it are consistent --,
    //     and the lanes behind it are not looked at.)
    int rounds = 0;
    const unsigned long long zero_m = __ballot(allzero), good_m = __ballot(valid && !tie);
    unsigned long long done_m = (1ull << kChainHead) - 1ull;     // lanes in front of the base
    double bv = (double)hv, Sb = readlane_f64(S, kChainHead);
    float res;
    for (;;) {
        ++rounds;
        if (rounds <= 9) { FLM_CHAIN_STAMP(4 + rounds) }
        const float st = (float)__dadd_rn(bv, __dsub_rn(S, Sb));
        const unsigned long long e_m = __ballot((__float_as_uint(st) & 0x7f800000u) == eb), t_m = __ballot(__fadd_rn(st, T) < top);
        const unsigned long long bad = ~(zero_m | (good_m & e_m & t_m) | done_m);
        if (bad == 0) { res = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint((float)__dadd_rn(bv, __dsub_rn(Si, Sb))), 63)); break; }
        const int f = __ffsll((long long)bad) - 1;
        float l = st;                                           // every lane runs its B steps from its presumed start; lane f's start is exact
#pragma unroll
        for (int q = 0; q < BV; ++q) { const float4 v = elem(q); FLM_SQ4(l, v) }
        res = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(l), f));
        if (f == 63) break;
        done_m = (2ull << f) - 1ull; bv = (double)res; Sb = readlane_f64(S, f + 1);
    }
#undef FLM_SQ4
    FLM_CHAIN_STAMP(14)
    if (tr && lane == 0) tr[15] = (unsigned long long)rounds;
    if (rounds_out) *rounds_out = rounds;
    return res;
}
// (BVR > 0 keeps the lane's elements in registers; inside k_gemv that costs 16+ VGPRs next to the two prefetched weight sets and the
//  kernel SPILLS -- scratch accesses then queue behind the weight loads in the memory pipeline and the chain took 10 us instead of 2.5:
//  measured.  The prologue therefore runs the LDS-fed form; the register form is for callers with registers to spare.)
template <int BVR>
__device__ __forceinline__ float sq_chain_spec_t(const float* p, const int bshift, int* rounds_out, unsigned long long* tr = nullptr) { return chain_spec_t<BVR, 0, 4>(p, bshift, rounds_out, tr); }
__device__ __forceinline__ float sq_chain_spec(const float* p, const int bshift, int* rounds_out = nullptr, unsigned long long* tr = nullptr) {
    return sq_chain_spec_t<0>(p, bshift, rounds_out, tr);
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
// xsrc: the activation comes from there instead of a.x (k_layers' first layer of a token: the embedding row, transformer.cpp:115-122)
template <int QT, int PRO, int XR, bool COH = false>
__device__ __forceinline__ void gemv_preload(const GemvArgs& a, float4 (&xv)[XR > 0 ? XR : 1], float4 (&wv)[XR > 0 ? XR : 1], const float* xsrc = nullptr) {
    if constexpr (PRO == PRO_QUANT || PRO == PRO_RMSNORM_QUANT) {
        // branch-free: raw buffer loads, elements past n read as zero
        typedef float v4f __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xsrc ? xsrc : a.x), 0, a.n * 4, 0x00020000);
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

// ------------------------------------------------------------------------------------------
// Data-tagged granules (round 6; cdna_hip_programming.md G16 form R2): the residual stream crosses workgroups INSIDE k_layers' one-launch token as 8-byte {value, tag}
// granules -- ONE aligned write-through store per element, the tag = the flag value the hand-off's line would have carried (epoch base + layer + 1: never repeats).  The
// data is its own flag: the producer neither drains its stores nor raises a line, the consumer neither polls lines nor reads the vector behind them -- every thread re-reads
// ITS four granules until their tags match (only the lanes whose granules are missing read again).  One round trip instead of drain + flag + look + read
// (tools/ubench/allgather.hip).  A granule is written by one store and read by one 16-byte load of two whole granules: no ordering between stores is relied on.
// ------------------------------------------------------------------------------------------
typedef unsigned v4u_g __attribute__((ext_vector_type(4)));
typedef unsigned long long granule_t;
__device__ __forceinline__ void st_granule(granule_t* g, unsigned tag, float v) {
    __hip_atomic_store(g, ((granule_t)tag << 32) | (granule_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_granule_value(const granule_t* g) { return __uint_as_float((unsigned)__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
// result store as granules: local (agent scope) + every peer (system scope: one 8-byte store each, over xGMI as one write)
__device__ __forceinline__ void st_result_granule(const GemvArgs& a, unsigned row, unsigned tag, float v) {
    st_granule(a.gout + row, tag, v);
    const granule_t g = ((granule_t)tag << 32) | (granule_t)__float_as_uint(v);
    for (int i = 0; i < a.n_peer; ++i) __hip_atomic_store(a.gout_peer[i] + row, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One quantizer round of a 16-lane row: the lane owns elements e .. e+3, 16 consecutive lanes own one 64-group (quant::quantize, quant_operators.cpp:26-47:
// scale = max|x| / F, q = (T)(x / scale)); the packed values go to xq[e ..], the group's scale to xs[e / 64].  Every lane of the row must call it (the group
// maximum is a DPP butterfly); `act` = the lane's elements exist.
#ifndef FLM_QUANT_SHARED_RCP
#define FLM_QUANT_SHARED_RCP 1          // the four divisions of a quantizer round share the refined reciprocal of the group's scale (flm_math.h: quant_elems4); 0: four IEEE divisions
#endif
template <int QT>
__device__ __forceinline__ void quant_round4(char* xq, float* xs, int e, bool act, const float4& v) {
    using T = QTraits<QT>;
    // group max over the 16 lanes that share this 64-element group (order-free, exact)
    const float mx = row16_max(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    const float sc = __fdiv_rn(mx, T::kF);           // scale = max|x| / F
    int qq[4];
#if FLM_QUANT_SHARED_RCP
    quant_elems4(v, sc, qq);                          // (every lane: the range check is a ballot)
#else
    qq[0] = quant_elem(v.x, sc); qq[1] = quant_elem(v.y, sc); qq[2] = quant_elem(v.z, sc); qq[3] = quant_elem(v.w, sc);
#endif
    if (act) {
        const int q0 = qq[0], q1 = qq[1], q2 = qq[2], q3 = qq[3];
        if constexpr (QT == QT_INT8) {
            const uint32_t pk = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
            *reinterpret_cast<uint32_t*>(xq + e) = pk;
        } else {
            uint2 pk;
            pk.x = (uint32_t)(q0 & 0xffff) | ((uint32_t)(q1 & 0xffff) << 16);
            pk.y = (uint32_t)(q2 & 0xffff) | ((uint32_t)(q3 & 0xffff) << 16);
            *reinterpret_cast<uint2*>(xq + (size_t)e * 2) = pk;
        }
        if ((threadIdx.x & 15) == 0) xs[e / kGroup] = sc;
    }
}

#ifdef FLM_TRACE_PRO_RT      // tools/trace_back.py: the prologue's stamps on the 100 MHz clock (one clock for all XCDs), wave 0 / wave 15
#define FLM_PRO_STAMP(k) if (kAblate && a.trace && (threadIdx.x == 0 || threadIdx.x == 960)) a.trace[blockIdx.x * 16 + (threadIdx.x ? 8 : 0) + (k)] = __builtin_amdgcn_s_memrealtime();
#elif defined(FLM_TRACE_PRO)
#define FLM_PRO_STAMP(k) if (kAblate && a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define FLM_PRO_STAMP(k)
#endif
// COH: the activation was written by other workgroups of the SAME kernel (k_attn_o) or by peer GPUs (tensor parallel) -> coherent loads
// LATE (rmsnorm prologue): the four chain waves request their weights BEHIND the prologue (at the start of run()) instead of in front of their chain.  The barrier behind the chain waits for every
// wave's issue, and a wave's issue waits for room in the CU's memory pipeline (~30 KB/us): with all 16 waves requesting two register sets the last request is
// accepted ~4.5 us after the first, long after the chain is done.  Only worth it where the run has something to start on while those sets are in flight (k_attn_ffn's stash).
template <int QT, int PRO, int XR, bool COH = false, bool LATE = false, class AfterStage>
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
        // copy the pre-quantized activation (op-level matmul; k_attn_o's heads hand it over quantized)
        const int nb16 = n * T::kEsz / 16;
        if constexpr (COH) {
            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.xq), 0, n * T::kEsz, 0x00020000);
            for (int c = tid; c < nb16; c += kGemvBlock)
                reinterpret_cast<v4i*>(xq)[c] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rq, c * 16, 0, kAuxCoherent));
            for (int g = tid; g < n / kGroup; g += kGemvBlock) xs[g] = ld_agent(a.xs + g);
        } else {
            for (int c = tid; c < nb16; c += kGemvBlock)
                reinterpret_cast<int4*>(xq)[c] = reinterpret_cast<const int4*>(a.xq)[c];
            for (int g = tid; g < n / kGroup; g += kGemvBlock) xs[g] = a.xs[g];
        }
        __syncthreads();
        return;
    } else {
        const int rounds = (n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float r = 1.0f;
#ifdef FLM_TRACE_PRO_RT
        FLM_PRO_STAMP(0)
#endif
        auto ldx = [&](int e) -> float4 {                   // activation elements e..e+3 (rounds past the preloaded registers)
            if constexpr (COH) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, n * 4, 0x00020000);
                const v4f t = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, e * 4, 0, kAuxCoherent));
                return make_float4(t.x, t.y, t.z, t.w);
            } else return *reinterpret_cast<const float4*>(a.x + e);
        };
        if constexpr (PRO == PRO_QUANT) {
            // no staging here: the hook (the weight prefetch) runs as soon as this thread's activation registers have landed
            if constexpr (XR > 0) { asm volatile("" :: "v"(xv[XR - 1].w)); }
            FLM_PRO_STAMP(3)
            if constexpr (!LATE) after_stage(0);              // (LATE: nothing is requested in front of the quantizer; run() requests what was not)
            FLM_PRO_STAMP(4)
        }
        if constexpr (PRO == PRO_RMSNORM_QUANT) {
            // simd::square_sum -> square_sum_avx128 (x86_simd.cpp:942-960; the AVX2 branch is dead, :1093):
            // lane c of 4 accumulates x[c], x[c+4], x[c+8]... by FMA, then res = ((0+l0)+l1)+l2)+l3.
            // Stage x transposed ([4][n/4]) so that 4 threads can each walk one strided lane sequentially.
            // Stage x as 4 strips (one per strided lane c = element index mod 4), each lane-major for sq_chain_spec: chain
            // element k lives in row k >> bs at column k & (B - 1), rows B + 4 floats apart (conflict-free 16-byte reads);
            // slots past the data are zero (loads past n return zero).
            const int bs = chain_bshift(n), B = 1 << bs, LS = B + 4, CS = 64 * LS, slots = 64 << bs;
            auto stage = [&](int i, const float4& v) {
                const int k = tid + i * kGemvBlock;
                if (k < slots) { const int o = (k >> bs) * LS + (k & (B - 1)); scratch[o] = v.x; scratch[CS + o] = v.y; scratch[2 * CS + o] = v.z; scratch[3 * CS + o] = v.w; }
            };
            const int srounds = (slots + kGemvBlock - 1) / kGemvBlock;
#pragma unroll
            for (int i = 0; i < XR; ++i) { if (i < srounds) stage(i, xv[i]); }
            for (int i = XR; i < srounds; ++i) {
                const int e = tid * 4 + i * kGemvBlock * 4;
                stage(i, e < n ? ldx(e) : z4);
            }
#ifdef FLM_TRACE_PRO_RT
            FLM_PRO_STAMP(1)
#endif
            __syncthreads();
            FLM_PRO_STAMP(3)
            // the hook issues the weight prefetch.  The chain waves (0..3, one strided lane each) go first (the others give
            // them ~128 cycles): their loads enter an empty memory pipeline at once and they are free for the chains;
            // queued behind the other waves' loads they would stall for ~1 us before (or after) the chain.
            if constexpr (!LATE) {
                if (tid >= 4 * kWave) __builtin_amdgcn_s_sleep(2);
                after_stage(0);
#ifdef FLM_TRACE_PRO2
                FLM_PRO_STAMP(1)
#endif
                if (tid < 4 * kWave && !(kAblate && (a.ablate & 2))) {
                    const float l = sq_chain_spec(scratch + (tid >> 6) * CS, bs);
                    if ((tid & 63) == 0) red[8 + (tid >> 6)] = l;
                }
            } else {
                // A wave-uniform branch (scalar condition): the chain's code never runs in a wave whose register sets have been requested, so the
                // allocator may give it the sets' registers -- the lane's elements stay in registers (BVR) and the head's reads run far ahead.
                // (the chain waves' own sets are requested at the start of run() -- GemvCtx::primedA/B --, behind the barriers that wait for the chain)
                const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
                if (wv >= 4) after_stage(0);
                else if (!(kAblate && (a.ablate & 2))) {
#ifdef FLM_TRACE_PRO_RT
                    unsigned long long* tr = (kAblate && a.trace && tid < 64) ? a.trace + 2 * 4096 + blockIdx.x * 16 : nullptr;
#else
                    unsigned long long* tr = nullptr;
#endif
                    const float* cp = scratch + wv * CS;
                    float l;
                    if (bs == 4) l = sq_chain_spec_t<4>(cp, bs, nullptr, tr);
                    else if (bs == 3) l = sq_chain_spec_t<2>(cp, bs, nullptr, tr);
                    else if (bs == 2) l = sq_chain_spec_t<1>(cp, bs, nullptr, tr);
                    else l = sq_chain_spec_t<0>(cp, bs, nullptr, tr);
                    if ((tid & 63) == 0) red[8 + wv] = l;
                }
            }
#ifdef FLM_TRACE_PRO_RT
            FLM_PRO_STAMP(2)
#endif
            FLM_PRO_STAMP(4)
            __syncthreads();
            const float ss = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, red[8]), red[9]), red[10]), red[11]);
            r = rms_scale(ss, n);                               // (scratch is not touched again before the barrier that ends the prologue)
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
            quant_round4<QT>(xq, xs, e, act, v);
        };
#pragma unroll
        for (int i = 0; i < XR; ++i) { if (i < rounds) round(i, xv[i], wv[i]); }
        for (int i = XR; i < rounds; ++i) {
            const int e = tid * 4 + i * kGemvBlock * 4;
            float4 v = z4, w = z4;
            if (e < n) {
                v = ldx(e);
                if constexpr (PRO == PRO_RMSNORM_QUANT) w = *reinterpret_cast<const float4*>(a.norm_w + e);
            }
            round(i, v, w);
        }
        if constexpr (PRO == PRO_QUANT) {
            FLM_PRO_STAMP(5)
        }
#ifdef FLM_TRACE_PRO_RT
        FLM_PRO_STAMP(6)
#endif
        __syncthreads();
#ifdef FLM_TRACE_PRO_RT
        FLM_PRO_STAMP(7)
#endif
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
// one fully coalesced buffer_load_dwordx4 per wave and every lane is busy for any K).  H consecutive row
// blocks of one column block form a STEP; the workgroup's steps are numbered and drawn by the waves from a
// counter in LDS (see GemvCtx::Set).  SWIGLU runs [W1 ; W3] as ONE matrix with twice the column blocks: both
// dot products of a row land in the same strip entry and the same chain lane.  Per block:
//   1. int32 dot per 16-byte chunk (v_dot4 / v_dot2), exact;
//   2. DPP sum over the 4 (int8) / 8 (int16) lanes of a quant group -> the group's int32 dot, exact;
//   3. group leaders park float(dot), the step's scale-role lanes park sW*sX, in the row's LDS strip;
// and per pass, after ONE workgroup barrier, one wave walks the strips, lane r = row r:
//        acc = fma(s[g], d[g], acc), g ascending -- the reference's summation order, bit-identical --
// amortising the sequential fp32 chain over Rm rows, and runs the epilogue.  Strips are double
// buffered, so the other waves are already in the next pass.  Two register sets of one step each
// keep 8 KiB per wave (128 KiB per CU) of weight loads in flight at all times; the first two steps of
// every wave are requested as soon as the activation has arrived, so HBM latency and the sequential
// rmsnorm chain overlap the stream.
//
// GemvCtx is the per-wave state of one GEMV: geometry, the step decoder and the two register sets.
// k_gemv and k_attn_o drive it:
//     init -> issue (weight loads of the first two steps) -> [activation prologue] -> run
// ------------------------------------------------------------------------------------------
typedef unsigned int u32;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int QT, int EPI, bool STASH = false>
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
    // LDS stash (STASH; round 4): steps [2 x 16, 2 x 16 + st_n) of the workgroup were fetched with LDS-DMA (buffer_load ... lds: no registers, no VALU)
    // into slots of kSlotBytes at LDS byte st_base while the CU's memory pipeline had nothing else to do (another phase's hand-off, the attention).
    // They are numbered like every other step and handed out by the same counter; load_step takes such a step from its slot instead of from memory.
    // The caller guarantees that every wave has waited for its own DMA (s_waitcnt vmcnt(0)) before the workgroup barrier in front of run().
    u32 st_base, st_n;
    static constexpr u32 kSlotBytes = H * 1024 + 256;                          // H weight blocks + the step's scale dwords
    __amdgpu_buffer_rsrc_t rW, rS;
    Set setA, setB;
    bool stored;                                                               // this wave wrote results to global memory
    const float* resid_src;                                                    // EPI_RESIDUAL: the old value of out[row] is read from resid_src[row] instead (null: out itself)
    // k_layers' granule hand-offs (granule_t above): gron = this GEMV's results leave as granules (GemvArgs::gout / gout_peer) with tag gtag instead of plain stores;
    // EPI_RESIDUAL: the old value comes from gsrc[row] (null: resid_src / out)
    const granule_t* gsrc; bool gron; unsigned gtag;
    // run_ao (Wo inside the whole-layer launches, round 6): only the waves [0, nw) hold steps (wave, wave + nw) and look for their producers; the waves [nw, 16) issue the workgroup's
    // LDS-DMA for the NEXT phase instead (stash_issue's w0) -- a wave's loads return in order, so a look returns behind whatever the SAME wave requested before it, and with every wave
    // carrying its share of the 106 KiB [W1; W3] stash the Wo workgroups saw the heads' lines ~4 us after they went up.  16: every wave holds steps (FFN2, every other user).
    u32 nw;

    static __device__ __forceinline__ u32 inv_of(u32 d) { return d > 1 ? 0xFFFFFFFFu / d + 1u : 0u; }
    static __device__ __forceinline__ u32 udiv(u32 x, u32 d, u32 inv) { return d > 1 ? __umulhi(x, inv) : x; }

    // ctr_slot: which of the two step counters in LDS this GEMV uses
    // write_ctr = false: a temporary context that only issues stash loads for a later phase (the LDS counter belongs to the running phase)
    __device__ __forceinline__ void init(const GemvArgs& a, u32 wg_, u32 nwg_, char* lds, u32 ctr_slot = 0, u32 st_base_ = 0, u32 st_n_ = 0, bool write_ctr = true) {
        n = a.n; wg = wg_; nwg = nwg_; st_base = st_base_; st_n = st_n_;
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
        stored = false; primedA = primedB = false; resid_src = nullptr; gsrc = nullptr; gron = false; gtag = 0; nw = kWavesPerBlock;
        // the first two steps of every wave are fixed (wave, wave + 16): they are requested before any barrier (the stash's steps are the next numbers)
        if (write_ctr && threadIdx.x == 0) *reinterpret_cast<u32*>(lds + ctr_off) = 2 * kWavesPerBlock;
    }

    // (a context that was initialised with write_ctr = false while another GEMV was using the LDS)
    __device__ __forceinline__ void reset_ctr(char* lds) const { if (threadIdx.x == 0) *reinterpret_cast<u32*>(lds + ctr_off) = 2 * kWavesPerBlock; }

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
    __device__ __forceinline__ void load_step(Set& S, u32 s, int ablate, const char* lds = nullptr) const {
        u32 wo, so;
        decode(s, S, wo, so);
        if constexpr (STASH) {
            const u32 si = s - 2 * kWavesPerBlock;
            if (si < st_n) {                                                   // (wave-uniform) the step lies in the stash: lane for lane what the loads below deliver
                const char* p = lds + st_base + si * kSlotBytes;
#pragma unroll
                for (int j = 0; j < H; ++j) S.w[j] = *reinterpret_cast<const v4i*>(p + j * 1024 + lane * 16);
                S.sw = *reinterpret_cast<const float*>(p + H * 1024 + lane * 4);
                return;
            }
        }
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
        if (part != 2) { load_step(setA, wave < nw ? wave : NS, ablate); primedA = true; }               // (a wave without steps: past the end, no memory access)
        if (part != 1) { load_step(setB, wave < nw ? wave + nw : NS, ablate); primedB = true; }
    }

    // this wave's stash slots: the step's H weight blocks and its scale dwords go straight to LDS, lane for lane what load_step puts into registers
    // (LDS address = M0 + 16 (4) x lane; blocks past the step's live ones / steps past the end get the out-of-range offset: no memory access)
    // w0: only the waves [w0, 16) issue (the workgroup's slots dealt among them); 0: every wave
    __device__ __forceinline__ void stash_issue(const char* lds, const u32 w0 = 0) const {
        const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
        if (wave < w0) return;
        for (u32 i = wave - w0; i < st_n; i += kWavesPerBlock - w0) {
            Set S; u32 wo, so;
            decode(2 * kWavesPerBlock + i, S, wo, so);
            const u32 dst = __builtin_amdgcn_readfirstlane(lds0 + st_base + i * kSlotBytes);
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const u32 woj = (u32)j < S.nlive ? lane_woff + wo + j * dW : kOOB;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen nt lds" :: "v"(woj), "s"(rW), "s"(dst + j * 1024) : "memory", "m0");
            }
            const u32 svo = lane_j < S.nlive ? lane_s2off + so : kOOB;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen nt lds" :: "v"(svo), "s"(rS), "s"(dst + H * 1024) : "memory", "m0");
        }
    }
    // whichever of the two fixed sets this wave has not requested yet
    __device__ __forceinline__ void issue_missing(int ablate) {
        if (!primedA) { load_step(setA, wave < nw ? wave : NS, ablate); primedA = true; }
        if (!primedB) { load_step(setB, wave < nw ? wave + nw : NS, ablate); primedB = true; }
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
        if constexpr (EPI == EPI_RESIDUAL) { if (rv) resid = gsrc ? ld_granule_value(gsrc + row) : ld_agent((resid_src ? resid_src : a.out) + row); }
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
        if (lane < Rm && !(kAblate && (a.ablate & 1))) {
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
                if constexpr (EPI == EPI_STORE) st_result(a, row, acc);
                else if (gron) st_result_granule(a, row, gtag, __fadd_rn(resid, acc));
                else st_result(a, row, __fadd_rn(resid, acc));       // o.add(tmp, offset) transformer.cpp:465,493
            }
        } else if constexpr (EPI == EPI_SWIGLU) {
            if (rv) { const float hv = swiglu_elem(acc, acc2); if (gron) st_result_granule(a, row, gtag, hv); else st_result(a, row, hv); }       // o1.swiglu(o3) transformer.cpp:481
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
                    if (row < (u32)a.dim) {
                        if (gron) { st_granule(a.gout + row, gtag, o0); st_granule(a.gout + row + 1, gtag, o1); }     // (the head sweeps q's granules; nothing else reads q)
                        else { st_agent(a.out + row, o0); st_agent(a.out + row + 1, o1); }
                    }
                    else {
                        float* kp = a.kcache + ((size_t)h * a.max_seq + pos) * hs + d; st_agent(kp, o0); st_agent(kp + 1, o1);
                        if (gron) { st_granule(a.gk + rr, gtag, o0); st_granule(a.gk + rr + 1, gtag, o1); }     // (the cache row is for the NEXT tokens; this token's head reads the granules)
                    }
                } else {
                    const u32 rr = row - a.dim - a.kv_dim;
                    const u32 h = rr / hs, d = rr - h * hs;
                    float* vp = a.vcache + ((size_t)h * a.max_seq + pos) * hs + d; st_agent(vp, x0); st_agent(vp + 1, x1);
                    if (gron) { st_granule(a.gv + rr, gtag, x0); st_granule(a.gv + rr + 1, gtag, x1); }
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
        auto finish_upto = [&](u32 limit) {
            while (it < limit) {
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
        };
        auto do_set = [&](Set& S) -> bool {
            finish_upto(S.itl);
            if (S.itl >= np_wg) return false;                                  // no work left (every later number is past the end too)
            reduce_step(S, lds, lds + off_scr + (nbuf > 1 ? (S.itl & 1) * buf_bytes : 0u), a.ablate);
            u32 s = 0;
            if (lane == 0) s = atomicAdd(ctr, 1u);
            load_step(S, __builtin_amdgcn_readfirstlane(s), a.ablate, lds);    // refill this set: a full cycle ahead
#if !defined(FLM_TRACE_PRO) && !defined(FLM_TRACE_BAR)
            if (!tr3) { tr3 = true; stamp(3); }
#endif
            return true;
        };
        if (!primedA) load_step(setA, wave, a.ablate);                         // (a wave that was busy elsewhere during the prologue)
        if (!primedB) load_step(setB, wave + kWavesPerBlock, a.ablate);
        while (do_set(setA) && do_set(setB)) {}
    }

    // ---- arrival-order activation (round 5) --------------------------------------------------------------------------------------------------------------
    // The activation is produced by other workgroups of the SAME launch (the heads' output for Wo, FFN13's hd for FFN2), each raising its own flag line when its
    // slice is in memory.  The classic hand-off (poll_lines + gemv_prologue + run) polls ALL lines, then the whole workgroup fetches / quantizes the whole vector
    // between two barriers: the consumer's integer work starts when the SLOWEST producer has finished, plus a poll, the vector's read, the quantizer and the barriers.
    // The int32 group dots are order-free and a step's column block depends on a handful of producers only.  So here every wave looks after its own steps
    // (wave + 16 k: k = 0, 1 the two register sets, k = 2, 3 the stash slots the wave itself filled): lanes [16 k, 16 k + 16) poll the lines of the producers of
    // step k's column block; when a block's lines are up the wave reads its elements with coherent loads, puts them -- quantized by itself (quant_round4: the same
    // operations on the same values, whoever runs them) or copied (PRO_NONE: the heads hand their output over quantized) -- where the block lives in LDS, and reduces
    // the step.  No workgroup barrier before the one in front of the fp32 chains; what is behind the LAST producer's flag is one look, one 1 KiB read, one wave's
    // quantizer round, one step, the chains.  The host guarantees (BackArgs::ao_*): one pass per workgroup, every step resident (NS <= 32 + st_n), a column block
    // of <= 256 elements (PRO_QUANT) with <= 16 producers.
    struct AoSrc {
        const unsigned* flags;          // the producers' flag lines (kFlagStride dwords apart)
        u32 rows, grid;                 // element i of the vector comes from line (i / rows) % grid
        unsigned target; int* err;
    };
    struct AoBlock { v4i d; float s; };
    template <int PRO>
    __device__ __forceinline__ AoBlock ao_fetch(const GemvArgs& a, u32 cc) const {
        AoBlock b; b.s = 0.f;
        const u32 CB = 1u << cbs;
        if constexpr (PRO == PRO_NONE) {
            // the block's CB chunks of 16 quantized bytes and its CB / LPG group scales
            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.xq), 0, (int)rowbytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xs), 0, (int)(sn * 4), 0x00020000);
            b.d = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rq, (int)(lane < CB ? ((cc << cbs) + lane) * 16 : kOOB), 0, kAuxCoherent));
            b.s = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(lane < (CB >> LPGS) ? (((cc << cbs) >> LPGS) + lane) * 4 : kOOB), 0, kAuxCoherent));
        } else {
            // the block's fp32 elements: lane l owns 4 l .. 4 l + 3 of them (16 lanes = one 64-group)
            const u32 EPB = (16u << cbs) / T::kEsz;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)(n * 4), 0x00020000);
            b.d = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(4 * lane < EPB ? (cc * EPB + 4 * lane) * 4 : kOOB), 0, kAuxCoherent));
        }
        return b;
    }
    template <int PRO>
    __device__ __forceinline__ void ao_place(char* lds, u32 cc, const AoBlock& b) const {
        char* xq = lds; float* xs = reinterpret_cast<float*>(lds + off_xs);
        const u32 CB = 1u << cbs;
        if constexpr (PRO == PRO_NONE) {
            if (lane < CB) *reinterpret_cast<v4i*>(xq + ((cc << cbs) + lane) * 16) = b.d;
            if (lane < (CB >> LPGS)) xs[((cc << cbs) >> LPGS) + lane] = b.s;
        } else {
            const u32 EPB = (16u << cbs) / T::kEsz, e = cc * EPB + 4 * lane;
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f t = __builtin_bit_cast(v4f, b.d);
            quant_round4<QT>(xq, xs, (int)e, 4 * lane < EPB && e < n, make_float4(t.x, t.y, t.z, t.w));
        }
    }
    // mid(): called once, when the first look has come back and the loads of the blocks it found complete are out (what the caller still wants to request: the rest of the weights)
    template <int PRO, class Mid, class Stamp>
    __device__ __forceinline__ void run_ao(const GemvArgs& a, char* lds, const AoSrc& src, Mid&& mid, Stamp&& stamp) {
        xa = v4i{0, 0, 0, 0}; sx2 = 0.f; cur_xo = 0xffffffffu;
        char* strips = lds + off_scr;
        // lane (k, j) = (lane >> 4, lane & 15): producer j of the column block of step wave + 16 k
        const u32 EPB = (16u << cbs) / T::kEsz, inv_rows = inv_of(src.rows);
        const u32 ks = wave < nw ? wave + nw * (lane >> 4) : NS;                 // (one pass per workgroup: a step number is the remainder inside the pass)
        const u32 kq = udiv(ks, NBCV, inv_NBCV), kc = ks - kq * NBCV;            // its row chunk, its column block
        const u32 e0 = kc * EPB, e1 = e0 + EPB < n ? e0 + EPB : n;
        const u32 p0 = udiv(e0, src.rows, inv_rows), p1 = udiv(e1 - 1, src.rows, inv_rows), pj = p0 + (lane & 15);
        const bool mine = ks < NS && pj <= p1;
        const unsigned* line = src.flags + (pj % src.grid) * 16 /* kFlagStride */;
        u32 pending = 0;
#pragma unroll
        for (u32 k = 0; k < 4; ++k) if (wave < nw && wave + nw * k < NS) pending |= 1u << k;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool first = true;
        u32 spins = 0;
        while (pending) {
            const unsigned f = mine ? __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src.target;
            const unsigned long long okm = __ballot((int)(f - src.target) >= 0);
            u32 now = 0;
#pragma unroll
            for (u32 k = 0; k < 4; ++k) if (((okm >> (16 * k)) & 0xFFFFull) == 0xFFFFull) now |= 1u << k;
            now &= pending;
            if (!now) {
                // 20 ms: the host re-runs the call on one kernel per phase.  (And when ANOTHER wait of the launch has already given up, do not spend this one's 20 ms as well:
                // a look at *err every 64 empty looks.)
                const bool late = __builtin_amdgcn_s_memrealtime() - t0 > 2000000ull;
                if (late) __hip_atomic_store(src.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (late || ((++spins & 63u) == 0u && __hip_atomic_load(src.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) now = pending;
            }
            AoBlock b[4];
#pragma unroll
            for (u32 k = 0; k < 4; ++k) if (now & (1u << k)) b[k] = ao_fetch<PRO>(a, (u32)__builtin_amdgcn_readlane((int)kc, 16 * k));
            if (first) { first = false; mid(); }
#pragma unroll
            for (u32 k = 0; k < 4; ++k) {
                if (now & (1u << k)) {
                    ao_place<PRO>(lds, (u32)__builtin_amdgcn_readlane((int)kc, 16 * k), b[k]);
                    asm volatile("" ::: "memory");                                 // (the same wave reads the block back: LDS keeps a wave's accesses in order)
                    cur_xo = 0xffffffffu;                                          // (the block's place in LDS has just been written: reduce_step fetches this lane's chunk)
                    if (k == 0) reduce_step(setA, lds, strips, a.ablate);
                    else if (k == 1) reduce_step(setB, lds, strips, a.ablate);
                    else {
                        wait_stores_done();                                        // this wave's own LDS-DMA into its stash slots
                        Set S;
                        load_step(S, wave + nw * k, a.ablate, lds);
                        reduce_step(S, lds, strips, a.ablate);
                    }
                }
            }
            if (pending && now == pending) stamp(3);
            pending &= ~now;
        }
        stamp(4);
        if (np_wg) finish_pass(a, wg, 0, strips, 0);
        stamp(5);
    }
};

// Tensor parallel, peer to peer: the flag round of an exchange inside the launch that CONSUMES the exchanged vector (instead of a one-workgroup
// kernel, k_xchg, between producer and consumer: 4 of a sharded layer's 9 launches).  The producing launch of this rank has completed (kernel boundary:
// its system-scope stores into every peer's buffer are done), so workgroup 0 tells every peer "my slice of exchange e is in your memory" (release fence,
// one flag line per (slot, rank)); every workgroup then waits until all ranks' lines in the LOCAL buffer have reached e and acquires.  e comes from the
// token's epoch base in device memory (the same on every rank: all ranks run the same tokens), so a captured graph replays correctly.
constexpr int kXchgSlots = 8;                  // flag slots: 0..3 k_xchg's kinds (att, x1, hd, logits), 4..6 the folded exchanges (att, x1, hd)
constexpr int kXchgAbortLine = kXchgSlots * 8; // flag line behind the [slot][8 ranks] lines: non-zero = some rank gave up, nobody waits any more
__device__ __forceinline__ void xchg_fold(const GemvArgs::XchgFold& x) {
    const unsigned e = *x.base + x.add;
    const int r = threadIdx.x;
    if (threadIdx.x < 64) {
        if (blockIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);
            if (r < x.world) __hip_atomic_store(x.peer_flags[r] + (x.slot * 8 + x.rank) * 16, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (true) {
            const unsigned f = r < x.world ? __hip_atomic_load(x.local_flags + (x.slot * 8 + r) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : e;
            if (__all((int)(f - e) >= 0)) break;
            const bool aborted = __hip_atomic_load(x.local_flags + kXchgAbortLine * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
            if (aborted || __builtin_amdgcn_s_memrealtime() - t0 > 2000000000ull) {                               // 20 s: ranks start seconds apart
                __hip_atomic_store(x.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (r < x.world) __hip_atomic_store(x.peer_flags[r] + kXchgAbortLine * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

// COH: the fp32 activation a.x was written by peer GPUs (tensor parallel) -> system-coherent loads
template <int QT, int PRO, int EPI, int XR, bool COH = false>
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
    if (kAblate && (a.ablate & 16)) return;
    // The activation first, and the weight prefetch only once it HAS ARRIVED (the hook runs after the staging barrier /
    // after the activation registers landed).  Weights do not depend on the activation and were once requested up
    // front -- but workgroups start ~1 us apart, and the activation loads of the late ones then queued in HBM behind
    // 32 MB of weight requests of the early ones: the activation came back 2.6 us later (measured), delaying the
    // whole rmsnorm chain.  Issued after the activation, the first 128 KiB per CU still arrive under the chain.
    if constexpr (COH) { if (a.xf.world) xchg_fold(a.xf); }           // tensor parallel: the exchange's flag round, here instead of in a launch of its own
    float4 xv[XR > 0 ? XR : 1], nv[XR > 0 ? XR : 1];
    gemv_preload<QT, PRO, XR, COH>(a, xv, nv);
    GemvCtx<QT, EPI> g;
    g.init(a, blockIdx.x, gridDim.x, lds);
    if constexpr (PRO == PRO_NONE) g.issue(kAblate ? a.ablate : 0);
#ifndef FLM_TRACE_PRO2
    stamp(1);
#endif
    gemv_prologue<QT, PRO, XR, COH>(a, lds, xv, nv, [&](int part) { g.issue(kAblate ? a.ablate : 0, part); });
    stamp(2);
    if (kAblate && (a.ablate & 32)) return;
    g.run(a, lds, stamp);
    stamp(6);
    if (kAblate && a.trace && threadIdx.x == 0) {
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        a.trace[blockIdx.x * 8 + 7] = rt1 - rt0;
        if (a.ablate & 64) { a.trace[blockIdx.x * 8 + 1] = rt0; a.trace[blockIdx.x * 8 + 2] = rt1; }   // tools/trace_skew.py: absolute 100 MHz stamps (one clock for all XCDs)
    }
}

} // namespace flm
