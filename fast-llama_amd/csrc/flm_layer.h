// flm_layer.h -- the back half of a decoder layer as ONE launch (single GPU): attention, Wo + residual, FFN13 + SwiGLU, FFN2 + residual
// (transformer.cpp:136-152: ATTN, ATTN_O, rmsnorm, quantize, FFN13, quantize, FFN2), with the weight stream running through the attention.
// Part of flm_kernels.h (hand-written gfx950 / CDNA4 kernels of the fast-llama per-token hot path); include that header.
#pragma once
#include "flm_math.h"
#include "flm_gemv.h"
#include "flm_attn.h"
// (bit-exactness hygiene: see flm_math.h -- no implicit FMA contraction in any of these headers)
#pragma clang fp contract(off)

namespace flm {

// ------------------------------------------------------------------------------------------
// k_attn_ffn = k_attn_o + k_ffn in one launch, phases and arithmetic verbatim (k_gemv's GemvCtx, attn_head), plus what only one launch can do:
//   * while the 32 head workgroups run their ~6 us of dependent fp32 chains, the other 224 CUs have nothing in flight once their share of Wo (79 KiB,
//     registers) has landed, and 160 KiB of LDS each that nothing uses: they fetch the first steps of THEIR rows of [W1; W3] into that LDS with LDS-DMA
//     (GemvCtx::stash_issue: no registers, no VALU) -- ~100 KiB per CU, a quarter of FFN13's bytes, off the streaming phase's clock;
//   * the x1 hand-off (Wo's rows -> every workgroup's rmsnorm) is a flag round instead of a kernel boundary + launch ramp;
//   * the head workgroups fetch their stash when their head is done (they wait for x1 anyway), every workgroup fetches a stash of W2 behind its own
//     rows of hd (it waits for the slowest workgroup's hd anyway).
// A stash is bounded (<= 25 steps of 4.25 KiB): what is requested in front of a flag poll delays that poll by the time the CU's in-order memory
// pipeline needs for it (~30 KB/us), which the median workgroup spends waiting for the slowest producer in any case.
// Hand-offs as in k_attn_o / k_ffn: write-through stores, s_waitcnt vmcnt(0), one 64-byte flag line per producer workgroup (value layer + 1, cleared by
// k_embed), lane i polls line i, coherent loads of the vector.  All workgroups resident (grid <= CUs: the census of flm_ctx_create); a poll that never
// succeeds gives up after ~20 ms and raises *err (the host re-runs the call on one kernel per phase).
// ------------------------------------------------------------------------------------------
struct BackArgs {
    int n_heads;                // head workgroups [0, n_heads); Wo workgroups [n_heads, n_heads + grido)
    int grido, grid13, grid2;
    int gridq;                  // QKV = true: the workgroups [0, gridq) of the QKV GEMV in front
    unsigned* flag_q;           // ... and their lines
    unsigned* flag_h;           // heads' lines
    unsigned* flag_x;           // Wo workgroups' lines (x1)
    unsigned* flag_hd;          // FFN13 workgroups' lines (hd)
    unsigned* flag_x2;          // k_layers: FFN2 workgroups' lines (the residual stream x for the next layer's QKV)
    int nstq, preq;             // k_layers: stash slots / early waves of [Wq; Wk; Wv] requested in front of the x poll
    unsigned target; int* err;
    unsigned st_base;           // LDS byte offset of the stash slots (above every phase's own layout)
    int nst13;                  // stash slots of [W1; W3] a Wo workgroup fills under the attention
    int nst13_head;             // ... a head workgroup fills when its head is done
    int nst2;                   // stash slots of W2 every workgroup fills behind its rows of hd
    int pre2;                   // the first pre2 waves request their first register set of W2 before the hd flag round
    int pre13;                  // the first pre13 waves of a workgroup request their first register set of [W1; W3] before the x1 flag round (16: all, as k_ffn does for W2)
    // arrival-order hand-offs (round 5; GemvCtx::run_ao): the consumer's waves poll the producers of their own steps' column blocks instead of the workgroup polling all lines
    int ao_o;                   // Wo: a wave copies a column block of the heads' quantized output when the block's heads have raised their lines (one workgroup per head)
    int ao_2;                   // FFN2: a wave quantizes a column block of hd when the block's FFN13 workgroups have raised theirs; 1: W2 whole (both sets + stash) in front of the first look,
                                // 2: the first register sets in front, the rest behind the first look, 3: first sets + stash in front, the second sets behind
    int r5;                     // the round-5 forms the launch's instantiation carries (layer_body's R5): 3 or 0
    unsigned kpre_off;          // split heads: LDS byte offset of the two K tiles a part's workgroup brings in by LDS-DMA under the QKV phase (attn_kpre_issue; 0: off)
    int nw_o;                   // arrival-order Wo: the waves [0, nw_o) hold Wo's steps and look for their heads, the waves [nw_o, 16) issue the workgroup's [W1; W3] stash (GemvCtx::nw); 16: every wave does both
    int nst2_ao;                // ... its stash slots: the steps beyond the two register sets (all of W2's share is resident)
    unsigned long long* trace;  // FLM_ABLATE builds: [grid][16] s_memrealtime stamps (100 MHz, one clock for all XCDs; tools/trace_back.py)
    // round 6: the residual stream as data-tagged granules inside the one-launch token (flm_gemv.h: granule_t): xg_a = x behind FFN2 (the next layer's / the classifier's input),
    // xg_b = x1 behind Wo (FFN13's input), [dim] each.  gr = 1: the two strict all-to-all edges of a layer carry no flag round -- the consumers sweep the granules themselves.
    // (TAIL launches only: their flag values count from an epoch and never repeat; the launch's first layer reads the embedding row, nothing else reads x.)
    // Tensor parallel (k_layers<.., TP>): ALL four cross-rank vectors of a layer are granules in every rank's exchange buffer -- xg_att = the heads' fp32 output, xg_hd = FFN13's hd
    // too --, so the rank-spanning launch raises no line and needs no fence: a granule is ONE 8-byte store (over xGMI: one write), and nothing is inferred from the order of stores.
    // xg_hd also serves any launch whose FFN2 takes hd in its all-to-all form (R5 bit 1 clear: int16 7B).  gres_off: this rank's first row of the residual stream (0 on one GPU).
    granule_t* xg_a; granule_t* xg_b; granule_t* xg_att; granule_t* xg_hd; int gr; int gres_off;
    // tensor parallel (layer_body<.., TP>, round 6): the launch SPANS the ranks -- every rank runs the same k_layers on its rows (heads, rows of Wo / W2, rows of [W1; W3]: the
    // reference's row split, transformer.cpp:264-287), the four all-to-all hand-offs of a layer cross the ranks.  A producer stores its slice into every rank's buffer
    // (GemvArgs::out_peer / AttnArgs::out_peer) and raises ITS line in every rank's array; a consumer polls the lines of ALL ranks' producers in its LOCAL array.  The ranks'
    // geometry is identical (equal shards, equal CU counts: checked when the group forms), so rank r's workgroup w of a kind is line r * (lines per rank) + w.
    // flag_h / flag_x / flag_hd / flag_x2 above then point into this rank's own region (peer[rank] + off_*); flag_q stays a local array (a head's q / k / v rows are its own rank's).
    struct Tp {
        int world, rank;
        unsigned* peer[8];      // every rank's flag region (in its exchange buffer, mapped here); peer[rank]: this rank's own
        unsigned off_h, off_x, off_hd, off_x2, off_cls;   // dword offsets of the kinds' line arrays inside a region
        int head_line0;         // this rank's first head-part line (plan.head_begin * G); the heads' array has n_heads_all lines
        int n_heads_all;        // head parts of the whole model
        const unsigned* base;   // the token's epoch base (device memory, advanced by k_embed on every rank alike): the launch's flag values count from it, nobody clears a line
        int abort_off;          // dword offset (signed: the line lies in front) of the group's abort line from a region's start (non-zero line: some rank gave up, nobody waits any more)
        int fence;              // bit 0: release fence in front of a raised line, bit 1: acquire fence behind a completed poll (both on: what the flags promise across xGMI)
    } tp;
};

// One wave's look at its lines until all have reached `target`.  A look is a round trip to memory (the lines are written and read across XCDs); with one look in flight a
// raised line is noticed half a round trip late on average.  FLM_POLL_DEPTH > 1 keeps that many looks in flight, FLM_POLL_GAP apart -- measured (tools/variants.sh): depth 3
// takes 0.2 .. 0.3 us off a short-context layer and ADDS 1.6 us to a long-context one (the split heads' loads queue behind three times the polling traffic): depth 1 stays.
#ifndef FLM_POLL_DEPTH
#define FLM_POLL_DEPTH 1
#endif
#ifndef FLM_POLL_GAP
#define FLM_POLL_GAP 6            // s_sleep units (64 cycles) between two looks
#endif
__device__ __forceinline__ void poll_wave(const unsigned* line, bool mine, unsigned target, int* err) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    auto look = [&]() -> unsigned { return mine ? __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target; };
#if FLM_POLL_DEPTH <= 1
    // (a wait of this launch has already given up: the host re-runs the call anyway; do not spend another 20 ms here -- the look at *err flies beside the first look at the lines)
    const int gave_up = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned f = look();
    if (gave_up) return;
    while (true) {
        if (__all((int)(f - target) >= 0)) break;
        f = look();
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
#else
    unsigned f[FLM_POLL_DEPTH];
#pragma unroll
    for (int i = 0; i < FLM_POLL_DEPTH; ++i) { f[i] = look(); if (i + 1 < FLM_POLL_DEPTH) __builtin_amdgcn_s_sleep(FLM_POLL_GAP); }
    while (true) {
        bool done = false;
#pragma unroll
        for (int i = 0; i < FLM_POLL_DEPTH; ++i) {
            if (__all((int)(f[i] - target) >= 0)) { done = true; break; }
            f[i] = look();
            __builtin_amdgcn_s_sleep(FLM_POLL_GAP);
        }
        if (done) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
#endif
}
// lane i (of the first 256 threads) polls line i of `n` lines until all have reached `target`
__device__ __forceinline__ void poll_lines(const unsigned* flag, int n, unsigned target, int* err) {
    if ((int)(threadIdx.x & ~63u) < n) poll_wave(flag + threadIdx.x * kFlagStride, (int)threadIdx.x < n, target, err);
}
// ---- the same across tensor-parallel ranks (layer_body<.., TP>): the lines are written by peer GPUs (system-scope stores into this rank's region), so the looks are
// system-scope loads; ranks start seconds apart (module loading, graph capture), so a wait is patient (20 s) -- but once ANY rank has given up (the abort line) everybody leaves.
// *err = 2: a group error (FLM_ERR_COMM; nothing is re-run on one rank alone).
__device__ __forceinline__ void poll_wave_tp(const unsigned* line, bool mine, unsigned target, const BackArgs& p) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    while (true) {
        const unsigned f = mine ? __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : target;
        if (__all((int)(f - target) >= 0)) break;
        if ((++spins & 255u) == 0u) {
            const bool aborted = __hip_atomic_load(p.tp.peer[p.tp.rank] + p.tp.abort_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (aborted || __builtin_amdgcn_s_memrealtime() - t0 > 2000000000ull) {
                __hip_atomic_store(p.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                for (int r = 0; r < p.tp.world; ++r) __hip_atomic_store(p.tp.peer[r] + p.tp.abort_off, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
}
// every thread takes the lines tid, tid + 1024, ... of `n` (<= 8 ranks x 256 workgroups); the caller's barrier follows
template <bool TP>
__device__ __forceinline__ void poll_lines_t(const unsigned* flag, int n, unsigned target, const BackArgs& p) {
    if constexpr (!TP) poll_lines(flag, n, target, p.err);
    else {
        for (int b = 0; b < n; b += kGemvBlock) {
            const int i = b + (int)threadIdx.x;
            if ((i & ~63) < n) poll_wave_tp(flag + (size_t)i * kFlagStride, i < n, target, p);
        }
        if (p.tp.fence & 2) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
}
// a producer workgroup's line goes up (its stores have completed: wait_stores_done + barrier in front): one local store, or one store into every rank's array
template <bool TP>
__device__ __forceinline__ void raise_line(const BackArgs& p, unsigned* local, unsigned off, unsigned line, unsigned target) {
    if constexpr (!TP) { if (threadIdx.x == 0) __hip_atomic_store(local + line * kFlagStride, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else if (threadIdx.x == 0) {
        if (p.tp.fence & 1) __atomic_thread_fence(__ATOMIC_RELEASE);
        for (int r = 0; r < p.tp.world; ++r) __hip_atomic_store(p.tp.peer[r] + off + line * kFlagStride, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// gemv_preload on a granule vector (flm_gemv.h: granule_t): xv[i] = elements 4 tid + 4096 i .. + 3 once their tags are `tag` (every thread re-reads ITS granules until they
// match; only the lanes whose granules are missing read again); wv[i] = the norm weights (PRO_RMSNORM_QUANT).  XR rounds cover the vector (the host checks).  A wait of ~20 ms
// raises *err (the host re-runs the call on one kernel per phase); TP: patient (20 s: ranks start seconds apart), leaves when any rank has given up (the abort line), *err = 2.
template <int PRO, int XR, bool TP>
__device__ __forceinline__ void gemv_preload_granules(const GemvArgs& a, float4 (&xv)[XR], float4 (&wv)[XR], const granule_t* g, const unsigned tag, const BackArgs& p) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<granule_t*>(g), 0, a.n * 8, 0x00020000);
    const int gave_up = __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (PRO == PRO_RMSNORM_QUANT) {
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.norm_w), 0, a.n * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const v4f u = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rn, (int)((threadIdx.x * 4 + i * kGemvBlock * 4) * 4), 0, 0));
            wv[i] = make_float4(u.x, u.y, u.z, u.w);
        }
    }
    v4u32 A[XR], C[XR]; bool ok[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) { A[i] = v4u32{0u, 0u, 0u, 0u}; C[i] = A[i]; ok[i] = (int)(threadIdx.x * 4 + i * kGemvBlock * 4) >= a.n; }   // (past the vector: zeros, as gemv_preload's bounds-checked loads)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    while (true) {
        asm volatile("" ::: "memory");                                         // (the loads are re-issued every pass)
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            if (!ok[i]) {
                const int off = (int)((threadIdx.x * 4 + i * kGemvBlock * 4) * 8);
                A[i] = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, kAuxCoherent));
                C[i] = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rg, off + 16, 0, kAuxCoherent));
            }
        }
        bool all = true;
#pragma unroll
        for (int i = 0; i < XR; ++i) { if (!ok[i]) ok[i] = A[i].y == tag && A[i].w == tag && C[i].y == tag && C[i].w == tag; all = all && ok[i]; }
        if (__all(all) || gave_up) break;
        if constexpr (!TP) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        } else if ((++spins & 255u) == 0u) {
            const bool aborted = __hip_atomic_load(p.tp.peer[p.tp.rank] + p.tp.abort_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (aborted || __builtin_amdgcn_s_memrealtime() - t0 > 2000000000ull) {
                __hip_atomic_store(p.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                for (int r = 0; r < p.tp.world; ++r) __hip_atomic_store(p.tp.peer[r] + p.tp.abort_off, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < XR; ++i) xv[i] = make_float4(__uint_as_float(A[i].x), __uint_as_float(A[i].z), __uint_as_float(C[i].x), __uint_as_float(C[i].z));
}

#ifndef FLM_BACK_LATE
#define FLM_BACK_LATE 1
#endif
#ifndef FLM_BACK_LATE2
#define FLM_BACK_LATE2 0          // FFN2's prologue: its second register set requested BEHIND the quantizer: measured slower (the phase is bound by the stream behind hd's arrival)
#endif
#ifndef FLM_LAYER_LATEQ
#define FLM_LAYER_LATEQ 1
#endif
// QKV = true: the whole layer -- the QKV GEMV (k_gemv<RMSNORM_QUANT, ROPE_KV> verbatim, every workgroup its rows of [Wq; Wk; Wv]) in front, as in k_qkv_attn_o: a head waits for
// the lines of the <= 3 * ceil(hs / Rm + 1) workgroups that reduced its rows of q, k and v and reads them with coherent loads; the other workgroups go straight
// from their last QKV row to the Wo / stash requests, so the memory pipeline has work across what used to be a kernel boundary and a launch ramp.
// SPLIT (long contexts): a head is spread over G = aa.G workgroups (attn_head<.., SPLIT>); n_heads counts head PARTS; a part's 32 output dimensions are half a quant group,
// so the Wo workgroups fetch the fp32 vector and quantize it themselves (dim <= 4096: one round).
// PERSIST (k_layers: several layers in one launch): xpoll = the layer's x comes from the previous layer's FFN2 in the SAME launch -- [Wq; Wk; Wv]'s first register sets (preq waves) and
// nstq stash slots are requested, then the FFN2 lines are polled and x is read with coherent loads; xflag = this layer's FFN2 raises its lines for the next layer.
// R5 (round 5; compile time, so that an instantiation carries one form of every hand-off -- both forms in one kernel spill): bit 0 Wo consumes the heads' output in arrival
// order (GemvCtx::run_ao; one workgroup per head), bit 1 FFN2 consumes hd in arrival order.  The host picks the instantiation whose forms the shape allows (plan_layer: BackArgs::r5).
// TP (round 6): the launch spans the tensor-parallel ranks (BackArgs::Tp).  The heads hand their output over as fp32 (every rank's Wo workgroups quantize it themselves, as
// with split heads); R5 = 0 (the hand-offs in their all-to-all form).
template <int QT, int XR2, bool QKV, bool SPLIT, bool PERSIST, int R5 = 0, bool TP = false, int GRM = 0>
// GRM: the granule form of the hand-offs (BackArgs::gr): 0 the instantiation runs flag rounds, 2 granules -- a compile-time choice (both forms in one kernel spill): the one-launch token
// (k_layers<.., TAIL>) IS the granule form (a shape that cannot run it has no one-launch token: plan_layer), the rank-spanning launch has both instantiations (k_layers<.., TP, GRT>).
// x0 (k_layers' one-launch token, first layer): the layer's input is read from there -- the embedding row -- by the QKV prologue and by Wo's residual epilogue instead of the residual
// stream's buffer (which Wo's rows then start); null: the buffer.
__device__ __forceinline__ void layer_body(const GemvArgs& aq, const AttnArgs& aa, const GemvArgs& ao, const GemvArgs& a13, const GemvArgs& a2, const BackArgs& p, char* lds,
                                           const unsigned target, const bool xpoll, const bool xflag, const bool tracing, const float* x0 = nullptr) {
    auto nostamp = [](int) {};
    auto stamp = [&](int k) { if (kAblate && tracing && p.trace && threadIdx.x == 0) p.trace[blockIdx.x * 16 + k] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    constexpr bool gr = PERSIST && GRM == 2;                                    // the x / x1 edges as granules
    constexpr bool gr_hd = gr && (R5 & 2) == 0;                                     // ... and hd, where FFN2 takes it in its all-to-all form
    // ... and q + this token's K / V row from the QKV phase to the heads (one GPU): a head of one workgroup has its <= 2 tiles all requested at once; a split head's part has its
    // first two K tiles pre-landed by DMA and the other two in the ring's registers in front of the sweep (attn_head<.., GRIN>)
#ifndef FLM_SPLIT_GRQ
#define FLM_SPLIT_GRQ 1
#endif
    constexpr bool gr_q = gr && !TP && QKV && (FLM_SPLIT_GRQ || !SPLIT);
    constexpr bool gr_att = gr && (TP || SPLIT);                                    // ... and the heads' fp32 output, where the Wo workgroups quantize it themselves (across ranks; split heads)
    unsigned nst13 = 0;
    if constexpr (QKV) {
        if ((int)blockIdx.x < p.gridq) {
            GemvCtx<QT, EPI_ROPE_KV, PERSIST> gq;
            float4 xq[1], nq[1];
            if (!PERSIST || !xpoll) gemv_preload<QT, PRO_RMSNORM_QUANT, 1, PERSIST>(aq, xq, nq, x0);    // (x is there: requested in front of the context's set-up)
            gq.init(aq, blockIdx.x, p.gridq, lds, 0, p.st_base, (PERSIST && xpoll) ? (unsigned)p.nstq : 0u);
            if (gr_q) { gq.gron = true; gq.gtag = target; }
            if constexpr (PERSIST) {
                if (xpoll) {
                    if ((int)gq.wave < p.preq) gq.issue(kAblate ? aq.ablate : 0, 1);
                    gq.stash_issue(lds);
                    if (gr) {
                        gemv_preload_granules<PRO_RMSNORM_QUANT, 1, TP>(aq, xq, nq, p.xg_a, target - 1u, p);     // (no line round: every thread waits for its own elements of x)
                        wait_stores_done();                                     // every wave: the stash slots it requested have landed (the prologue's first barrier follows)
                    } else {
                    poll_lines_t<TP>(p.flag_x2, (TP ? p.tp.world : 1) * p.grid2, target - 1u, p);
                    wait_stores_done();                                         // every wave: the stash slots it requested have landed
                    __syncthreads();
                    gemv_preload<QT, PRO_RMSNORM_QUANT, 1, true>(aq, xq, nq);
                    }
                }
            }
            if constexpr (SPLIT) {
                // a split head's part: its first two K tiles (earlier tokens' rows: they depend on nothing of this layer) by LDS-DMA, now -- x is in registers, the rmsnorm chain and
                // the quantizer follow, the memory pipeline has only the early register sets in it -- instead of behind the QKV phase, where the part's scores waited for them
                if ((int)blockIdx.x < p.n_heads && p.kpre_off) attn_kpre_issue(aa, blockIdx.x / aa.G, blockIdx.x % aa.G, aa.G, *aa.pos_ptr + 1, lds, p.kpre_off);
            }
            gemv_prologue<QT, PRO_RMSNORM_QUANT, 1, PERSIST, FLM_LAYER_LATEQ != 0>(aq, lds, xq, nq, [&](int) { gq.issue_missing(kAblate ? aq.ablate : 0); });
            stamp(12);
            gq.run(aq, lds, nostamp);
            stamp(13);
            if (gr_q) __syncthreads();                                          // (the LDS is free for the next phase; q and the new K / V row travel as granules)
            else {
            wait_stores_done();                                                 // every wave: its q / cache rows are where the heads will read them
            __syncthreads();                                                    // (and the LDS is free for the next phase)
            if (threadIdx.x == 0) __hip_atomic_store(p.flag_q + blockIdx.x * kFlagStride, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            stamp(14);
        }
    }
    if ((int)blockIdx.x < p.n_heads) {
        const int G = SPLIT ? aa.G : 1, hh = blockIdx.x / G;
        // the lines of the workgroups that reduced this head's rows of q, k and v: polled INSIDE the head (attn_head<.., PRE>), behind its requests for the earlier tokens' cache rows
        auto qwait = [&]() {
            if (threadIdx.x < 256) {
                // lane i: does workgroup i reduce a row of this head's q, k or v?  (pass p = rows [p Rm, (p + 1) Rm) of [Wq; Wk; Wv], workgroup p mod gridq)
                bool need = false;
                if ((int)threadIdx.x < p.gridq) {
                    const unsigned Rm = aq.rows_per_pass, hs = aa.hs, nq = p.gridq, h = hh;
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const unsigned r0 = (m == 0 ? 0u : m == 1 ? (unsigned)aq.dim : (unsigned)(aq.dim + aq.kv_dim)) + h * hs;
                        const unsigned pa = r0 / Rm, pb = (r0 + hs - 1) / Rm;
                        need |= (threadIdx.x + nq - pa % nq) % nq <= pb - pa;
                    }
                }
                poll_wave(p.flag_q + threadIdx.x * kFlagStride, need, target, p.err);
            }
            __syncthreads();
        };
        if constexpr (QKV) attn_head_any<true, SPLIT, true, gr_q>(aa, hh, lds, *aa.pos_ptr + 1, aa.q, aa.out, blockIdx.x % G, G, qwait, target, gr_att, gr, (SPLIT && p.kpre_off) ? reinterpret_cast<const float*>(lds + p.kpre_off) : nullptr);
        else attn_head_any<false, SPLIT>(aa, hh, lds, *aa.pos_ptr + 1, aa.q, aa.out, blockIdx.x % G, G);
        stamp(1);
        if (gr_att) __syncthreads();                                            // (the LDS is free; the output's granules are their own flags)
        else {
        wait_stores_done();                                                     // every wave: its part of the head's output is where the others will read it
        __syncthreads();                                                        // (and the LDS is free)
        raise_line<TP>(p, p.flag_h, p.tp.off_h, (TP ? (unsigned)p.tp.head_line0 : 0u) + blockIdx.x, target);
        }
        stamp(2);
        if (p.nst13_head > 0 && (int)blockIdx.x < p.grid13) {
            nst13 = (unsigned)p.nst13_head;
            GemvCtx<QT, EPI_SWIGLU, true> t;
            t.init(a13, blockIdx.x, p.grid13, lds, 0, p.st_base, nst13, false);
            t.stash_issue(lds);
        }
    } else if ((int)blockIdx.x >= p.n_heads + p.grido) {                        // (Wo's passes do not reach this workgroup: the stash only)
        if (p.nst13 > 0 && (int)blockIdx.x < p.grid13) {
            nst13 = (unsigned)p.nst13;
            GemvCtx<QT, EPI_SWIGLU, true> t;
            t.init(a13, blockIdx.x, p.grid13, lds, 0, p.st_base, nst13, false);
            t.stash_issue(lds);
        }
    } else {
        GemvCtx<QT, EPI_RESIDUAL> g;
        g.init(ao, blockIdx.x - p.n_heads, p.grido, lds);
        const unsigned nwo = (!SPLIT && (R5 & 1) != 0 && p.nw_o > 0 && p.nw_o < kWavesPerBlock) ? (unsigned)p.nw_o : (unsigned)kWavesPerBlock;
        g.nw = nwo;
        g.resid_src = x0;
        if (gr) { g.gsrc = xpoll ? p.xg_a + p.gres_off : nullptr; g.gron = true; g.gtag = target; }   // (the launch's first layer: the old value is the embedding row's / the plain vector's)
        g.issue(kAblate ? ao.ablate : 0);
        if (p.nst13 > 0 && (int)blockIdx.x < p.grid13) {
            nst13 = (unsigned)p.nst13;
            GemvCtx<QT, EPI_SWIGLU, true> t;
            t.init(a13, blockIdx.x, p.grid13, lds, 0, p.st_base, nst13, false);
            t.stash_issue(lds, nwo < (unsigned)kWavesPerBlock ? nwo : 0u);     // (the waves that hold no Wo step: their DMA is in front of nobody's look)
        }
        stamp(1);
#ifdef FLM_TRACE_PRO_RT
        // (tools/trace_back.py: when has everything this CU requested -- Wo's register sets and the [W1; W3] stash -- landed?  Wave 15 requested last; loads complete in order)
        if (kAblate && p.trace && threadIdx.x == 960) { wait_stores_done(); p.trace[blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memrealtime(); }
#endif
        if constexpr (!SPLIT && (R5 & 1) != 0) {                                // arrival order: a wave copies a column block when ITS heads' lines are up (GemvCtx::run_ao)
            const typename GemvCtx<QT, EPI_RESIDUAL>::AoSrc src{p.flag_h, (unsigned)aa.hs, (unsigned)p.n_heads, target, p.err};
            g.template run_ao<PRO_NONE>(ao, lds, src, []() {}, [&](int k) { if (k == 3) stamp(2); });
        } else if (gr_att) {
            float4 xv[1], nv[1];
            gemv_preload_granules<PRO_QUANT, 1, TP>(ao, xv, nv, p.xg_att, target, p);
            stamp(2);
            gemv_prologue<QT, PRO_QUANT, 1>(ao, lds, xv, nv, [](int) {});
            g.run(ao, lds, nostamp);
        } else {
        poll_lines_t<TP>(p.flag_h, TP ? p.tp.n_heads_all : p.n_heads, target, p);
        __syncthreads();
        stamp(2);
        float4 xv[1], nv[1];
        if constexpr (SPLIT || TP) {
            gemv_preload<QT, PRO_QUANT, 1, true>(ao, xv, nv);
            gemv_prologue<QT, PRO_QUANT, 1>(ao, lds, xv, nv, [](int) {});
        } else gemv_prologue<QT, PRO_NONE, 0, true>(ao, lds, xv, nv, [](int) {});   // the heads' output arrives quantized (PREQ)
        g.run(ao, lds, nostamp);
        }
        stamp(3);
        if (gr) __syncthreads();                                                // (the LDS is free for the next phase; x1's granules are their own flags)
        else {
        wait_stores_done();                                                     // every wave: its rows of x1 are where the others will read them
        __syncthreads();
        raise_line<TP>(p, p.flag_x, p.tp.off_x, (TP ? (unsigned)(p.tp.rank * p.grido) : 0u) + (blockIdx.x - p.n_heads), target);
        }
        stamp(4);
    }
    // ---- FFN13: k_gemv<RMSNORM_QUANT, SWIGLU> behind the x1 flag round (x1 through coherent loads)
    if ((int)blockIdx.x < p.grid13) {
        GemvCtx<QT, EPI_SWIGLU, true> g;
        g.init(a13, blockIdx.x, p.grid13, lds, 0, p.st_base, nst13);
        if (gr_hd) { g.gron = true; g.gtag = target; }
        if ((int)g.wave < p.pre13) g.issue(kAblate ? a13.ablate : 0, 1);       // the first pre13 waves: their first register set in front of the x1 hand-off
        float4 xv[1], nv[1];
        if (gr) {
            gemv_preload_granules<PRO_RMSNORM_QUANT, 1, TP>(a13, xv, nv, p.xg_b, target, p);
            wait_stores_done();                                                 // every wave: the stash slots it requested have landed
            stamp(5);
        } else {
        poll_lines_t<TP>(p.flag_x, (TP ? p.tp.world : 1) * p.grido, target, p);
        wait_stores_done();                                                     // every wave: the stash slots it requested have landed
        __syncthreads();
        stamp(5);
        gemv_preload<QT, PRO_RMSNORM_QUANT, 1, true>(a13, xv, nv);
        }
        gemv_prologue<QT, PRO_RMSNORM_QUANT, 1, true, FLM_BACK_LATE != 0>(a13, lds, xv, nv, [&](int) { g.issue_missing(kAblate ? a13.ablate : 0); });
        stamp(6);
#ifdef FLM_TRACE_PRO_RT
        // (tools/trace_back.py: thread 0 = the wave that runs the pass's chains: first refill requested / last step reduced, at the barrier / chains, epilogue and stores issued)
        g.run(a13, lds, [&](int k) { if (kAblate && a13.trace && threadIdx.x == 0 && k >= 3 && k <= 5) a13.trace[3 * 4096 + blockIdx.x * 16 + k] = __builtin_amdgcn_s_memrealtime(); });
#else
        g.run(a13, lds, nostamp);
#endif
        stamp(7);
    } else if (!gr) {
        poll_lines_t<TP>(p.flag_x, (TP ? p.tp.world : 1) * p.grido, target, p);       // (keeps the order x1 -> hd for a workgroup without rows)
    }
    if (gr_hd) __syncthreads();                                                 // (the LDS is free for the last phase; hd's granules are their own flags)
    else {
    wait_stores_done();                                                         // every wave: its rows of hd are where the others will read them
    __syncthreads();                                                            // (and the LDS is free for the last phase)
    raise_line<TP>(p, p.flag_hd, p.tp.off_hd, (TP ? (unsigned)p.tp.rank * gridDim.x : 0u) + blockIdx.x, target);
    }
    stamp(8);
    if ((int)blockIdx.x < p.grid2) {
    // ---- FFN2: k_gemv<QUANT, RESIDUAL> behind the hd flag round
    GemvCtx<QT, EPI_RESIDUAL, true> g2;
    if constexpr ((R5 & 2) != 0) {
        // arrival order (GemvCtx::run_ao): no workgroup-wide poll, no prologue -- a wave quantizes the column block of each of its steps when the FFN13 workgroups that
        // produced THAT block have raised their lines.  W2's whole share is resident (two register sets per wave + nst2_ao stash slots), requested as ao_2 says.
        g2.init(a2, blockIdx.x, p.grid2, lds, 0, p.st_base, (unsigned)p.nst2_ao);
        if (gr) { g2.gsrc = p.xg_b + p.gres_off; g2.gron = xflag; g2.gtag = target; }   // (a launch's last layer without tail: the classifier's launch reads the plain vector)
        const typename GemvCtx<QT, EPI_RESIDUAL, true>::AoSrc src{p.flag_hd, (unsigned)a13.rows_per_pass, (unsigned)p.grid13, target, p.err};
        g2.issue(kAblate ? a2.ablate : 0, p.ao_2 == 1 ? 0 : 1);
        if (p.ao_2 != 2) g2.stash_issue(lds);
        g2.template run_ao<PRO_QUANT>(a2, lds, src, [&]() { g2.issue_missing(kAblate ? a2.ablate : 0); if (p.ao_2 == 2) g2.stash_issue(lds); },
                                      [&](int k) { if (k == 3) stamp(9); else if (k == 4) stamp(10); });
        stamp(11);
    } else {
    g2.init(a2, blockIdx.x, p.grid2, lds, 0, p.st_base, (unsigned)p.nst2);
    if (gr) { g2.gsrc = p.xg_b + p.gres_off; g2.gron = xflag; g2.gtag = target; }
    if ((int)g2.wave < p.pre2) g2.issue(kAblate ? a2.ablate : 0, 1);            // the first pre2 waves: ONE set now, the rest when hd has arrived (k_ffn: all 16)
    g2.stash_issue(lds);
    float4 xv2[XR2 > 0 ? XR2 : 1], nv2[XR2 > 0 ? XR2 : 1];
    if (gr_hd) {
        gemv_preload_granules<PRO_QUANT, (XR2 > 0 ? XR2 : 1), TP>(a2, xv2, nv2, p.xg_hd, target, p);
        wait_stores_done();                                                     // every wave: the stash slots it requested have landed
        stamp(9);
    } else {
    poll_lines_t<TP>(p.flag_hd, (TP ? p.tp.world : 1) * (int)gridDim.x, target, p);
    wait_stores_done();                                                         // every wave: the stash slots it requested have landed
    __syncthreads();
    stamp(9);
    gemv_preload<QT, PRO_QUANT, XR2, true>(a2, xv2, nv2);
    }
    gemv_prologue<QT, PRO_QUANT, XR2, true, FLM_BACK_LATE2 == 1>(a2, lds, xv2, nv2, [&](int) { g2.issue_missing(kAblate ? a2.ablate : 0); });
    stamp(10);
    g2.run(a2, lds, nostamp);
    stamp(11);
    }
    }
    if constexpr (PERSIST) {
        if (xflag) {
            if (gr) __syncthreads();                                            // (the LDS is free for the next layer; x's granules are their own flags)
            else {
            wait_stores_done();                                                 // every wave: its rows of x are where the next layer will read them
            __syncthreads();                                                    // (and the LDS is free for the next layer)
            if ((int)blockIdx.x < p.grid2) raise_line<TP>(p, p.flag_x2, p.tp.off_x2, (TP ? (unsigned)(p.tp.rank * p.grid2) : 0u) + blockIdx.x, target);
            }
        }
    }
}

template <int QT, int XR2, bool QKV = false, bool SPLIT = false>
__global__ void __launch_bounds__(kGemvBlock, 4) k_attn_ffn(const GemvArgs aq, const AttnArgs aa, const GemvArgs ao, const GemvArgs a13, const GemvArgs a2, const BackArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    layer_body<QT, XR2, QKV, SPLIT, false>(aq, aa, ao, a13, a2, p, lds, p.target, false, false, true);
}

// ------------------------------------------------------------------------------------------
// k_layers: the layers [l0, l1) of a token in ONE launch -- layer_body in a loop, the argument blocks of every layer in device memory (built once by the host: they hold
// pointers only; position and token are read through pos_ptr).  What the loop buys is the edge between two layers: instead of a kernel boundary and a launch ramp, a flag round
// in front of which the next layer's [Wq; Wk; Wv] starts to stream (preq waves' first register sets + nstq stash slots: the LDS is empty there).
// ------------------------------------------------------------------------------------------
struct LayerArgs { GemvArgs aq, ao, a13, a2; AttnArgs aa; };
// TAIL (round 5): the WHOLE token in this launch -- the embedding row is the first layer's input (read where it is: transformer.cpp:115-122), the classifier (final rmsnorm, quantize,
// rows of the output matrix: transformer.cpp:154-160) is a phase behind the last layer's x flag round like the QKV phase of a layer (its first register sets and stash slots requested in
// front of that round), and the greedy argmax (first maximum wins, sampler.cpp:36-47) + the decode state's advance close it: every classifier workgroup leaves the best of its own rows
// in a slot and raises a line, workgroup 0 waits for all of them and picks.  Nobody clears the flag lines between tokens (k_embed did): every target of the launch counts from an epoch
// base in device memory that the launch's last act moves on by L + 2 -- stale lines are always below.  fp32 embedding tables only (the host checks).
struct TailArgs {
    GemvArgs acls;              // the classifier GEMV (EPI_STORE into the logits)
    const float* emb; const int* tok_ptr; int dim, vocab;
    unsigned* epoch;            // the launch's flag values count from *epoch (>= 4096: above anything k_embed-era launches leave in a line)
    unsigned* flag_cls;         // one line per classifier workgroup
    float* slots;               // [gridc][2] { best logit, its row (as bits) } of a workgroup's rows
    int gridc;
    DecodeState* st; int* out_tokens; int out_cap;
};
template <int QT>
__device__ __forceinline__ void tail_phase(const TailArgs& T, const BackArgs& p, char* lds, const unsigned xtarget, const unsigned ctarget, const unsigned next_epoch) {
    auto nostamp = [](int) {};
    if ((int)blockIdx.x < T.gridc) {
        const GemvArgs& a = T.acls;
        GemvCtx<QT, EPI_STORE, true> gc;
        gc.init(a, blockIdx.x, T.gridc, lds, 0, p.st_base, (unsigned)p.nstq);
        if ((int)gc.wave < p.preq) gc.issue(kAblate ? a.ablate : 0, 1);
        gc.stash_issue(lds);
        float4 xq[1], nq[1];
        gemv_preload_granules<PRO_RMSNORM_QUANT, 1, false>(a, xq, nq, p.xg_a, xtarget, p);     // (the one-launch token hands x over as granules: layer_body's GRM)
        wait_stores_done();                                                     // every wave: the stash slots it requested have landed
        gemv_prologue<QT, PRO_RMSNORM_QUANT, 1, true, FLM_LAYER_LATEQ != 0>(a, lds, xq, nq, [&](int) { gc.issue_missing(kAblate ? a.ablate : 0); });
        gc.run(a, lds, nostamp);
        wait_stores_done();                                                     // every wave: its logits are in memory
        __syncthreads();
        // the best of this workgroup's rows (pass wg + it nwg = rows [pass Rm, pass Rm + Rm)): thread r of wave 0 walks its row of every pass in ascending order, strict '>'
        if (threadIdx.x < 64) {
            float best = -INFINITY; int idx = 0x7fffffff;
            for (unsigned it = 0; it < gc.np_wg; ++it) {
                const unsigned row = (gc.wg + it * gc.nwg) * gc.Rm + threadIdx.x;
                if (threadIdx.x < gc.Rm && row < (unsigned)T.vocab) { const float v = ld_agent(a.out + row); if (v > best) { best = v; idx = (int)row; } }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(best, o, kWave); const int oi = __shfl_xor(idx, o, kWave);
                if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
            }
            if (threadIdx.x == 0) { st_agent(T.slots + 2 * blockIdx.x, best); st_agent(T.slots + 2 * blockIdx.x + 1, __int_as_float(idx)); }
            wait_stores_done();
            if (threadIdx.x == 0) __hip_atomic_store(T.flag_cls + blockIdx.x * kFlagStride, ctarget, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (blockIdx.x == 0) {
        // the token's argmax over the workgroups' candidates; lower row wins ties (sample_argmax: the first maximum)
        poll_lines(T.flag_cls, T.gridc, ctarget, p.err);
        __syncthreads();
        float* bv = reinterpret_cast<float*>(lds); int* bi = reinterpret_cast<int*>(lds) + 16;
        float best = -INFINITY; int idx = 0x7fffffff;
        if ((int)threadIdx.x < T.gridc) { best = ld_agent(T.slots + 2 * threadIdx.x); idx = __float_as_int(ld_agent(T.slots + 2 * threadIdx.x + 1)); }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, kWave); const int oi = __shfl_xor(idx, o, kWave);
            if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
        }
        if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kWavesPerBlock; ++w) if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
            if (idx == 0x7fffffff) idx = 0;                                     // all -inf / NaN: the reference returns index 0
            DecodeState* st = T.st;
            if (T.out_tokens && st->step >= 0 && st->step < T.out_cap) T.out_tokens[st->step] = idx;
            st->tok = idx; st->pos += 1; st->step += 1;
            *T.epoch = next_epoch;
        }
    }
}
// TP (round 6): the launch spans the tensor-parallel ranks (BackArgs::Tp; every rank launches it on its own stream, all of them must be running for any to finish): the
// flag values count from the token's epoch base (k_embed moved it on, on every rank alike), the lines live in the ranks' exchange buffers and are never cleared.
template <int QT, int XR2, bool SPLIT, int R5 = 0, bool TAIL = false, bool TP = false, bool GRT = false>     // GRT: the rank-spanning launch on granules (BackArgs::gr)
__global__ void __launch_bounds__(kGemvBlock, 4) k_layers(const LayerArgs* __restrict__ LA, const BackArgs p, const int l0, const int l1, const TailArgs* __restrict__ TA = nullptr) {
    static_assert(!TP || (R5 == 0 && !TAIL), "the rank-spanning launch carries the all-to-all hand-offs, the classifier is a launch of its own");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // (the argument blocks through the CONSTANT address space: uniform scalar loads at the point of use, like kernel arguments -- through a generic pointer they would sit in vector registers)
    typedef const LayerArgs __attribute__((address_space(4))) CLayerArgs;
    CLayerArgs* LAc = (CLayerArgs*)(unsigned long long)LA;
    unsigned base = 0; const float* x0 = nullptr;
    if constexpr (TAIL) {
        typedef const TailArgs __attribute__((address_space(4))) CTailArgs;
        const TailArgs& T = *(const TailArgs*)((CTailArgs*)(unsigned long long)TA);
        base = *T.epoch;
        x0 = T.emb + (size_t)(*T.tok_ptr) * T.dim;
    }
    if constexpr (TP) base = *p.tp.base;
    for (int l = l0; l < l1; ++l) {
        const LayerArgs& A = *(const LayerArgs*)(LAc + l);
        layer_body<QT, XR2, true, SPLIT, true, R5, TP, (TP ? (GRT ? 2 : 0) : (TAIL ? 2 : 0))>(A.aq, A.aa, A.ao, A.a13, A.a2, p, lds, base + (unsigned)(l + 1), l > l0, TAIL || l + 1 < l1, l == (l1 - l0 > 1 ? l0 + 1 : l0),   // (trace builds: the stamps of the launch's second layer)
                                                   (TAIL && l == l0) ? x0 : nullptr);
    }
    if constexpr (TAIL) {
        typedef const TailArgs __attribute__((address_space(4))) CTailArgs;
        const TailArgs& T = *(const TailArgs*)((CTailArgs*)(unsigned long long)TA);
        tail_phase<QT>(T, p, lds, base + (unsigned)l1, base + (unsigned)l1 + 1u, base + (unsigned)l1 + 2u);
    }
}

} // namespace flm
