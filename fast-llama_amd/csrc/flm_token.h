// flm_token.h -- the persistent whole-token kernel (k_token; opt-in).
// Part of flm_kernels.h (hand-written gfx950 / CDNA4 kernels of the fast-llama per-token hot path); include that header.
#pragma once
#include "flm_math.h"
#include "flm_gemv.h"
#include "flm_attn.h"
// (bit-exactness hygiene: see flm_math.h -- no implicit FMA contraction in any of these headers)
#pragma clang fp contract(off)

namespace flm {

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
    if (ts.stored) wait_stores_done();
    if constexpr (ATTN) {
        const bool attn_wg = (int)wg < t.n_heads;                               // this workgroup runs attention heads next
        if (!attn_wg) prefetch();                                               // (a head's K/V loads must not queue behind weight loads)
        grid_barrier(t, epoch);
        if (attn_wg) {   // one head per workgroup
            const AttnArgs aa = kload(aap);
            for (int h = wg; h < t.n_heads; h += nwg) attn_head_any<true>(aa, h, lds, *aa.pos_ptr + 1, aa.q, aa.out);
            wait_stores_done();
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

} // namespace flm
