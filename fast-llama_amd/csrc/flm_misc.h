// flm_misc.h -- embedding, argmax + state advance, and the op-level kernels of the parity tests.
// Part of flm_kernels.h (hand-written gfx950 / CDNA4 kernels of the fast-llama per-token hot path); include that header.
#pragma once
#include "flm_math.h"
#include "flm_gemv.h"
#include "flm_layer.h"
// (bit-exactness hygiene: see flm_math.h -- no implicit FMA contraction in any of these headers)
#pragma clang fp contract(off)

namespace flm {

// ------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------
// x1 = embedding[token] (copy or dequantize; transformer.cpp:115-122)
constexpr unsigned kEpochStride = 1024;      // the token's epoch base advances by this: exchanges (4 per layer + 2) per token must stay below it
inline __global__ void k_embed(float* x, const void* emb, const float* emb_s, int emb_qt, int dim, const int* tok_ptr, unsigned* bar, unsigned* eng_base = nullptr, unsigned long long* ffn_counter = nullptr) {
    const int tok = *tok_ptr;
    // a new token: the epoch base of the tensor-parallel exchanges' flag values moves on; wrap long before a value could become 0
    if (eng_base && blockIdx.x == 0 && threadIdx.x == 0) { const unsigned b = *eng_base; *eng_base = b >= 0xFFF00000u ? 0u : b + kEpochStride; }
    // k_ffn<TP> finds a rank's last workgroup with (count + 1) % grid: a new token starts the count at 0, whatever grid earlier launches had (or a launch that gave up left behind)
    if (ffn_counter && blockIdx.x == 0 && threadIdx.x == 0) *ffn_counter = 0ull;
    if (bar && blockIdx.x == 0) {   // the flag lines (kFlagLines = 1536, 64 B apart) the workgroups of the token's fused launches wait on
#pragma unroll
        for (int k = 0; k < 6; ++k) bar[(threadIdx.x + k * blockDim.x) * 16] = 0;
    }
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
inline __global__ void __launch_bounds__(1024) k_argmax_advance(const float* logits, int n, DecodeState* st, int* out_tokens, int advance, int out_cap) {
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
        if (out_tokens && st->step >= 0 && st->step < out_cap) out_tokens[st->step] = idx;      // (out_cap: the buffer's size; a caller that forgot to reset `step` must not write past it)
        if (advance) { st->tok = idx; st->pos += 1; }
        st->step += 1;
    }
}
// prompt feeding: pos <- pos+1, tok <- prompt[++step]
inline __global__ void k_advance_prompt(DecodeState* st, const int* prompt) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { st->step += 1; st->pos += 1; st->tok = prompt[st->step]; }
}
inline __global__ void k_set_step(DecodeState* st, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) st->step = v; }
// ------------------------------------------------------------------------------------------
// Tensor parallel, one-shot peer-to-peer exchange (instead of an RCCL all-gather per activation vector: ~3 us against ~12).
// The producing kernel has stored its slice of the vector into EVERY rank's buffer (GemvArgs::out_peer).  This one-workgroup
// kernel runs right behind it in the stream: the kernel boundary has completed those stores, so thread r tells rank r
// "my slice of exchange number e is in your memory" by writing e into its line of rank r's flag array (system scope), and
// then waits until all ranks' lines in the LOCAL flag array have reached e.  e counts this kind's exchanges in device memory, so
// a captured graph replays correctly.  flags: [kind][rank] lines of 64 bytes.
// ------------------------------------------------------------------------------------------
struct XchgArgs { unsigned* local_flags; unsigned* peer_flags[8]; unsigned* epoch; int* err; int rank, world, kind; };
inline __global__ void __launch_bounds__(64) k_xchg(const XchgArgs x) {
    const int r = threadIdx.x;
    const unsigned e = *x.epoch + 1;
    // release: the slices this rank's producing kernel stored into the peers' buffers (system-scope stores, completed by the kernel boundary in front of
    // this launch) are ordered before the flag for ANY observer -- on one device that cannot fail, across xGMI links it is what the flag promises
    __atomic_thread_fence(__ATOMIC_RELEASE);
    if (r < x.world) __hip_atomic_store(x.peer_flags[r] + (x.kind * 8 + x.rank) * 16, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (true) {
        const unsigned f = r < x.world ? __hip_atomic_load(x.local_flags + (x.kind * 8 + r) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : e;
        if (__all((int)(f - e) >= 0)) break;
        // ranks start seconds apart (graph capture, module loading): be patient; but once ANY rank has given up, everybody leaves at once
        const bool aborted = __hip_atomic_load(x.local_flags + kXchgAbortLine * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
        if (aborted || __builtin_amdgcn_s_memrealtime() - t0 > 2000000000ull) {                                   // 20 s of the 100 MHz clock
            __hip_atomic_store(x.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (r < x.world) __hip_atomic_store(x.peer_flags[r] + kXchgAbortLine * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    // acquire: nothing this rank reads after the wait (the consumers' system-coherent loads in the next launch) is served from before the peers' flags
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (r == 0) *x.epoch = e;
}

// ---- op-level test kernels: thin launchers over the same __device__ functions ----
// square_sum both ways: out[0] the speculative wave evaluation (sq_chain_spec), out[1] the plain sequential chains (sq_chain);
// out[2..5] the 4 strided lanes from sq_chain_spec, out[6..9] its round counts (-1: it fell back to the plain chain)
inline __global__ void __launch_bounds__(256) k_op_square_sum(float* out, const float* x, int n) {
    extern __shared__ float sm[];
    const int n4 = n / 4, ns = n4 + 8;
    const int bs = chain_bshift(n), B = 1 << bs, LS = B + 4, CS = 64 * LS;
    float* seq = sm + 4 * CS;                                     // [4][n4 + 8] for the plain chains
    for (int i = threadIdx.x; i < 4 * CS; i += blockDim.x) sm[i] = 0.f;
    for (int i = threadIdx.x; i < 4 * ns; i += blockDim.x) seq[i] = 0.f;
    __syncthreads();
    for (int e = threadIdx.x; e < n4 * 4; e += blockDim.x) {
        const int c = e & 3, k = e >> 2;
        sm[c * CS + (k >> bs) * LS + (k & (B - 1))] = x[e];
        seq[c * ns + k] = x[e];
    }
    __shared__ float red[8];
    __syncthreads();
    int its = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float l = sq_chain_spec(sm + (threadIdx.x >> 6) * CS, bs, &its);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    {   // the register-fed form (k_attn_ffn's prologue: elements in registers, the head through v_readlane) must give the same bits
        const float* cp = sm + (threadIdx.x >> 6) * CS;
        float lr = l;
        if (bs == 4) lr = sq_chain_spec_t<4>(cp, bs, nullptr); else if (bs == 3) lr = sq_chain_spec_t<2>(cp, bs, nullptr); else if (bs == 2) lr = sq_chain_spec_t<1>(cp, bs, nullptr);
        if (__float_as_uint(lr) != __float_as_uint(l)) l = __uint_as_float(0x7fc00000u);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = l; out[6 + (threadIdx.x >> 6)] = (float)its; }
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x < 4) red[4 + threadIdx.x] = sq_chain(seq + threadIdx.x * ns, n4);
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[10] = (float)(t1 - t0); out[11] = (float)(t3 - t2); }      // shader-clock ticks: speculative / plain chain
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, red[0]), red[1]), red[2]), red[3]);
        out[1] = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, red[4]), red[5]), red[6]), red[7]);
        out[2] = red[0]; out[3] = red[1]; out[4] = red[2]; out[5] = red[3];
    }
}
inline __global__ void k_op_swiglu(float* xo, const float* xr, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) xo[i] = swiglu_elem(xo[i], xr[i]);
}
// elementary functions as the kernels evaluate them: fn 0 expf_ref(x), 1 sqrtf(x), 2 x / y, 3 rms_scale(x, n = (int)y), 4 / 5 the quantizer's element step (below)
inline __global__ void k_op_math(int fn, float* x, const float* y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (fn == 4 || fn == 5) {      // the quantizer's element step, x = value, y = the group's scale: 4 through quant_elems4 (shared reciprocal), 5 through quant_elem (IEEE division)
            int q[4]; const float4 v4 = make_float4(v, -v, v, v);
            if (fn == 4) quant_elems4(v4, y[i], q); else { q[0] = quant_elem(v, y[i]); q[1] = quant_elem(-v, y[i]); }
            x[i] = (float)(q[0] - 1024 * q[1]);                      // (both signs in one number: |q| <= 127 F fits)
            continue;
        }
        x[i] = fn == 0 ? expf_ref(v) : fn == 1 ? __builtin_sqrtf(v) : fn == 2 ? __fdiv_rn(v, y[i]) : rms_scale(v, (int)y[i]);
    }
}
inline __global__ void k_op_rope(float* o, const float* x, int n_dims, const float* c, const float* s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n_dims) rope_pair(x[2 * i], x[2 * i + 1], c[i], s[i], o[2 * i], o[2 * i + 1]);
}
// softmax_sisd over n entries, one workgroup (same statements as k_attn_decode's softmax)
inline __global__ void __launch_bounds__(kBlock) k_op_softmax(float* x, int n) {
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
inline __global__ void k_op_kv_append(float* q, const float* k, const float* v, float* kc, float* vc, const float* c, const float* s,
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

// ------------------------------------------------------------------------------------------
// Hand-off litmus (tests/test_gpu_ops.py::test_handoff_litmus): the EXACT publish / poll / read sequence of layer_body's flag rounds, run `rounds` times by one
// workgroup per CU, every read checked -- so that a compiler, firmware or memory-model change that reorders it fails a named test, not a token id.
//   publish : write-through agent-scope stores of the payload (st_agent) -> wait_stores_done (s_waitcnt vmcnt(0)) -> workgroup barrier -> thread 0: relaxed
//             agent-scope store of the round number into the workgroup's own 64-byte flag line
//   consume : lane i of the first waves polls line i with relaxed agent-scope loads (poll_lines) -> workgroup barrier -> coherent (sc0 sc1) buffer loads
// Payload: 64 dwords per (round parity, workgroup), values a hash of (producer, round, index): data-dependent, never repeating.  A workgroup writes round r + 2 into the
// buffer of round r only after it has seen every line at r + 1, i.e. after every consumer has finished reading round r: two buffers suffice.
// err[0] = wrong values read, err[1] = a wait timed out.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned litmus_value(unsigned prod, unsigned round, unsigned i) {
    unsigned h = prod * 0x9E3779B1u ^ round * 0x85EBCA77u ^ i * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h | 1u;
}
inline __global__ void __launch_bounds__(256) k_handoff_litmus(float* payload, unsigned* flags, int rounds, int* err, int* timeout) {
    const unsigned G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(payload, 0, (int)(2u * G * 64u * 4u), 0x00020000);
    unsigned bad = 0;
    for (unsigned r = 1; r <= (unsigned)rounds; ++r) {
        float* mine = payload + ((size_t)(r & 1u) * G + b) * 64;
        if (tid < 64) st_agent(mine + tid, __uint_as_float(litmus_value(b, r, tid)));
        wait_stores_done();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + b * kFlagStride, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        poll_lines(flags, (int)G, r, timeout);
        __syncthreads();
        if (__hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        // thread t reads 16 bytes of producer (b + 1 + t / 16 + 16 k) % G, k = 0 .. G / 16: every workgroup reads every producer's whole payload
        for (unsigned k = 0; k * 16 < G; ++k) {
            const unsigned prod = (b + 1 + tid / 16 + 16 * k) % G, i0 = (tid & 15) * 4;
            if (tid / 16 + 16 * k >= G) continue;
            typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
            const v4u32 v = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rp, (int)((((r & 1u) * G + prod) * 64 + i0) * 4), 0, kAuxCoherent));
            bad += (v.x != litmus_value(prod, r, i0)) + (v.y != litmus_value(prod, r, i0 + 1)) + (v.z != litmus_value(prod, r, i0 + 2)) + (v.w != litmus_value(prod, r, i0 + 3));
        }
    }
    if (bad) atomicAdd(err, (int)bad);
}

} // namespace flm
