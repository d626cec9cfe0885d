// flm_gpu.hip -- device context, weight upload, per-token forward and the C ABI (include/flm_gpu.h).
//
// Host-side orchestration of the kernels in flm_kernels.h.  What the reference does with 161
// fork-joins over pinned CPU threads per token (src/transformer/transformer.cpp:105-161)
// is here one stream of 5*L+3 kernel launches whose position/token operands live in device memory, so
// the whole token can be replayed from a hipGraph with no host round trip.
#include "flm_host.h"

namespace fh {

thread_local std::string g_last_error;

int fail(flm_ctx* c, int code, const char* msg) { if (c) c->err = msg; g_last_error = msg; return code; }

int prepare_all(flm_ctx* c);
int esz_of(int qt) { return qt == FLM_QT_INT8 ? 1 : qt == FLM_QT_INT16 ? 2 : 4; }

// balanced contiguous split (split_rows, transformer.cpp:264-287)
void split_even(int total, int parts, int idx, int* begin, int* count) {
    const int itv = total / parts, rem = total % parts;
    if (idx < rem) { *begin = (itv + 1) * idx; *count = itv + 1; }
    else { *begin = (itv + 1) * rem + itv * (idx - rem); *count = itv; }
}

// quantize a flat fp32 array on the device with the fused path's quantizer (A13, load time):
// one 16-lane group per 64-element group.
template <int QT>
__global__ void k_quantize_flat(void* q, float* s, const float* x, size_t n) {
    using T = QTraits<QT>;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    const size_t nr = (n + stride - 1) / stride;
    for (size_t it = 0; it < nr; ++it) {
        const size_t e = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4 + it * stride;
        const bool act = e < n;
        float4 v = act ? *reinterpret_cast<const float4*>(x + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float mx = row16_max(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        const float sc = __fdiv_rn(mx, T::kF);
        if (act) {
            const int q0 = quant_elem(v.x, sc), q1 = quant_elem(v.y, sc), q2 = quant_elem(v.z, sc), q3 = quant_elem(v.w, sc);
            if constexpr (QT == QT_INT8) {
                reinterpret_cast<uint32_t*>(q)[e / 4] = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
            } else {
                uint2 pk; pk.x = (uint32_t)(q0 & 0xffff) | ((uint32_t)(q1 & 0xffff) << 16); pk.y = (uint32_t)(q2 & 0xffff) | ((uint32_t)(q3 & 0xffff) << 16);
                reinterpret_cast<uint2*>(q)[e / 4] = pk;
            }
            if ((threadIdx.x & 15) == 0) s[e / kGroup] = sc;
        }
    }
}
int quantize_flat(flm_ctx* c, hipStream_t st, int qt, void* q, float* s, const float* x, size_t n) {
    if (n % kGroup) return fail(c, FLM_ERR_INVALID, "quantize: n must be a multiple of 64");
    size_t blocks = (n / 4 + kBlock - 1) / kBlock; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    if (qt == FLM_QT_INT8) hipLaunchKernelGGL(k_quantize_flat<QT_INT8>, dim3((unsigned)blocks), dim3(kBlock), 0, st, q, s, x, n);
    else if (qt == FLM_QT_INT16) hipLaunchKernelGGL(k_quantize_flat<QT_INT16>, dim3((unsigned)blocks), dim3(kBlock), 0, st, q, s, x, n);
    else return fail(c, FLM_ERR_UNSUPPORTED, "quantize: type");
    HIPC(c, hipGetLastError());
    return FLM_OK;
}

// RoPE table with the reference's fp32 recurrence (rope_v2, src/blas/tf_operators.cpp:362-396):
// theta_0 = pos, theta_{i+1} = theta_i * powf(10000, -2/hs); cosf/sinf from the host libm, the same
// library the reference calls, so the table is bit-identical to what rope_v2 computes per call.
void build_rope_table(int hs, int max_seq, std::vector<float>& cs, std::vector<float>& sn) {
    cs.resize((size_t)max_seq * (hs / 2)); sn.resize(cs.size());
    const float theta_scale = powf(10000.0f, -2.0f / hs);
    for (int p = 0; p < max_seq; ++p) {
        float theta = (float)p;
        for (int i = 0; i < hs / 2; ++i) {
            cs[(size_t)p * (hs / 2) + i] = cosf(theta);
            sn[(size_t)p * (hs / 2) + i] = sinf(theta);
            theta *= theta_scale;
        }
    }
}

int alloc_qmat(flm_ctx* c, QMat& m, int rows, int cols, int qt, bool with_st = false) {
    m.rows = rows; m.cols = cols;
    HIPC(c, hipMalloc(&m.q, (size_t)rows * cols * esz_of(qt)));
    HIPC(c, hipMalloc((void**)&m.s, (size_t)rows * (cols / kGroup) * sizeof(float)));
    if (with_st) HIPC(c, hipMalloc((void**)&m.st, (size_t)rows * (cols / kGroup) * sizeof(float)));
    return FLM_OK;
}

// copy a (row range x column range) window of a host matrix into a device QMat at dst_row0.
// fp32 sources are staged and quantized on the device (A13).
int upload_window(flm_ctx* c, QMat& m, int dst_row0, int src_qt, const void* values, const float* scales,
                  int src_cols, int row0, int nrows, int col0, int ncols) {
    const int qt = c->d.quant_type, gs = kGroup;
    if (ncols != m.cols) return fail(c, FLM_ERR_INVALID, "upload: column window does not match the device matrix");
    if (src_qt == FLM_QT_NONE) {
        DevMem stage_mem;
        HIPC(c, hipMalloc(&stage_mem.p, (size_t)nrows * ncols * sizeof(float)));
        float* stage = (float*)stage_mem.p;
        HIPC(c, hipMemcpy2DAsync(stage, (size_t)ncols * 4, (const float*)values + (size_t)row0 * src_cols + col0, (size_t)src_cols * 4,
                                 (size_t)ncols * 4, nrows, hipMemcpyHostToDevice, c->stream));
        int r = quantize_flat(c, c->stream, qt, (char*)m.q + (size_t)dst_row0 * ncols * c->esz, m.s + (size_t)dst_row0 * (ncols / gs), stage, (size_t)nrows * ncols);
        if (r) return r;
        HIPC(c, hipStreamSynchronize(c->stream));
        return FLM_OK;
    }
    if (src_qt != qt) return fail(c, FLM_ERR_INVALID, "upload: tensor quant type differs from the model's");
    const int e = c->esz;
    HIPC(c, hipMemcpy2DAsync((char*)m.q + (size_t)dst_row0 * ncols * e, (size_t)ncols * e,
                             (const char*)values + ((size_t)row0 * src_cols + col0) * e, (size_t)src_cols * e,
                             (size_t)ncols * e, nrows, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpy2DAsync(m.s + (size_t)dst_row0 * (ncols / gs), (size_t)(ncols / gs) * 4,
                             scales + (size_t)row0 * (src_cols / gs) + col0 / gs, (size_t)(src_cols / gs) * 4,
                             (size_t)(ncols / gs) * 4, nrows, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return FLM_OK;
}

bool model_complete(const flm_ctx* c) {
    if (!c->got_emb || !c->got_out_norm || !c->got_cls) return false;
    const unsigned need = (1u << 0) | (1u << 1) | (1u << 2) | (1u << 3) | (1u << 4) | (1u << 5) | (1u << 6) | (1u << 7) | (1u << 8);
    for (auto& l : c->layers) if ((l.got & need) != need) return false;
    return true;
}

// k_attn_o reports a cross-workgroup wait that never completed (a head workgroup was not resident: another process held
// CUs) through *xwg_err.  The call's results are then invalid: the fused launch is switched off for the rest of this
// context's life and FLM_RETRY tells the caller (inside this library) to run the call again on one kernel per phase.
int xwg_check(flm_ctx* c) {
    if (!c->fuse_attn_o && !c->fuse_ffn && !c->fuse_back && c->attn_split == 0 && !c->p2p) return FLM_OK;
    // (on the context's own stream: a copy on the legacy stream synchronises with every blocking stream of the process -- and fails
    //  outright while another context's thread is capturing its token graph; seen once in ~10 runs of the threaded tensor-parallel tests)
    int e = 0;
    if (c->err_word_fresh) { e = c->err_word; c->err_word_fresh = false; }       // (d2h brought it along with the call's results)
    else {
        HIPC(c, hipMemcpyAsync(&e, c->xwg_err, 4, hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
    }
    if (!e) return FLM_OK;
    HIPC(c, hipMemsetAsync(c->xwg_err, 0, 4, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (e == 2) return fail(c, FLM_ERR_COMM, "tensor parallel: a peer rank did not deliver its slice (20 s), or another rank gave up; the context group cannot be used any more");
    if (c->world > 1) {
        // A cross-workgroup wait inside this rank timed out (a split head's scores).  The other ranks cannot re-run the call with it -- they
        // have gone on with this rank's bad slices -- so this is a GROUP error: tell them (abort line) and say so; no silent retry.
        if (c->p2p) {
            for (int r = 0; r < c->world; ++r) if (c->peer[r]) { unsigned one = 1; (void)hipMemcpyAsync(c->peer[r] + c->x_flags_off + kXchgAbortLine * 64, &one, 4, hipMemcpyHostToDevice, c->stream); }
            (void)hipStreamSynchronize(c->stream);
        }
        c->attn_split = 0;
        return fail(c, FLM_ERR_COMM, "tensor parallel: a cross-workgroup wait on this rank timed out; the group's results are invalid and the context group cannot be used any more");
    }
    if (!c->fb_active) { c->fb_saved[0] = c->fuse_attn_o; c->fb_saved[1] = c->fuse_ffn; c->fb_saved[2] = c->fuse_qkv; c->fb_saved[3] = c->fuse_back; c->fb_saved[4] = c->fuse_token; c->fb_saved[5] = c->attn_split; }
    c->fuse_attn_o = 0; c->fuse_ffn = 0; c->fuse_qkv = 0; c->fuse_back = 0; c->fuse_token = 0; c->attn_split = 0; c->fell_back += 1; c->fb_active = true; c->fb_tokens = 0;
    for (auto& g : c->graphs) hipGraphExecDestroy(g.second);
    c->graphs.clear();
    return FLM_RETRY;
}


// Census: the fused launches (k_attn_o, k_ffn, k_qkv_attn_o, k_attn_ffn, split heads) wait for each other's flags, so every workgroup of a
// cu_count-wide launch of 1024-thread workgroups with most of the CU's LDS must be RESIDENT at once.  The occupancy API cannot see a masked or
// partitioned device (MI355X_MICROARCH.md: verify with a census kernel): every workgroup checks in and waits (bounded) until all have.
__global__ void __launch_bounds__(1024) k_census(unsigned* counter, unsigned n, int* ok) {
    extern __shared__ char census_lds[];
    if (threadIdx.x == 0) {
        census_lds[0] = 1;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool all = false;
        while (!all && __builtin_amdgcn_s_memrealtime() - t0 < 200000ull) {                       // 2 ms of the 100 MHz clock
            all = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n;
            if (!all) __builtin_amdgcn_s_sleep(8);
        }
        if (!all) __hip_atomic_store(ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the census of a cu_count-wide launch on the context's stream: 1 = every workgroup resident at once (up to 3 attempts: another process busy on the device for the census' 2 ms
// must not switch the fused launches off), 0 = not, -1 = a HIP error
static int run_census(flm_ctx* c) {
    unsigned* cnt = (unsigned*)((char*)c->xwg_err + 32); int* okp = c->xwg_err + 4;
    for (int attempt = 0; attempt < 3; ++attempt) {
        int one = 1, ok = 0;
        if (hipMemsetAsync(cnt, 0, 4, c->stream) != hipSuccess || hipMemcpyAsync(okp, &one, 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) { (void)hipGetLastError(); return -1; }
        hipLaunchKernelGGL(k_census, dim3(c->cu_count), dim3(1024), 150 * 1024, c->stream, cnt, (unsigned)c->cu_count, okp);
        if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&ok, okp, 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess) { if (ok) return 1; }
        else { (void)hipGetLastError(); return -1; }
    }
    return 0;
}
// A context that fell back to one kernel per phase (xwg_check: a cross-workgroup wait gave up -- a co-tenant held CUs for 20 ms) does not stay there: after kFallbackProbation tokens
// on the per-phase path the census runs again, and if every workgroup is resident the launch structure the context had comes back ("fallback" counts the episodes, "fallback_active"
// says where the context is).  Called by the token entry points after a completed call.
constexpr int kFallbackProbation = 64;
int maybe_recover(flm_ctx* c, int tokens) {
    if (!c->fb_active) return FLM_OK;
    c->fb_tokens += tokens;
    if (c->fb_tokens < kFallbackProbation) return FLM_OK;
    c->fb_tokens = 0;
    if (run_census(c) != 1) return FLM_OK;                               // still crowded: another probation period
    c->fuse_attn_o = c->fb_saved[0]; c->fuse_ffn = c->fb_saved[1]; c->fuse_qkv = c->fb_saved[2]; c->fuse_back = c->fb_saved[3]; c->fuse_token = c->fb_saved[4]; c->attn_split = c->fb_saved[5];
    c->fb_active = false;
    for (auto& g : c->graphs) hipGraphExecDestroy(g.second);
    c->graphs.clear();
    c->la_valid[0] = c->la_valid[1] = false;
    return prepare_all(c);
}

// run one token, through a cached hipGraph when enabled.  T = positions the token's attention covers (known to the host:
// it picks how many workgroups a head is spread over; the graphs are keyed by it)
// n greedy tokens whose attention spreads a head over the same number of workgroups, as ONE cached graph of n token sequences (a chunk): between two graph launches the
// device idles ~10 us, between two nodes of a graph ~1.5 -- with the token one launch long that gap is the largest item left outside it.  Chunks of 16, 8, 4, 2 tokens, then single ones.
constexpr int kChunk = 16;
static bool graphs_in_use(const flm_ctx* c) { return c->use_graph && !c->timing && !((c->world > 1 || (c->comm && c->force_tp)) && !c->p2p); }   // (RCCL collectives stay eager)
static bool chunks_in_use(const flm_ctx* c) { return c->use_graph && !c->timing && c->world == 1 && !(c->comm && c->force_tp) && c->graph_chunks; }
// the cached graph of `n` token sequences (n >= 2: greedy tokens, a chunk; n == 1: one token by (classifier, advance)) for G workgroups per head: captured and instantiated on first use --
// flm_prepare (and the end of the upload) asks for every graph the entry points replay, so that this happens THERE and not inside a forward
static int token_graph(flm_ctx* c, bool with_cls, int advance, int G, int n, hipGraphExec_t* out) {
    const int key = (with_cls ? 4 : 0) + advance + 8 * G + 4096 * (n >= 2 ? n : 0);
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        HIPC(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        int r = FLM_OK;
        for (int i = 0; i < n && !r; ++i) r = enqueue_token(c, c->stream, with_cls, advance, G);
        hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (r) { if (g) hipGraphDestroy(g); return r; }
        HIPC(c, e);
        HIPC(c, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        HIPC(c, hipGraphDestroy(g));
        it = c->graphs.emplace(key, ge).first;
    }
    *out = it->second;
    return FLM_OK;
}
// k_layers' argument blocks: device memory, never built inside a capture -- and both head splits at once: the copy synchronizes the stream, which must not happen
// at the position where a decode loop crosses from one split to the other, inside somebody's timed region
static int prepare_layers(flm_ctx* c, int G) {
    int r = layers_prepare(c, G); if (!r) r = layers_prepare(c, attn_parts(c, 1)); if (!r) r = layers_prepare(c, attn_parts(c, c->d.max_seq_len));
    return r;
}
static int run_greedy_chunk(flm_ctx* c, int T, int n) {
    const int G = attn_parts(c, T);
    int r = prepare_layers(c, G); if (r) return r;
    hipGraphExec_t ge = nullptr;
    r = token_graph(c, true, 1, G, n, &ge); if (r) return r;
    HIPC(c, hipGraphLaunch(ge, c->stream));
    return FLM_OK;
}
// greedy tokens at positions pos .. pos + n - 1 (the attention of token i covers pos + i + 1 positions)
int run_greedy_tokens(flm_ctx* c, int pos, int n) {
    const bool chunks = chunks_in_use(c);
    int i = 0;
    while (i < n) {
        const int T = pos + i + 1, G = attn_parts(c, T);
        int run = 1;                                                  // tokens from here on with the same head split
        if (chunks) while (run < kChunk && i + run < n && attn_parts(c, T + run) == G) ++run;
        int m = 1; while (2 * m <= run) m *= 2;                       // the largest power of two of them: graphs of 16, 8, 4, 2 tokens (a handful of cached graphs), then single ones
        int r;
        if (m >= 2) r = run_greedy_chunk(c, T, m); else r = run_token(c, true, 1, T);
        if (r) return r;
        i += m;
    }
    return FLM_OK;
}
int run_token(flm_ctx* c, bool with_cls, int advance, int T) {
    const int G = attn_parts(c, T);
    int r = prepare_layers(c, G); if (r) return r;
    if (!graphs_in_use(c)) return enqueue_token(c, c->stream, with_cls, advance, G);
    hipGraphExec_t ge = nullptr;
    r = token_graph(c, with_cls, advance, G, 1, &ge); if (r) return r;
    HIPC(c, hipGraphLaunch(ge, c->stream));
    return FLM_OK;
}
// Everything the token entry points use beyond the buffers of flm_ctx_create: k_layers' argument blocks (both head splits) and every graph flm_forward* / flm_decode_* replay --
// single tokens by (classifier, advance) and the greedy chunks of 2 .. 16 tokens, for one workgroup per head and for split heads.  Captures and instantiates, launches nothing
// (a tensor-parallel rank must not wait for peers here).  Called when the last tensor of a model arrives, at the end of flm_p2p_import, and by flm_prepare.
int prepare_all(flm_ctx* c) {
    if (!model_complete(c)) return FLM_OK;
    if ((c->world > 1 || (c->comm && c->force_tp)) && !c->p2p && !c->comm) return FLM_OK;      // (a tensor-parallel rank that has not met its peers yet: flm_p2p_import prepares)
    const int G1 = attn_parts(c, 1), G2 = attn_parts(c, c->d.max_seq_len);
    int r = prepare_layers(c, G1); if (r) return r;
    if (!graphs_in_use(c)) return FLM_OK;
    const int Gs[2] = {G1, G2};
    for (int i = 0; i < (G2 != G1 ? 2 : 1); ++i) {
        const int G = Gs[i];
        hipGraphExec_t ge = nullptr;
        if ((r = token_graph(c, true, 0, G, 1, &ge)) || (r = token_graph(c, true, 1, G, 1, &ge)) || (r = token_graph(c, false, 2, G, 1, &ge))) return r;
        if (chunks_in_use(c)) for (int n = 2; n <= kChunk; n *= 2) if ((r = token_graph(c, true, 1, G, n, &ge))) return r;
    }
    return FLM_OK;
}

// Everything a forward needs is allocated at flm_ctx_create ("zero allocations during inference", reference README and
// transformer.cpp:110-130): the prompt / output id buffers and the batched-prefill activations are sized by max_seq_len.
int alloc_run_bufs(flm_ctx* c) {
    const auto& d = c->d;
    c->prompt_cap = d.max_seq_len; c->out_cap = d.max_seq_len;
    HIPC(c, hipMalloc((void**)&c->prompt_dev, sizeof(int) * c->prompt_cap));
    HIPC(c, hipMalloc((void**)&c->out_tokens_dev, sizeof(int) * c->out_cap));
    c->bounce_bytes = (size_t)d.vocab_size * 4; if (c->bounce_bytes < sizeof(int) * (size_t)d.max_seq_len) c->bounce_bytes = sizeof(int) * (size_t)d.max_seq_len;
    HIPC(c, hipHostMalloc((void**)&c->bounce, c->bounce_bytes + 64, hipHostMallocDefault));          // (+ a line for the error word that rides along: d2h)
    // (tensor parallel: the full-width activations are regions of the exchange buffer, the rest is this rank's shard)
    const size_t cap = d.max_seq_len < 64 ? 64 : (size_t)d.max_seq_len, nmax = d.hidden_dim > d.dim ? d.hidden_dim : d.dim;
    if (!c->pf_in_xbuf) {
        HIPC(c, hipMalloc((void**)&c->pf_x, cap * d.dim * 4));
        HIPC(c, hipMalloc((void**)&c->pf_att, cap * d.dim * 4));
        HIPC(c, hipMalloc((void**)&c->pf_hd, cap * d.hidden_dim * 4));
    }
    HIPC(c, hipMalloc((void**)&c->pf_qkv, cap * 3 * c->dim_local * 4));
    HIPC(c, hipMalloc((void**)&c->pf_q, cap * c->dim_local * 4));
    HIPC(c, hipMalloc((void**)&c->pf_gu, cap * 2 * c->hidden_local * 4));
    HIPC(c, hipMalloc((void**)&c->pf_xs, 2 * cap * (nmax / kGroup) * 4 + 64));       // row-major [tokens][groups], then group-major [groups][tokens] (+ slack: the GEMM tiles read token pairs)
    c->pf_xst = c->pf_xs + cap * (nmax / kGroup);
    HIPC(c, hipMalloc(&c->pf_xq, cap * nmax * c->esz));
    if (c->hs % 32 == 0 && c->hs <= 128 && hipMalloc((void**)&c->pf_scores, (size_t)c->heads_local * cap * d.max_seq_len * 4) != hipSuccess) {
        c->pf_scores = nullptr; (void)hipGetLastError();        // (quadratic in max_seq_len: without it prompts take the kernels that compute their own scores)
    }
    c->pf_cap = (int)cap;
    return FLM_OK;
}

// decode state <- {pos, tok, step}: by value through a one-thread kernel (an async copy from a host stack frame would be
// read after the frame is gone)
__global__ void k_set_state(DecodeState* st, int pos, int tok, int step) { if (threadIdx.x == 0 && blockIdx.x == 0) { st->pos = pos; st->tok = tok; st->step = step; st->pad = 0; } }
int set_state(flm_ctx* c, int pos, int tok, int step) {
    hipLaunchKernelGGL(k_set_state, dim3(1), dim3(64), 0, c->stream, c->state, pos, tok, step);
    HIPC(c, hipGetLastError());
    return FLM_OK;
}

int check_ready(flm_ctx* c, int n, int pos) {
    if (!c) return FLM_ERR_INVALID;
    c->err_word_fresh = false;                                                  // (an error word a previous call's read-back brought along says nothing about this call)
    if (!model_complete(c)) return fail(c, FLM_ERR_STATE, "forward before all tensors were uploaded");
    if (n < 1 || pos < 0 || pos + n > c->d.max_seq_len) return fail(c, FLM_ERR_INVALID, "tokens/pos outside [0, max_seq_len]");
    HIPC(c, hipSetDevice(c->device));
    return FLM_OK;
}

bool tp_prefill_capable(const flm_ctx* c) {
    return c->pf_in_xbuf && c->use_mfma && c->use_qk_mfma && c->use_pv_mfma && c->use_prefill_mq &&
           c->pf_scores && c->hs <= 128 && c->hs % 2 == 0 && c->dim_local % 32 == 0;
}
// device -> caller's buffer / caller's buffer -> device through the context's page-locked bounce buffer (flm_host.h: bounce), in pieces of its size; d2h synchronises.
// The bytes are moved by a KERNEL on the context's stream (the bounce buffer is mapped into the device's address space), not by hipMemcpyAsync: a copy engine's queue is
// shared by every context of the process and served in order, and a copy that waits for its stream's kernels blocks it -- under tensor parallelism (rank A's copy of token t
// in front of rank B's copy of token t - 1, A's kernels waiting for B's next slice) that is a deadlock; the engines' queues are also created lazily (device memory taken
// inside a forward: tools/alloc_diag.py).
// (word_src -> word_dst: one more word from elsewhere -- the cross-workgroup error flag rides along with a call's results, so that xwg_check needs no copy of its own)
__global__ void k_copy_words(const unsigned* __restrict__ src, unsigned* __restrict__ dst, unsigned n, const unsigned* __restrict__ word_src, unsigned* __restrict__ word_dst) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
    if (word_src && blockIdx.x == 0 && threadIdx.x == 0) *word_dst = *word_src;
}
static int copy_words(flm_ctx* c, const void* src, void* dst, size_t bytes, bool with_err = false) {
    const unsigned n = (unsigned)(bytes / 4);                                       // (ids, logits: whole words)
    const unsigned blocks = n < 256 * 64 ? (n + 255) / 256 : 64;
    hipLaunchKernelGGL(k_copy_words, dim3(blocks ? blocks : 1), dim3(256), 0, c->stream, (const unsigned*)src, (unsigned*)dst, n,
                       with_err ? (const unsigned*)c->xwg_err : (const unsigned*)nullptr, (unsigned*)(c->bounce + c->bounce_bytes));
    HIPC(c, hipGetLastError());
    return FLM_OK;
}
int d2h(flm_ctx* c, void* dst, const void* src_dev, size_t bytes) {
    for (size_t o = 0; o < bytes; o += c->bounce_bytes) {
        const size_t nb = bytes - o < c->bounce_bytes ? bytes - o : c->bounce_bytes;
        const bool last = o + nb >= bytes;
        int r = copy_words(c, (const char*)src_dev + o, c->bounce, nb, last); if (r) return r;
        HIPC(c, hipStreamSynchronize(c->stream));
        memcpy((char*)dst + o, c->bounce, nb);
        if (last) { c->err_word = *(volatile int*)(c->bounce + c->bounce_bytes); c->err_word_fresh = true; }   // (read behind the call's last kernel: what xwg_check looks at)
    }
    return FLM_OK;
}
int h2d(flm_ctx* c, void* dst_dev, const void* src, size_t bytes) {
    for (size_t o = 0; o < bytes; o += c->bounce_bytes) {
        const size_t nb = bytes - o < c->bounce_bytes ? bytes - o : c->bounce_bytes;
        if (o) HIPC(c, hipStreamSynchronize(c->stream));                          // (the previous piece has left the buffer)
        memcpy(c->bounce, (const char*)src + o, nb);
        int r = copy_words(c, c->bounce, (char*)dst_dev + o, nb); if (r) return r;
    }
    return FLM_OK;
}
// feed tokens[0..n) sequentially (row i of the reference's batched prefill depends only on rows
// <= i through the KV cache, so token-by-token evaluation performs the same per-row arithmetic).
int feed(flm_ctx* c, const int32_t* tokens, int n, int pos, int final_advance) {
    int r;
    if (n > c->prompt_cap) return fail(c, FLM_ERR_INVALID, "more tokens than max_seq_len");
    for (int i = 0; i < n; ++i) if (tokens[i] < 0 || tokens[i] >= c->d.vocab_size) return fail(c, FLM_ERR_INVALID, "token id out of range");
    { const int rc = h2d(c, c->prompt_dev, tokens, sizeof(int) * (size_t)n); if (rc) return rc; }     // (prompt_cap <= the bounce buffer: one piece; the stream orders it in front of the kernels, nothing touches the buffer before the call's read-back)
    // batched: single GPU always; tensor parallel over the peer-to-peer exchange with the matrix-core kernels (the kernels that store their
    // column slices into the peers' buffers)
    const bool tp_ok = c->world > 1 && c->p2p && c->tp_prefill;          // agreed by all ranks at flm_p2p_import
    if (c->use_prefill && (c->world == 1 || tp_ok) && n - 1 >= kPrefillMin) {
        // all tokens but the last in one batch (cache rows only), then the last one through the decode kernels
        r = prefill_batched_qt(c, n - 1, pos);
        if (r) return r;
        r = set_state(c, pos + n - 1, tokens[n - 1], 0); if (r) return r;
        return run_token(c, true, final_advance, pos + n);
    }
    r = set_state(c, pos, tokens[0], 0); if (r) return r;
    for (int i = 0; i + 1 < n; ++i) { r = run_token(c, false, 2, pos + i + 1); if (r) return r; }
    // last token: classifier; state.step is reset so out_tokens[0] receives the argmax
    if (n > 1) {
        // step was used as the prompt cursor; zero it for the argmax slot
        hipLaunchKernelGGL(k_set_step, dim3(1), dim3(64), 0, c->stream, c->state, 0);
        HIPC(c, hipGetLastError());
    }
    return run_token(c, true, final_advance, pos + n);
}

} // namespace fh

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* flm_last_error(const flm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int flm_plan_shards(const flm_model_desc* d, int rank, int world, flm_shard_plan* out) {
    if (!d || !out || world < 1 || rank < 0 || rank >= world) return FLM_ERR_INVALID;
    // Every matmul is split by OUTPUT ROWS, exactly like the reference's worker threads (split_rows,
    // transformer.cpp:264-287): a row is always reduced on one rank in the reference's order, so sharded
    // results stay bit-identical to the single-GPU / CPU path; ranks exchange activations by all-gather,
    // which needs equal contiguous slices.
    if (d->n_heads % world || d->hidden_dim % world || d->dim % world) return FLM_ERR_UNSUPPORTED;
    out->head_count = d->n_heads / world;      out->head_begin = out->head_count * rank;
    out->hidden_count = d->hidden_dim / world; out->hidden_begin = out->hidden_count * rank;
    out->dim_count = d->dim / world;           out->dim_begin = out->dim_count * rank;
    const int slot = (d->vocab_size + world - 1) / world;  // ceil split: slot layout == vocab layout, padding past vocab_size
    int b = slot * rank, n = d->vocab_size - b; if (n > slot) n = slot; if (n < 0) n = 0;
    out->vocab_begin = b; out->vocab_count = n;
    return FLM_OK;
}

int flm_comm_unique_id(void* out128) {
    if (!out128) return FLM_ERR_INVALID;
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) { g_last_error = ncclGetErrorString(r); return FLM_ERR_COMM; }
    memcpy(out128, &id, 128);
    return FLM_OK;
}

int flm_ctx_create(const flm_model_desc* desc, int device_id, int rank, int world, const void* comm_id, flm_ctx** out) {
    if (!desc || !out) return fail(nullptr, FLM_ERR_INVALID, "null argument");
    *out = nullptr;
    const auto& d = *desc;
    if (d.dim < 64 || d.hidden_dim < 64 || d.n_layers < 1 || d.n_heads < 1 || d.vocab_size < 1 || d.max_seq_len < 1)
        return fail(nullptr, FLM_ERR_INVALID, "invalid model dimensions");
    if (d.quant_group_size != kGroup) return fail(nullptr, FLM_ERR_UNSUPPORTED, "quant_group_size must be 64");
    if (d.quant_type != FLM_QT_INT8 && d.quant_type != FLM_QT_INT16) return fail(nullptr, FLM_ERR_UNSUPPORTED, "quant_type must be INT8 or INT16");
    if (d.n_kv_heads != d.n_heads) return fail(nullptr, FLM_ERR_UNSUPPORTED, "n_kv_heads != n_heads: the reference's grouped-query path is broken (transformer.cpp:449); not reproduced");
    if (d.dim % d.n_heads || d.dim % kGroup || d.hidden_dim % kGroup) return fail(nullptr, FLM_ERR_INVALID, "dim/hidden_dim must be multiples of 64 and dim of n_heads");
    const int hs = d.dim / d.n_heads;
    if (hs % 8 || hs < 32 || hs > 256) return fail(nullptr, FLM_ERR_UNSUPPORTED, "head_size must be a multiple of 8 in [32, 256] (the reference's 8-lane dot_product path, x86_simd.cpp:1677-1699; 256: the attention tile staging)");
    if (world < 1 || rank < 0 || rank >= world) return fail(nullptr, FLM_ERR_INVALID, "rank/world");
    if (attn_lds_bytes(d.max_seq_len, hs) > kLdsMax)
        return fail(nullptr, FLM_ERR_UNSUPPORTED, "max_seq_len: a head's scores (4 bytes per position) and its K/V tiles must fit the 160 KiB of LDS of one CU");

    flm_ctx* c = new flm_ctx();
    c->d = d; c->device = device_id; c->rank = rank; c->world = world; c->hs = hs; c->esz = esz_of(d.quant_type);
    int r = flm_plan_shards(desc, rank, world, &c->plan);
    if (r) { delete c; return fail(nullptr, r, "cannot shard this model over the requested world size"); }
    c->heads_local = c->plan.head_count; c->dim_local = c->heads_local * hs; c->hidden_local = c->plan.hidden_count;
    c->drow_begin = c->plan.dim_begin; c->drow_count = c->plan.dim_count;
    c->vocab_slot = (d.vocab_size + world - 1) / world;
    auto bail = [&](int code) { g_last_error = c->err; flm_ctx_destroy(c); return code; };
#define HIPB(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->err = std::string(#expr " failed: ") + hipGetErrorString(e_); return bail(FLM_ERR_HIP); } } while (0)
    HIPB(hipSetDevice(device_id));
    hipDeviceProp_t prop; HIPB(hipGetDeviceProperties(&prop, device_id));
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; c->cu_total = c->cu_count;

    HIPB(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (comm_id) {   // (without an id the ranks exchange peer to peer only: flm_p2p_export / flm_p2p_import; world 1 with an id: the sharded token path
                     //  with RCCL exchanges over a 1-rank communicator -- how the tests run the RCCL branch on a 1-GPU box)
        ncclUniqueId id; memcpy(&id, comm_id, 128);
        ncclResult_t nr = ncclCommInitRank(&c->comm, world, id, rank);
        if (nr != ncclSuccess) { c->err = std::string("ncclCommInitRank failed: ") + ncclGetErrorString(nr); return bail(FLM_ERR_COMM); }
    }
    const int L = d.n_layers, qt = d.quant_type;
    c->layers.resize(L);
    for (int l = 0; l < L; ++l) {
        LayerW& w = c->layers[l];
        if (alloc_qmat(c, w.qkv, 3 * c->dim_local, d.dim, qt, true) || alloc_qmat(c, w.o, c->drow_count, d.dim, qt, true) ||
            alloc_qmat(c, w.w13, 2 * c->hidden_local, d.dim, qt, true) ||
            alloc_qmat(c, w.w2, c->drow_count, d.hidden_dim, qt, true)) return bail(FLM_ERR_OOM);
        HIPB(hipMalloc((void**)&w.att_norm, d.dim * 4)); HIPB(hipMalloc((void**)&w.ffn_norm, d.dim * 4));
    }
    if (alloc_qmat(c, c->cls, c->plan.vocab_count > 0 ? c->plan.vocab_count : 1, d.dim, qt)) return bail(FLM_ERR_OOM);
    c->cls.rows = c->plan.vocab_count;
    HIPB(hipMalloc((void**)&c->out_norm, d.dim * 4));
#ifndef FLM_KV_PAD
#define FLM_KV_PAD 8
#endif
    c->kv_rows = d.max_seq_len + FLM_KV_PAD;
    const size_t kvn = (size_t)L * c->heads_local * c->kv_rows * hs;
    HIPB(hipMalloc((void**)&c->kcache, kvn * 4)); HIPB(hipMalloc((void**)&c->vcache, kvn * 4));
    HIPB(hipMemsetAsync(c->kcache, 0, kvn * 4, c->stream)); HIPB(hipMemsetAsync(c->vcache, 0, kvn * 4, c->stream));
    HIPB(hipMalloc((void**)&c->qbuf, c->dim_local * 4));
    {   // the exchange buffer: att_out | x1 | hd | logits | flag lines [4 kinds][8 ranks] (full vectors on every rank under TP)
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t o_att = 0, o_x1 = up(o_att + (size_t)d.dim * 4), o_hd = up(o_x1 + (size_t)d.dim * 4), o_lg = up(o_hd + (size_t)d.hidden_dim * 4);
        const size_t o_fl = up(o_lg + (size_t)c->vocab_slot * world * 4);
        // tensor parallel: the batched prompt path's full-width activations [tokens][dim | dim | hidden] live here too (every rank stores its
        // column slices into every rank's copy)
        const size_t pcap = d.max_seq_len < 64 ? 64 : (size_t)d.max_seq_len;
        const size_t o_hf = up(o_fl + (kXchgSlots * 8 + 1) * 64);                                  // tensor parallel: one line per head part of the whole model (k_attn_o's hand-off across ranks)
        const size_t o_px = up(o_hf + (world > 1 ? (256 + 8) * 64 : 0));                             // (+ one line per rank: k_ffn's hand-off across ranks)
        const size_t o_pa = up(o_px + (world > 1 ? pcap * d.dim * 4 : 0)), o_ph = up(o_pa + (world > 1 ? pcap * d.dim * 4 : 0));
        const size_t o_tl = up(o_ph + pcap * d.hidden_dim * 4);                                      // tensor parallel: the rank-spanning k_layers' lines [heads: 256][x1 | hd | x | cls: world x 256 each]
        const size_t o_gr = up(o_tl + (size_t)(1 + 4 * world) * kTpLinesPerRank * 64);                  // tensor parallel: the rank-spanning k_layers' granule vectors [x | x1 | att | hd] (flm_layer.h BackArgs::xg_*)
        const size_t total = world > 1 ? up(o_gr + ((size_t)3 * d.dim + d.hidden_dim) * sizeof(granule_t)) : o_fl + (kXchgSlots * 8 + 1) * 64;      // (flags: + the abort line)
        hipError_t ae = hipErrorUnknown;
        if (world > 1) { ae = hipExtMallocWithFlags((void**)&c->xbuf, total, hipDeviceMallocFinegrained); c->xbuf_fine = ae == hipSuccess; }   // written by peer GPUs
        if (ae != hipSuccess) { (void)hipGetLastError(); HIPB(hipMalloc((void**)&c->xbuf, total)); }
        c->xbuf_bytes = total; c->x_flags_off = o_fl; c->x_hflags_off = o_hf; c->x_tlines_off = world > 1 ? o_tl : 0; c->x_gran_off = world > 1 ? o_gr : 0;
        HIPB(hipMemsetAsync(c->xbuf, 0, total, c->stream));
        c->att_out = (float*)(c->xbuf + o_att); c->x1 = (float*)(c->xbuf + o_x1); c->hd = (float*)(c->xbuf + o_hd); c->logits = (float*)(c->xbuf + o_lg);
        c->peer[rank] = c->xbuf;
        if (world > 1) { c->pf_x = (float*)(c->xbuf + o_px); c->pf_att = (float*)(c->xbuf + o_pa); c->pf_hd = (float*)(c->xbuf + o_ph); c->pf_in_xbuf = true; }
        HIPB(hipMalloc((void**)&c->xepoch, 64)); HIPB(hipMemsetAsync(c->xepoch, 0, 64, c->stream));
        HIPB(hipMalloc((void**)&c->ffn_counter, 64)); HIPB(hipMemsetAsync(c->ffn_counter, 0, 64, c->stream));
    }
    if (world > 1) c->xg = (granule_t*)(c->xbuf + c->x_gran_off);                 // (cleared with the exchange buffer: tag 0, below every epoch)
    else { const size_t gb = ((size_t)6 * c->d.dim + c->d.hidden_dim) * sizeof(granule_t);   /* x | x1 | att | hd | q | k | v */ HIPB(hipMalloc((void**)&c->xg, gb)); HIPB(hipMemsetAsync(c->xg, 0, gb, c->stream)); }
    HIPB(hipMalloc((void**)&c->flag_lines, 1536 * 64)); HIPB(hipMalloc((void**)&c->xwg_err, 64));   // lines 0..255: k_attn_o's heads, 256..511: split heads' scores, 512..767: k_ffn, 768..1023: k_qkv_attn_o's QKV rows, 1024..1279: k_attn_ffn's x1 rows (k_embed clears all 1536)
    HIPB(hipMemsetAsync(c->flag_lines, 0, 1536 * 64, c->stream)); HIPB(hipMemsetAsync(c->xwg_err, 0, 64, c->stream));
    {   // the one-launch token (k_layers<.., TAIL>): [0] its epoch base, one flag line per classifier workgroup, their argmax slots
        const size_t tail_bytes = (16 + 256 * 16) * 4 + 256 * 2 * 4;
        HIPB(hipMalloc((void**)&c->tail_mem, tail_bytes)); HIPB(hipMemsetAsync(c->tail_mem, 0, tail_bytes, c->stream));
        const unsigned e0 = 4096u; HIPB(hipMemcpyAsync(c->tail_mem, &e0, 4, hipMemcpyHostToDevice, c->stream));
    }
    HIPB(hipMalloc((void**)&c->eng_base, 64)); HIPB(hipMemsetAsync(c->eng_base, 0, 64, c->stream));   // the token's epoch base
    for (int k = 0; k < 2; ++k) {   // k_layers' argument blocks, one set per head split (filled by layers_prepare; allocated here: nothing is allocated inside a forward)
        HIPB(hipMalloc((void**)&c->la_dev[k], sizeof(LayerArgs) * (size_t)d.n_layers)); HIPB(hipMalloc((void**)&c->tail_dev[k], sizeof(TailArgs)));
    }
    HIPB(hipMalloc(&c->att_q, (size_t)d.dim * c->esz)); HIPB(hipMalloc((void**)&c->att_qs, (size_t)(d.dim / kGroup) * 4));
    HIPB(hipMalloc((void**)&c->att_sc, (size_t)c->heads_local * d.max_seq_len * 8)); HIPB(hipMemsetAsync(c->att_sc, 0, (size_t)c->heads_local * d.max_seq_len * 8, c->stream));   // (8 bytes per score: the parts of a split head exchange them as {score, tag} granules inside k_layers' granule launches, as floats elsewhere)
    HIPB(hipMalloc((void**)&c->state, sizeof(DecodeState)));
    HIPB(hipMemsetAsync(c->state, 0, sizeof(DecodeState), c->stream));
    std::vector<float> cs, sn; build_rope_table(hs, d.max_seq_len, cs, sn);
    HIPB(hipMalloc((void**)&c->rope_cos, cs.size() * 4)); HIPB(hipMalloc((void**)&c->rope_sin, sn.size() * 4));
    // (copies on the context's stream, never on the legacy stream: another context's thread may be capturing its token graph)
    HIPB(hipMemcpyAsync(c->rope_cos, cs.data(), cs.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPB(hipMemcpyAsync(c->rope_sin, sn.data(), sn.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPB(hipStreamSynchronize(c->stream));                                    // (cs / sn go out of scope)
    if (alloc_run_bufs(c)) return bail(FLM_ERR_OOM);
    HIPB(hipStreamSynchronize(c->stream));
    {   // can the fused launches run here?  Decided once, up front -- not after a 20 ms stall in the first token
        static std::mutex mu; static bool attr_done[64] = {false};
        {
            std::lock_guard<std::mutex> lk(mu);
            if (device_id >= 0 && device_id < 64 && !attr_done[device_id]) {
                HIPB(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_census), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
                attr_done[device_id] = true;
            }
        }
        c->resident = run_census(c) == 1 ? 1 : 0;
        if (!c->resident) { c->fuse_attn_o = 0; c->fuse_ffn = 0; c->fuse_qkv = 0; c->fuse_back = 0; c->attn_split = 0; }
    }
#undef HIPB
    *out = c;
    return FLM_OK;
}

void flm_ctx_destroy(flm_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto& g : c->graphs) hipGraphExecDestroy(g.second);
    auto fq = [](QMat& m) { if (m.q) hipFree(m.q); if (m.s) hipFree(m.s); if (m.st) hipFree(m.st); };
    for (auto& l : c->layers) { fq(l.qkv); fq(l.o); fq(l.w13); fq(l.w2); if (l.att_norm) hipFree(l.att_norm); if (l.ffn_norm) hipFree(l.ffn_norm); }
    fq(c->cls);
    for (int r = 0; r < c->world; ++r) if (c->peer_opened[r] && c->peer[r]) hipIpcCloseMemHandle(c->peer[r]);
    void* ptrs[] = {c->emb, c->emb_s, c->out_norm, c->kcache, c->vcache, c->xbuf, c->xepoch, c->qbuf,
                    c->rope_cos, c->rope_sin, c->state, c->prompt_dev, c->out_tokens_dev,
                    c->world > 1 ? nullptr : (void*)c->xg, c->flag_lines, c->xwg_err, c->att_q, c->att_qs, c->att_sc, c->trace, c->eng_base, c->ffn_counter, c->la_dev[0], c->la_dev[1], c->tail_dev[0], c->tail_dev[1], c->tail_mem,
                    c->pf_in_xbuf ? nullptr : c->pf_x, c->pf_qkv, c->pf_q, c->pf_in_xbuf ? nullptr : c->pf_att, c->pf_gu, c->pf_in_xbuf ? nullptr : c->pf_hd, c->pf_xs, c->pf_xq, c->pf_scores};
    for (void* p : ptrs) if (p) hipFree(p);
    if (c->bounce) hipHostFree(c->bounce);
    if (c->comm) ncclCommDestroy(c->comm);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

// ---- tensor parallel, peer-to-peer bootstrap ---------------------------------------------------
// Each rank exports a 128-byte blob (IPC handle of its exchange buffer, its process and device); the caller gathers the blobs
// of all ranks in rank order by whatever transport it has (bench.py: torch.distributed.all_gather; a C++ host: MPI, a file, a
// socket) and hands them to flm_p2p_import, which maps every peer's buffer (hipIpcOpenMemHandle; ranks living in the SAME
// process share the pointer directly).  From then on activation slices travel by direct stores over xGMI plus one flag
// round (k_xchg) instead of an RCCL all-gather, and the token is replayed from a hipGraph like the single-GPU one.
namespace {
struct P2pBlob { unsigned long long magic; int pid, device, rank, world; unsigned long long bytes; void* raw; hipIpcMemHandle_t h; int caps; /* bit 0: this rank can run the batched prompt path */ char pad[128 - 8 - 16 - 8 - 8 - sizeof(hipIpcMemHandle_t) - 4]; };
static_assert(sizeof(P2pBlob) == FLM_P2P_BLOB_BYTES, "blob size");
constexpr unsigned long long kP2pMagic = 0x464C4D5032503031ull;   // "FLMP2P01"
}
#include <unistd.h>
int flm_p2p_export(flm_ctx* c, void* blob128) {
    if (!c || !blob128) return FLM_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    P2pBlob b{}; b.magic = kP2pMagic; b.pid = (int)getpid(); b.device = c->device; b.rank = c->rank; b.world = c->world; b.bytes = c->xbuf_bytes; b.raw = c->xbuf;
    // caps: bit 0 batched prompt path possible, 1 one workgroup per CU resident (census), 2 heads can be split over workgroups, 3 fold_xchg, 4-5 tp_fuse_attn, 6 tp_fuse_ffn,
    //       7 tp_trust_fused, 8-11 cu_parts, 12-15 attn_split
    {
        const int Gfull = c->hs / kSplitDims;
        const bool can = c->hs % kSplitDims == 0 && Gfull >= 2 && c->hs <= 128 && c->d.max_seq_len <= kSplitMaxSeq && c->heads_local * Gfull + 8 <= c->cu_count && c->heads_local * Gfull <= 256;
        b.caps = (tp_prefill_capable(c) ? 1 : 0) | (c->resident ? 2 : 0) | (can ? 4 : 0) | (c->fold_xchg ? 8 : 0) | ((c->tp_fuse_attn < 0 ? 0 : c->tp_fuse_attn > 2 ? 2 : c->tp_fuse_attn) << 4) | (c->tp_fuse_ffn ? 64 : 0)
               | (c->tp_trust_fused ? 128 : 0) | ((c->cu_parts & 15) << 8) | ((c->attn_split < 0 ? 0 : c->attn_split > 15 ? 15 : c->attn_split) << 12)
               | (c->tp_fuse_layers && c->fuse_token ? 1 << 16 : 0) | ((c->cu_count & 1023) << 17) | (c->gr_edges ? 1 << 27 : 0);          // 16 tp_fuse_layers, 17-26 the CUs this rank's launches are sized for
    }
    HIPC(c, hipIpcGetMemHandle(&b.h, c->xbuf));
    memcpy(blob128, &b, sizeof b);
    return FLM_OK;
}
int flm_p2p_import(flm_ctx* c, const void* blobs, int n) {
    if (!c || !blobs || n != c->world) return FLM_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    const P2pBlob* b = (const P2pBlob*)blobs;
    // the batched prompt path runs 4 exchanges per layer, token-by-token feeding 4 per layer and TOKEN: every rank must take the same one
    // (decided once, from what all ranks can do; options that would change it afterwards are refused)
    // validate every blob before anything of the context changes: a foreign blob must not leave a half-updated group structure behind cached graphs
    for (int r = 0; r < n; ++r)
        if (b[r].magic != kP2pMagic || b[r].rank != r || b[r].world != c->world || b[r].bytes != c->xbuf_bytes) return fail(c, FLM_ERR_INVALID, "p2p_import: blobs are not those of this tensor-parallel group, in rank order");
    // (from here on the group's structure may change: graphs captured under the old one must not be replayed, whatever happens below)
    for (auto& g : c->graphs) hipGraphExecDestroy(g.second);
    c->graphs.clear();
    c->la_valid[0] = c->la_valid[1] = false;
    c->tp_prefill = true;
    for (int r = 0; r < n; ++r) if (!(b[r].caps & 1)) c->tp_prefill = false;
    c->ranks_on_device = 0;
    for (int r = 0; r < n; ++r) if (b[r].device == c->device) ++c->ranks_on_device;
    if (!tp_prefill_capable(c)) c->tp_prefill = false;
    {   // the group's launch structure: the weakest any rank can do, computed alike on every rank from the same blobs
        bool fold = true, span = true, can = true, multi_dev = false, trust = true, tpl = true, ggr = true; int fa = 2, ff = 1, split = (b[0].caps >> 12) & 15;
        for (int r = 0; r < n; ++r) {
            const int cp = (b[r].caps >> 8) & 15; int rod = 0;
            for (int q = 0; q < n; ++q) { if (b[q].device == b[r].device) ++rod; else multi_dev = true; }
            const bool fold_r = (b[r].caps & 8) && cp >= rod, span_r = fold_r && ((b[r].caps & 2) || cp > 1);
            fold = fold && fold_r; span = span && span_r; can = can && (b[r].caps & 4); trust = trust && (b[r].caps & 128);
            const int fa_r = (b[r].caps >> 4) & 3; if (fa_r < fa) fa = fa_r;
            if (!(b[r].caps & 64)) ff = 0;
            if (!(b[r].caps & (1 << 27))) ggr = false;
            if (((b[r].caps >> 12) & 15) != split) split = 0;                   // (ranks that disagree: nobody splits)
            if (!(b[r].caps & (1 << 16)) || ((b[r].caps >> 17) & 1023) != ((b[0].caps >> 17) & 1023)) tpl = false;   // (the rank-spanning k_layers: every rank wants it, identical launch geometry)
        }
        // ranks on distinct devices: the folded exchanges and the rank-spanning launches rely on system-scope store / flag ordering over xGMI that was only ever
        // exercised between CU partitions of ONE GPU -> the k_xchg launches (a flag round behind a kernel boundary) unless every rank says "tp_trust_fused"
        if (multi_dev && !trust) { fold = false; span = false; }
        if (4 * c->d.n_layers + 2 >= (int)kEpochStride) { fold = false; span = false; }   // (the folded rounds' epoch values 4 l + kind + 1 must stay inside one token's stride)
        c->grp_fold = fold; c->grp_span = span; c->grp_can_split = can; c->grp_tpfa = span ? fa : 0; c->grp_tpff = span ? ff : 0; c->grp_split = can ? split : 0; c->grp_tpl = span && tpl; c->grp_gr = ggr;
    }
    for (int r = 0; r < n; ++r) {
        if (r == c->rank) continue;
        if (c->peer[r]) continue;                                         // already mapped
        if (b[r].device != c->device) {                                   // (also for a peer of the same process on another GPU: one host thread per GPU)
            int can = 0; HIPC(c, hipDeviceCanAccessPeer(&can, c->device, b[r].device));
            if (!can) return fail(c, FLM_ERR_UNSUPPORTED, "p2p_import: no peer access between the two devices");
            hipError_t e = hipDeviceEnablePeerAccess(b[r].device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIPC(c, e);
            (void)hipGetLastError();
        }
        if (b[r].pid == (int)getpid()) { c->peer[r] = (char*)b[r].raw; continue; }   // same process: the pointer is valid here, only the IPC mapping is skipped
        void* p = nullptr;
        HIPC(c, hipIpcOpenMemHandle(&p, b[r].h, hipIpcMemLazyEnablePeerAccess));
        c->peer[r] = (char*)p; c->peer_opened[r] = true;
    }
    c->p2p = 1;
    for (auto& g : c->graphs) hipGraphExecDestroy(g.second);
    c->graphs.clear();
    {   // the group's structure is known now: argument blocks and token graphs (captured, not launched -- nobody waits for a peer here); an error here resurfaces at the first forward
        const std::string err0 = c->err, gerr0 = g_last_error;
        if (prepare_all(c) != FLM_OK) { c->err = err0; g_last_error = gerr0; (void)hipGetLastError(); }
    }
    return FLM_OK;
}

int flm_set_option(flm_ctx* c, const char* key, int value) {
    if (!c || !key) return FLM_ERR_INVALID;
    std::string k(key);
    if (c->world > 1 && c->p2p && (k == "use_mfma" || k == "use_pv_mfma" || k == "use_prefill_mq" || k == "use_qk_mfma"))
        return fail(c, FLM_ERR_STATE, "set_option: which prompt kernels a tensor-parallel group runs is agreed at flm_p2p_import; set this option on every rank before importing (\"use_prefill\" may be switched later, on every rank alike)");
    if (!c->resident && value != 0 && (k == "fuse_attn_o" || k == "fuse_ffn" || k == "fuse_qkv" || k == "fuse_back" || k == "attn_split"))
        return fail(c, FLM_ERR_UNSUPPORTED, "set_option: this device does not keep one workgroup per CU resident (census at flm_ctx_create); the fused launches stay off");
    for (const char* tk : kTuningKeys)
        if (k == tk && !c->tuning) return fail(c, FLM_ERR_INVALID, "set_option: an experiment dial (csrc/flm_tuning.h), not part of the boundary: set option \"tuning\" 1 first");
    if (k == "tuning") { c->tuning = value != 0; return FLM_OK; }
    if (k == "wg_per_cu") { c->wg_per_cu = value > 0 ? value : 1; }
    else if (k == "use_graph") c->use_graph = value;
    else if (k == "graph_chunks") c->graph_chunks = value;
    else if (k == "inject_wait_failure") {   // (tuning mode only: flm_tuning.h)
        if (value) { const int one = 1; HIPC(c, hipMemcpyAsync(c->xwg_err, &one, 4, hipMemcpyHostToDevice, c->stream)); HIPC(c, hipStreamSynchronize(c->stream)); }
        return FLM_OK;
    }
    else if (k == "use_prefill") c->use_prefill = value;
    else if (k == "use_mfma") c->use_mfma = value;
    else if (k == "use_pv_mfma") c->use_pv_mfma = value;
    else if (k == "fuse_attn_o") c->fuse_attn_o = value;
    else if (k == "fuse_ffn") c->fuse_ffn = value;
    else if (k == "fuse_qkv") c->fuse_qkv = value;
    else if (k == "fuse_back") c->fuse_back = value;
    else if (k == "fuse_layer") c->fuse_layer = value;
    else if (k == "fuse_token") c->fuse_token = value;
    else if (k == "fuse_tail") c->fuse_tail = value;
    else if (k == "tok_nstq") c->tok_nstq = value;
    else if (k == "tok_preq") c->tok_preq = value;
    else if (k == "back_nst13") c->back_nst13 = value;
    else if (k == "back_nst13_head") c->back_nst13_head = value;
    else if (k == "back_nst2") c->back_nst2 = value;
    else if (k == "back_pre13") c->back_pre13 = value;
    else if (k == "back_pre2") c->back_pre2 = value;
    else if (k == "back_ao") c->back_ao = value;
    else if (k == "back_ao2") c->back_ao2 = value;
    else if (k == "gr_edges") c->gr_edges = value;
    else if (k == "back_nwo") c->back_nwo = value;
    else if (k == "attn_kpre") c->attn_kpre = value;
    else if (k == "use_prefill_mq") c->use_prefill_mq = value;
    else if (k == "attn_split") c->attn_split = value;
    else if (k == "fold_xchg") c->fold_xchg = value;
    else if (k == "tp_fuse_attn") c->tp_fuse_attn = value;
    else if (k == "tp_fuse_ffn") c->tp_fuse_ffn = value;
    else if (k == "tp_fuse_layers") c->tp_fuse_layers = value;
    else if (k == "tp_fence") c->tp_fence = value < 0 ? -1 : value & 3;
    else if (k == "tp_trust_fused") c->tp_trust_fused = value;
    else if (k == "force_tp") c->force_tp = value;
    else if (k == "cu_parts") {
        // confine this context's stream to 1 / value of the device's CUs (part rank % value) and size its launches for them: how several tensor-parallel
        // ranks share ONE GPU without a waiting consumer launch taking the CUs its peers' producers need (tests; a real rank owns a device: value 1)
        if (value < 1 || value > 8 || c->cu_total % value) return fail(c, FLM_ERR_INVALID, "cu_parts: 1, 2, 4 or 8");
        HIPC(c, hipStreamSynchronize(c->stream));
        hipStream_t ns = nullptr;
        if (value == 1) HIPC(c, hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
        else {
            const int per = c->cu_total / value, first = (c->rank % value) * per;
            std::vector<uint32_t> mask((c->cu_total + 31) / 32, 0u);
            for (int i = first; i < first + per; ++i) mask[i / 32] |= 1u << (i % 32);
            HIPC(c, hipExtStreamCreateWithCUMask(&ns, (uint32_t)mask.size(), mask.data()));
        }
        HIPC(c, hipStreamDestroy(c->stream));
        c->stream = ns; c->cu_parts = value; c->cu_count = c->cu_total / value;
        if (value > 1) { c->fuse_attn_o = 0; c->fuse_ffn = 0; c->fuse_qkv = 0; c->fuse_back = 0; }
    }
    else if (k == "use_qk_mfma") c->use_qk_mfma = value;
    else if (k == "use_p2p") {     // 0: exchange by RCCL all-gathers although the peers are mapped (needs the communicator); 1: back to peer-to-peer
        if (value) { for (int r = 0; r < c->world; ++r) if (!c->peer[r]) return fail(c, FLM_ERR_STATE, "use_p2p: flm_p2p_import has not mapped every peer"); }
        else if (c->world > 1 && !c->comm) return fail(c, FLM_ERR_STATE, "use_p2p 0: no RCCL communicator (comm_id was NULL at create)");
        c->p2p = value ? 1 : 0;
    }
    else if (kAblate && k == "ablate") c->ablate = value;              // FLM_ABLATE builds only: a product library cannot skip work
    else if (kAblate && k == "trace") {   // value = kernel class to trace (KC_*), -1 off
        c->trace_class = value;
        if (!c->trace) { HIPC(c, hipMalloc((void**)&c->trace, 131072 * 8)); }
        HIPC(c, hipMemset(c->trace, 0, 131072 * 8));
    }
    else return fail(c, FLM_ERR_INVALID, "unknown option");
    for (auto& g : c->graphs) hipGraphExecDestroy(g.second);
    c->graphs.clear();
    c->la_valid[0] = c->la_valid[1] = false;
    return FLM_OK;
}


int flm_query(flm_ctx* c, const char* key, int* value) {
    if (!c || !key || !value) return FLM_ERR_INVALID;
    const std::string k(key);
    const struct { const char* k; int v; } tab[] = {
        {"tuning", c->tuning ? 1 : 0}, {"wg_per_cu", c->wg_per_cu}, {"use_graph", c->use_graph}, {"graph_chunks", c->graph_chunks}, {"use_prefill", c->use_prefill}, {"use_mfma", c->use_mfma}, {"use_pv_mfma", c->use_pv_mfma},
        {"fuse_attn_o", c->fuse_attn_o}, {"fuse_ffn", c->fuse_ffn}, {"fuse_qkv", c->fuse_qkv}, {"fuse_back", c->fuse_back}, {"fuse_layer", c->fuse_layer}, {"fuse_token", c->fuse_token}, {"fuse_tail", c->fuse_tail}, {"tok_nstq", c->tok_nstq}, {"tok_preq", c->tok_preq}, {"back_nst13", c->back_nst13}, {"back_nst13_head", c->back_nst13_head}, {"back_nst2", c->back_nst2}, {"back_pre13", c->back_pre13}, {"back_pre2", c->back_pre2}, {"back_ao", c->back_ao}, {"back_ao2", c->back_ao2}, {"gr_edges", c->gr_edges}, {"back_nwo", c->back_nwo}, {"attn_kpre", c->attn_kpre}, {"kpre_active", c->la_valid[1] ? (int)c->la_p[1].kpre_off : -1}, {"nwo_active", c->la_valid[0] ? c->la_p[0].nw_o : -1}, {"preq_active", c->la_valid[0] ? c->la_p[0].preq : -1}, {"pre13_active", c->la_valid[0] ? c->la_p[0].pre13 : -1}, {"preq_active_split", c->la_valid[1] ? c->la_p[1].preq : -1}, {"pre13_active_split", c->la_valid[1] ? c->la_p[1].pre13 : -1},   /* the early register sets the last one-launch token ran with (plan_layer's by-launch values) */ {"gr_active", ((c->la_valid[0] && c->la_ok[0] && c->la_p[0].gr && (c->world > 1 ? (c->p2p && c->grp_tpl) : c->tail_ok[0])) ? 1 : 0) | ((c->la_valid[1] && c->la_ok[1] && c->la_p[1].gr && (c->world > 1 ? (c->p2p && c->grp_tpl) : c->tail_ok[1])) ? 2 : 0)},   /* the granule hand-offs are what the one-launch token / the rank-spanning launch runs: bit 0 one workgroup per head, bit 1 split heads */ {"use_prefill_mq", c->use_prefill_mq}, {"attn_split", c->attn_split},
        {"use_qk_mfma", c->use_qk_mfma}, {"use_p2p", c->p2p}, {"fold_xchg", c->fold_xchg}, {"tp_fuse_attn", c->tp_fuse_attn}, {"tp_fuse_ffn", c->tp_fuse_ffn}, {"cu_parts", c->cu_parts}, {"fold_active", (c->world > 1 && c->p2p && c->grp_fold) ? 1 : 0}, {"span_active", (c->world > 1 && c->p2p && c->grp_span) ? 1 : 0}, {"tp_trust_fused", c->tp_trust_fused}, {"force_tp", c->force_tp},
        {"tp_fuse_layers", c->tp_fuse_layers}, {"tp_fence", c->tp_fence}, {"tp_fence_active", c->tp_fence >= 0 ? c->tp_fence : (c->ranks_on_device == c->world ? 0 : 3)}, {"grp_gr", (c->world > 1 && c->grp_gr) ? 1 : 0}, {"grp_tp_fuse_layers", (c->world > 1 && c->p2p && c->grp_tpl) ? 1 : 0}, {"tp_layers_active", (c->world > 1 && c->p2p && c->grp_tpl && (c->la_valid[0] || c->la_valid[1])) ? (c->la_valid[0] && c->la_ok[0] ? 1 : 0) | (c->la_valid[1] && c->la_ok[1] ? 2 : 0) : -1},   /* the rank-spanning k_layers was planned: bit 0 one workgroup per head, bit 1 split heads */ {"grp_tp_fuse_attn", c->grp_tpfa}, {"grp_tp_fuse_ffn", c->grp_tpff}, {"grp_attn_split", c->grp_split}, {"resident", c->resident}, {"fallback", c->fell_back}, {"fallback_active", c->fb_active ? 1 : 0},
        {"ao_active", c->la_ok[0] ? (c->la_p[0].ao_o ? 1 : 0) | (c->la_p[0].ao_2 ? 2 : 0) : -1},      // which hand-offs of the token's launch (short contexts) are consumed in arrival order; -1: that launch was not planned (yet)
        {"token_path", (c->world == 1 ? ((c->fuse_attn_o ? 1 : 0) | (c->fuse_ffn ? 2 : 0) | (c->fuse_attn_o && c->fuse_qkv == 1 ? 4 : 0) | (c->fuse_attn_o && c->fuse_qkv >= 2 ? 8 : 0) | (c->fuse_back && c->fuse_attn_o && c->fuse_ffn ? (c->fuse_layer ? 128 + 256 + (c->fuse_token ? 512 + (c->fuse_tail && c->tail_ok[0] ? 1024 : 0) : 0) : 128) : 0)) : 0) | (c->attn_split ? 64 : 0)},
    };
    for (const auto& t : tab) if (k == t.k) { *value = t.v; return FLM_OK; }
    return fail(c, FLM_ERR_INVALID, "query: unknown key");
}

static int upload_tensor_impl(flm_ctx* c, int kind, int layer, int src_qt, const void* values, const float* scales, int rows, int cols) {
    if (c) { c->st_ready = false; c->la_valid[0] = c->la_valid[1] = false; }       // (the device-resident argument blocks of k_layers hold pointers into the tensors -- the embedding table's is re-allocated below -- and depend on their types)
    if (!c || !values) return FLM_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    const auto& d = c->d;
    const int hs = c->hs;
    if (src_qt != FLM_QT_NONE && !scales) return fail(c, FLM_ERR_INVALID, "quantized tensor without scales");
    if (kind >= 16 && (layer < 0 || layer >= d.n_layers)) return fail(c, FLM_ERR_INVALID, "layer out of range");
    auto vec = [&](float* dst, int n) -> int {
        if (src_qt != FLM_QT_NONE || (size_t)rows * cols != (size_t)n) return fail(c, FLM_ERR_INVALID, "norm tensor must be fp32 [dim]");
        HIPC(c, hipMemcpyAsync(dst, values, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        return FLM_OK;
    };
    const int hb = c->plan.head_begin * hs, hn = c->dim_local;
    switch (kind) {
    case FLM_T_TOKEN_EMBD: {
        if (rows != d.vocab_size || cols != d.dim) return fail(c, FLM_ERR_INVALID, "embedding shape");
        const size_t n = (size_t)rows * cols;
        if (c->emb) { hipFree(c->emb); c->emb = nullptr; } if (c->emb_s) { hipFree(c->emb_s); c->emb_s = nullptr; }
        HIPC(c, hipMalloc(&c->emb, n * esz_of(src_qt)));
        HIPC(c, hipMemcpyAsync(c->emb, values, n * esz_of(src_qt), hipMemcpyHostToDevice, c->stream));
        if (src_qt != FLM_QT_NONE) { HIPC(c, hipMalloc((void**)&c->emb_s, n / kGroup * 4)); HIPC(c, hipMemcpyAsync(c->emb_s, scales, n / kGroup * 4, hipMemcpyHostToDevice, c->stream)); }
        HIPC(c, hipStreamSynchronize(c->stream));
        c->emb_qt = src_qt; c->got_emb = true;
        for (auto& g : c->graphs) hipGraphExecDestroy(g.second);
        c->graphs.clear();
        return FLM_OK; }
    case FLM_T_OUTPUT_NORM: { int r = vec(c->out_norm, d.dim); if (!r) c->got_out_norm = true; return r; }
    case FLM_T_INPUT_NORM:  { int r = vec(c->layers[layer].att_norm, d.dim); if (!r) c->layers[layer].got |= 1u << 0; return r; }
    case FLM_T_POST_NORM:   { int r = vec(c->layers[layer].ffn_norm, d.dim); if (!r) c->layers[layer].got |= 1u << 1; return r; }
    case FLM_T_CLASSIFIER: {
        if (rows != d.vocab_size || cols != d.dim) return fail(c, FLM_ERR_INVALID, "classifier shape");
        if (c->plan.vocab_count > 0) { int r = upload_window(c, c->cls, 0, src_qt, values, scales, cols, c->plan.vocab_begin, c->plan.vocab_count, 0, cols); if (r) return r; }
        c->got_cls = true; return FLM_OK; }
    case FLM_T_ATTN_Q: case FLM_T_ATTN_K: case FLM_T_ATTN_V: {
        if (rows != d.dim || cols != d.dim) return fail(c, FLM_ERR_INVALID, "q/k/v shape");
        const int which = kind - FLM_T_ATTN_Q;
        int r = upload_window(c, c->layers[layer].qkv, which * hn, src_qt, values, scales, cols, hb, hn, 0, cols);
        if (!r) c->layers[layer].got |= 1u << (2 + which);
        return r; }
    case FLM_T_ATTN_O: {
        if (rows != d.dim || cols != d.dim) return fail(c, FLM_ERR_INVALID, "o shape");
        int r = upload_window(c, c->layers[layer].o, 0, src_qt, values, scales, cols, c->drow_begin, c->drow_count, 0, cols);
        if (!r) c->layers[layer].got |= 1u << 5;
        return r; }
    case FLM_T_MLP_GATE: case FLM_T_MLP_UP: {
        if (rows != d.hidden_dim || cols != d.dim) return fail(c, FLM_ERR_INVALID, "ffn1/3 shape");
        int r = upload_window(c, c->layers[layer].w13, kind == FLM_T_MLP_GATE ? 0 : c->hidden_local, src_qt, values, scales, cols, c->plan.hidden_begin, c->plan.hidden_count, 0, cols);
        if (!r) c->layers[layer].got |= 1u << (kind == FLM_T_MLP_GATE ? 6 : 7);
        return r; }
    case FLM_T_MLP_DOWN: {
        if (rows != d.dim || cols != d.hidden_dim) return fail(c, FLM_ERR_INVALID, "ffn2 shape");
        int r = upload_window(c, c->layers[layer].w2, 0, src_qt, values, scales, cols, c->drow_begin, c->drow_count, 0, cols);
        if (!r) c->layers[layer].got |= 1u << 8;
        return r; }
    default: return fail(c, FLM_ERR_INVALID, "unknown tensor kind");
    }
}
// Once per context, when its model is complete (single GPU: a tensor-parallel rank must not wait for peers at load time): one short prompt through the batched kernels and one token
// with logits, on dummy ids, so that whatever the HIP runtime sets up lazily at a first launch -- queue-side pools that grow with the number of launches in flight: 2 MiB of device
// memory at the first prompt of a process (tools/alloc_diag.py) -- is set up at LOAD time and not inside the caller's first flm_forward.  The cache rows it wrote are cleared again.
static void warm_up(flm_ctx* c) {
    if (c->warmed || c->world != 1 || c->comm || !model_complete(c)) return;
    c->warmed = true;
    const std::string err0 = c->err, gerr0 = g_last_error;
    int32_t toks[kPrefillMin + 2] = {0};
    const int n = c->d.max_seq_len > kPrefillMin + 2 ? kPrefillMin + 2 : 1;
    bool ok = feed(c, toks, n, 0, 0) == FLM_OK && hipStreamSynchronize(c->stream) == hipSuccess;
    // ... and every graph a greedy loop replays, once: chunks of 16, 8, 4, 2 tokens and the single token, for one workgroup per head and (from kSplitFrom positions on) for split
    // heads -- whatever a graph's FIRST launch costs (seen once: ~2 ms inside a timed region of 20 tokens) is paid here
    const int kEach = 2 * kChunk - 1;
    if (ok && c->d.max_seq_len > n + kEach) ok = set_state(c, n, 0, 0) == FLM_OK && run_greedy_tokens(c, n, kEach) == FLM_OK && hipStreamSynchronize(c->stream) == hipSuccess;
    if (ok && c->d.max_seq_len > kSplitFrom + 8 + kEach && attn_parts(c, kSplitFrom + 8) != attn_parts(c, 1))
        ok = set_state(c, kSplitFrom + 8, 0, 0) == FLM_OK && run_greedy_tokens(c, kSplitFrom + 8, kEach) == FLM_OK && hipStreamSynchronize(c->stream) == hipSuccess;
    if (ok) (void)xwg_check(c);                                                // (a wait that gave up here puts the context on the per-phase kernels like any other)
    const size_t kvn = (size_t)c->d.n_layers * c->heads_local * c->kv_rows * c->hs;     // the cache rows the dummy tokens wrote: cleared again
    (void)hipMemsetAsync(c->kcache, 0, kvn * 4, c->stream); (void)hipMemsetAsync(c->vcache, 0, kvn * 4, c->stream);
    (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError(); c->err = err0; g_last_error = gerr0;               // (an error here is not the caller's: it resurfaces at the first forward)
}
int flm_upload_tensor(flm_ctx* c, int kind, int layer, int src_qt, const void* values, const float* scales, int rows, int cols) {
    const int r = upload_tensor_impl(c, kind, layer, src_qt, values, scales, rows, cols);
    if (r == FLM_OK && model_complete(c)) {
        // the model's last tensor (or a replacement) has arrived: k_layers' argument blocks and the token graphs are built NOW, not inside the first forward
        // (transformer.cpp:110-130: no allocation during inference).  An error here is not the upload's: it resurfaces at the first forward.
        const std::string err0 = c->err, gerr0 = g_last_error;
        if (prepare_all(c) != FLM_OK) { c->err = err0; g_last_error = gerr0; (void)hipGetLastError(); }
        else warm_up(c);
    }
    return r;
}

int flm_prepare(flm_ctx* c) {
    if (!c) return FLM_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    if (!model_complete(c)) return fail(c, FLM_ERR_STATE, "prepare before all tensors were uploaded");
    const int r = prepare_all(c);
    if (r == FLM_OK) warm_up(c);
    return r;
}

int flm_reset_kv(flm_ctx* c) {
    if (!c) return FLM_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    const size_t kvn = (size_t)c->d.n_layers * c->heads_local * c->kv_rows * c->hs;
    HIPC(c, hipMemsetAsync(c->kcache, 0, kvn * 4, c->stream)); HIPC(c, hipMemsetAsync(c->vcache, 0, kvn * 4, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return FLM_OK;
}

// debugging tap (tests): copy an internal device buffer to the host. what: 0 x1, 1 q, 2 att_out, 3 hd, 4 kcache(layer), 5 vcache(layer), 6 logits
int flm_debug_read(flm_ctx* c, int what, int layer, float* out, size_t n) {
    if (!c || !out) return FLM_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    const size_t kvl = (size_t)c->heads_local * c->d.max_seq_len * c->hs, kvs = (size_t)c->heads_local * c->kv_rows * c->hs;   // what the caller sees ([heads][max_seq][hs]) / what a layer occupies
    const float* src = nullptr; size_t cap = 0;
    switch (what) {
    case 0: src = c->x1; cap = c->d.dim; break;
    case 1: src = c->qbuf; cap = c->dim_local; break;
    case 2: src = c->att_out; cap = c->d.dim; break;
    case 3: src = c->hd; cap = c->d.hidden_dim; break;
    case 4: case 5: {   // the cache rows of a layer, without the padding rows between two heads
        if (n > kvl || layer < 0 || layer >= c->d.n_layers) return fail(c, FLM_ERR_INVALID, "debug_read: size/layer");
        const float* base = (what == 4 ? c->kcache : c->vcache) + (size_t)layer * kvs;
        const size_t row = (size_t)c->d.max_seq_len * c->hs * 4, heads = (n * 4 + row - 1) / row;
        std::vector<float> tmp(heads * row / 4);
        HIPC(c, hipMemcpy2DAsync(tmp.data(), row, base, (size_t)c->kv_rows * c->hs * 4, row, heads, hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        memcpy(out, tmp.data(), n * 4);
        return FLM_OK; }
    case 6: src = c->logits; cap = (size_t)c->vocab_slot * c->world; break;
    case 7: {   // GEMV timeline (FLM_ABLATE builds): [grid][8] ticks relative to the earliest workgroup start; column 7 = 100 MHz ticks start -> end
        if (!c->trace || n > 4096 * 8) return fail(c, FLM_ERR_INVALID, "debug_read: no trace");
        HIPC(c, hipStreamSynchronize(c->stream));
        std::vector<unsigned long long> t(4096 * 8);
        HIPC(c, hipMemcpy(t.data(), c->trace, t.size() * 8, hipMemcpyDeviceToHost));
        // ticks relative to the first stamp of the row's workgroup (`layer` = rows per workgroup, 1 or 16): the
        // shader clocks of different XCDs are not synchronised, and fp32 cannot hold absolute tick counts
        const size_t rpw = layer > 0 ? (size_t)layer : 1;
        for (size_t i = 0; i < n; ++i) {
            const size_t row = i / 8, base = (row - row % rpw) * 8;
            out[i] = (i % 8 == 7 && rpw == 1) ? (float)t[i] : ((t[i] && t[base]) ? (float)(long long)(t[i] - t[base]) : -1.f);
        }
        return FLM_OK; }
    case 8: {   // tools/trace_skew.py (FLM_ABLATE builds, ablate & 64): columns 1 and 2 = 100 MHz real-time ticks of a workgroup's start / end, relative to the earliest start
        if (!c->trace || n > 4096 * 8) return fail(c, FLM_ERR_INVALID, "debug_read: no trace");
        HIPC(c, hipStreamSynchronize(c->stream));
        std::vector<unsigned long long> t(4096 * 8);
        HIPC(c, hipMemcpy(t.data(), c->trace, t.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (size_t i = 1; i < n; i += 8) if (t[i] && t[i] < t0) t0 = t[i];
        for (size_t i = 0; i < n; ++i) out[i] = ((i % 8 == 1 || i % 8 == 2) && t[i]) ? (float)(long long)(t[i] - t0) : -1.f;
        return FLM_OK; }
    case 10: {  // tools/trace_back.py (FLM_ABLATE builds): k_attn_ffn's stamps [workgroup][16] on the 100 MHz clock (one clock for all XCDs) as microseconds after the earliest one; -1 = not stamped
        if (!c->trace || n > 131072) return fail(c, FLM_ERR_INVALID, "debug_read: no trace");
        HIPC(c, hipStreamSynchronize(c->stream));
        std::vector<unsigned long long> t(n);
        HIPC(c, hipMemcpy(t.data(), c->trace, n * 8, hipMemcpyDeviceToHost));
        // words [0, 3 * 4096): 100 MHz stamps; from 3 * 4096 on: rows of 16 shader-clock stamps (the rmsnorm chain's stages), given as ticks after the row's first, [15] a raw count
        const size_t nrt = n < 3 * 4096 ? n : 3 * 4096;
        unsigned long long t0 = ~0ull;
        for (size_t i = 0; i < nrt; ++i) if (t[i] && t[i] < t0) t0 = t[i];
        for (size_t i = 0; i < nrt; ++i) out[i] = t[i] ? (float)((double)(long long)(t[i] - t0) * 0.01) : -1.f;
        // (words [4 * 4096, 5 * 4096): 100 MHz stamps again -- FFN13's run() --, relative to the same start)
        for (size_t i = nrt; i < n && i < 4 * 4096; ++i) { const size_t b = i - i % 16; out[i] = i % 16 == 15 ? (float)t[i] : ((t[i] && t[b]) ? (float)(long long)(t[i] - t[b]) : -1.f); }
        for (size_t i = 4 * 4096; i < n; ++i) out[i] = t[i] ? (float)((double)(long long)(t[i] - t0) * 0.01) : -1.f;
        return FLM_OK; }
    default: return fail(c, FLM_ERR_INVALID, "debug_read: unknown buffer");
    }
    if (n > cap || layer < 0 || layer >= c->d.n_layers) return fail(c, FLM_ERR_INVALID, "debug_read: size/layer");
    HIPC(c, hipMemcpyAsync(out, src, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return FLM_OK;
}

int flm_sync(flm_ctx* c) { if (!c) return FLM_ERR_INVALID; HIPC(c, hipSetDevice(c->device)); HIPC(c, hipStreamSynchronize(c->stream)); return FLM_OK; }

// Every entry point below runs its work and then looks at the cross-workgroup error flag (xwg_check); if a hand-off
// inside the fused attention + Wo launch timed out, the SAME work runs again on one kernel per phase: the cache rows
// and logits of the failed attempt are simply overwritten, and the caller gets correct results and FLM_OK.
int flm_forward(flm_ctx* c, const int32_t* tokens, int n, int pos, float* logits_host) {
    if (!tokens || !logits_host) return FLM_ERR_INVALID;
    int r = check_ready(c, n, pos); if (r) return r;
    for (int attempt = 0; attempt < 2; ++attempt) {
        r = feed(c, tokens, n, pos, 0); if (r) return r;
        r = d2h(c, logits_host, c->logits, (size_t)c->d.vocab_size * 4); if (r) return r;
        r = xwg_check(c); if (r != FLM_RETRY) return r ? r : maybe_recover(c, n);
    }
    return fail(c, FLM_ERR_HIP, "cross-workgroup wait timed out twice");
}

int flm_forward_argmax(flm_ctx* c, const int32_t* tokens, int n, int pos, int32_t* next_token) {
    if (!tokens || !next_token) return FLM_ERR_INVALID;
    int r = check_ready(c, n, pos); if (r) return r;
    for (int attempt = 0; attempt < 2; ++attempt) {
        r = feed(c, tokens, n, pos, 1); if (r) return r;
        r = d2h(c, next_token, c->out_tokens_dev, 4); if (r) return r;
        r = xwg_check(c); if (r != FLM_RETRY) return r ? r : maybe_recover(c, n);
    }
    return fail(c, FLM_ERR_HIP, "cross-workgroup wait timed out twice");
}

static int decode_loop(flm_ctx* c, int32_t first_token, int pos, int n_steps, hipEvent_t e0, hipEvent_t e1) {
    int r = check_ready(c, n_steps, pos); if (r) return r;
    if (first_token < 0 || first_token >= c->d.vocab_size) return fail(c, FLM_ERR_INVALID, "token id out of range");
    if (n_steps > c->out_cap) return fail(c, FLM_ERR_INVALID, "more steps than max_seq_len");
    r = set_state(c, pos, first_token, 0); if (r) return r;
    if (e0) HIPC(c, hipEventRecord(e0, c->stream));
    r = run_greedy_tokens(c, pos, n_steps); if (r) return r;
    if (e1) HIPC(c, hipEventRecord(e1, c->stream));
    return FLM_OK;
}

int flm_decode_greedy(flm_ctx* c, int32_t first_token, int pos, int n_steps, int32_t* out_tokens) {
    if (!out_tokens) return FLM_ERR_INVALID;
    for (int attempt = 0; attempt < 2; ++attempt) {
        int r = decode_loop(c, first_token, pos, n_steps, nullptr, nullptr); if (r) return r;
        r = d2h(c, out_tokens, c->out_tokens_dev, sizeof(int) * (size_t)n_steps); if (r) return r;
        r = xwg_check(c); if (r != FLM_RETRY) return r ? r : maybe_recover(c, n_steps);
    }
    return fail(c, FLM_ERR_HIP, "cross-workgroup wait timed out twice");
}

int flm_decode_timed(flm_ctx* c, int32_t first_token, int pos, int n_steps, float* ms) {
    if (!ms || !c) return FLM_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    EvPair ev; HIPC(c, hipEventCreate(&ev.e0)); HIPC(c, hipEventCreate(&ev.e1));
    int r = FLM_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {
        r = decode_loop(c, first_token, pos, n_steps, ev.e0, ev.e1);
        if (!r) { hipError_t e = hipEventSynchronize(ev.e1); if (e == hipSuccess) e = hipEventElapsedTime(ms, ev.e0, ev.e1); if (e != hipSuccess) { c->err = hipGetErrorString(e); r = FLM_ERR_HIP; } }
        if (!r) r = xwg_check(c);
        if (r != FLM_RETRY) break;
    }
    if (r == FLM_OK) return maybe_recover(c, n_steps);
    return r == FLM_RETRY ? fail(c, FLM_ERR_HIP, "cross-workgroup wait timed out twice") : r;
}

// the ids the last flm_decode_greedy / flm_decode_timed* call generated (still in device memory): out[n]
int flm_last_tokens(flm_ctx* c, int n, int32_t* out) {
    if (!c || !out || n < 1 || n > c->out_cap) return FLM_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    return d2h(c, out, c->out_tokens_dev, sizeof(int) * (size_t)n);
}

// the same loop with an event after every token: ms_each[n_steps] (for a median; the events cost a few us per token, so the
// headline figure comes from flm_decode_timed)
int flm_decode_timed_each(flm_ctx* c, int32_t first_token, int pos, int n_steps, float* ms_each) {
    if (!ms_each || !c || n_steps < 1) return FLM_ERR_INVALID;
    int r = check_ready(c, n_steps, pos); if (r) return r;
    if (first_token < 0 || first_token >= c->d.vocab_size || n_steps > c->out_cap) return fail(c, FLM_ERR_INVALID, "token id / steps out of range");
    struct Evs { std::vector<hipEvent_t> e; ~Evs() { for (auto x : e) if (x) hipEventDestroy(x); } } ev;
    ev.e.assign((size_t)n_steps + 1, nullptr);
    for (auto& x : ev.e) HIPC(c, hipEventCreate(&x));
    r = set_state(c, pos, first_token, 0); if (r) return r;
    HIPC(c, hipEventRecord(ev.e[0], c->stream));
    for (int i = 0; i < n_steps; ++i) { r = run_token(c, true, 1, pos + i + 1); if (r) return r; HIPC(c, hipEventRecord(ev.e[i + 1], c->stream)); }
    HIPC(c, hipEventSynchronize(ev.e[n_steps]));
    for (int i = 0; i < n_steps; ++i) HIPC(c, hipEventElapsedTime(&ms_each[i], ev.e[i], ev.e[i + 1]));
    r = xwg_check(c);
    return r == FLM_RETRY ? fail(c, FLM_ERR_HIP, "cross-workgroup wait timed out while timing") : r;
}

// Per-class kernel time, measured live with HIP events on the ctx stream.  Single GPU: for each class the L launches
// of one token (layer 0 .. L-1, the real argument blocks) are enqueued back to back between ONE pair of events, so the
// average is launch duration + the dependent-dispatch gap and agrees with a rocprofv3 kernel trace; an event pair per
// launch would add ~5 us of marker latency to each.  The launches run out of token order, so the activations, the KV
// row at `pos` and the decode state are meaningless afterwards: call flm_reset_kv / feed a new prompt before decoding on.
// Tensor-parallel contexts time whole tokens with an event pair per launch (the collectives need token order).
} // extern "C"
