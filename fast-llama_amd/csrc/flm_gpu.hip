// flm_gpu.hip -- device context, weight upload, per-token forward and the C ABI (include/flm_gpu.h).
//
// Host-side orchestration of the kernels in flm_kernels.h.  What the reference does with 161
// fork-joins over pinned CPU threads per token (src/transformer/transformer.cpp:105-161)
// is here one stream of 5*L+3 kernel launches whose position/token operands live in device memory, so
// the whole token can be replayed from a hipGraph with no host round trip.
#include "flm_gpu.h"
#include "flm_kernels.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <mutex>
#include <vector>

using namespace flm;

namespace {

thread_local std::string g_last_error;

struct QMat { void* q = nullptr; float* s = nullptr; int rows = 0, cols = 0; float* st = nullptr; /* s group-major [cols / 64][rows]: the prompt path's GEMM tiles */ };
struct LayerW {
    QMat qkv, o, w13, w2;     // w13 = [W1 (gate) ; W3 (up)] back to back: the SwiGLU GEMV walks them as one matrix
    float* att_norm = nullptr; float* ffn_norm = nullptr;
    unsigned got = 0;     // bitmask of uploaded kinds
};

enum KClass { KC_EMBED = 0, KC_QKV, KC_ATTN, KC_ATTN_O, KC_FFN13, KC_FFN2, KC_CLS, KC_ARGMAX, KC_ALLREDUCE, KC_ATTN_WO /* k_attn_o: attention + Wo */, KC_FFN /* k_ffn: FFN13 + FFN2 */, KC_QKV_ATTN_WO /* k_qkv_attn_o */, KC_LAYER /* k_attn_ffn with the QKV GEMV in front: the whole layer */, KC_BACK /* k_attn_ffn: attention + Wo + FFN13 + FFN2 */ };

struct TimedLaunch { int kclass; hipEvent_t e0, e1; };
// owners that release on every exit path (the error macros return from the middle of a function)
struct DevMem { void* p = nullptr; ~DevMem() { if (p) hipFree(p); } };
struct EvPair { hipEvent_t e0 = nullptr, e1 = nullptr; ~EvPair() { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); } };

} // namespace

constexpr size_t kLdsMax = 160 * 1024;                            // LDS of a gfx950 CU = the most one workgroup can have
struct flm_ctx {
    flm_model_desc d{};
    int device = 0, rank = 0, world = 1;
    flm_shard_plan plan{};
    int hs = 0, esz = 1, cu_count = 256, cu_total = 256, n_xcd = 8;   // cu_count: CUs this context's launches are sized for (cu_total / cu_parts)
    int dim_local = 0, hidden_local = 0, heads_local = 0, vocab_slot = 0;    // dim_local = heads_local*hs: q/k/v rows and attention outputs owned
    int drow_begin = 0, drow_count = 0;                                       // rows of Wo / W2 (= slice of the residual stream) owned
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;

    std::vector<LayerW> layers;
    void* emb = nullptr; float* emb_s = nullptr; int emb_qt = 0; bool got_emb = false;
    float* out_norm = nullptr; bool got_out_norm = false;
    QMat cls; bool got_cls = false;

    float *kcache = nullptr, *vcache = nullptr;       // [L][heads_local][max_seq][hs]
    float *x1 = nullptr, *qbuf = nullptr, *att_out = nullptr, *hd = nullptr;
    float *logits = nullptr;
    float *rope_cos = nullptr, *rope_sin = nullptr;
    DecodeState* state = nullptr; int* prompt_dev = nullptr; int* out_tokens_dev = nullptr;
    int prompt_cap = 0, out_cap = 0;

    // options
    int wg_per_cu = 1; int use_graph = 1; int ablate = 0;
    int use_mfma = 1;                                  // option "use_mfma": int8 prefill GEMM tile shape on v_mfma_i32_32x32x32_i8: 1 by size, 2 (0) 64 x 64, 3 128 x 128
    int use_prefill = 1;                               // option "use_prefill": prompts of >= kPrefillMin+1 tokens go through the batched kernels
    int pf_cap = 0;                                    // token capacity of the batched-prefill buffers below
    bool pf_in_xbuf = false;                           // tensor parallel: pf_x / pf_att / pf_hd are regions of the exchange buffer (peers store into them)
    float *pf_x = nullptr, *pf_qkv = nullptr, *pf_q = nullptr, *pf_att = nullptr, *pf_gu = nullptr, *pf_hd = nullptr, *pf_xs = nullptr, *pf_xst = nullptr; void* pf_xq = nullptr;
    int use_prefill_mq = 1;                            // option "use_prefill_mq": batched prefill attention with 8 queries per workgroup (0: one query per workgroup)
    int use_pv_mfma = 1;                               // option "use_pv_mfma": prefill weighted sum (softmax x V) on the matrix cores as well (needs use_qk_mfma), 0: VALU chains
    int use_qk_mfma = 1;                               // option "use_qk_mfma": prefill scores on the matrix cores (fp32 MFMA, bit-identical), 0: VALU chains inside the attention kernel
    float* pf_scores = nullptr;                        // [heads][max_seq][max_seq] prefill scores (k_qk_mfma -> k_attn_prefill_mq<true>)
    int fold_xchg = 1;                                 // option "fold_xchg": tensor parallel, peer to peer: the exchanges' flag rounds inside the consuming GEMV launches
    int ranks_on_device = 1;                           // ranks of the group that live on this context's device (flm_p2p_import): folding needs a CU partition each
    int cu_parts = 1;                                  // option "cu_parts": the ctx's stream is confined to 1 / cu_parts of the device's CUs (rank % cu_parts picks which)
    bool tp_prefill = false;                           // tensor parallel: every rank of the group can (and will) feed prompts through the batched kernels
    int fuse_attn_o = 1;                               // option "fuse_attn_o": attention + Wo GEMV in one launch (k_attn_o; single GPU)
    bool st_ready = false;                             // the layer matrices' group-major scale copies (QMat::st) are up to date
    int fuse_ffn = 1;                                  // option "fuse_ffn": FFN13 + FFN2 in one launch (k_ffn; single GPU)
    int fuse_back = 1;                                 // option "fuse_back": attention + Wo + FFN13 + FFN2 in one launch with [W1; W3] stashed in LDS under the attention (k_attn_ffn; single GPU,
                                                       // head size a multiple of 64, one workgroup per head)
    int fuse_layer = 1;                                // option "fuse_layer": ... with the QKV GEMV in front: the whole layer in one launch
    int back_nst13 = -1, back_nst13_head = -1, back_nst2 = 0, back_pre13 = 16;   // options "back_*": k_attn_ffn's stash slots (-1: as many as the LDS holds) and early register set (flm_layer.h)
    int fuse_qkv = 1;                                  // option "fuse_qkv": QKV in front of attention + Wo in the same launch (k_qkv_attn_o; single GPU): 0 never,
                                                       // 1 when a head is spread over several workgroups (long contexts: where it pays), 2 always
    unsigned* flag_lines = nullptr; int* xwg_err = nullptr;   // k_attn_o: one 64-byte flag line per head; "a cross-workgroup wait timed out"
    void* att_q = nullptr; float* att_qs = nullptr;    // k_attn_o: the heads' output already quantized (head_size a multiple of 64)
    // tensor parallel, peer-to-peer: ONE exchange buffer per rank -- [att_out | x1 | hd | logits | flag lines] -- shared with the
    // peers (hipIpc); att_out / x1 / hd / logits point into it.  peer[r] = rank r's buffer mapped here (peer[rank] = xbuf).
    char* xbuf = nullptr; size_t xbuf_bytes = 0, x_flags_off = 0, x_hflags_off = 0; bool xbuf_fine = false;
    int tp_fuse_ffn = 0;                               // option "tp_fuse_ffn": the same for FFN13 + FFN2 (k_ffn across ranks: one line per rank, raised by the rank's last workgroup); off by
                                                       // default: on one GPU under CU masks it is slower at 2-4 ranks and faster at 8 (profiles/r03_tp_onegpu.txt) -- a multi-GPU box has to decide
    unsigned long long* ffn_counter = nullptr;         // (its device counter)
    int tp_fuse_attn = 2;                              // option "tp_fuse_attn": tensor parallel with folded exchanges: 1 = attention + Wo GEMV in one launch across the ranks (k_attn_o),
                                                       // 2 (default) = with the QKV GEMV in front (k_qkv_attn_o: its rows are the rank's own heads), 0 = separate launches
    char* peer[8] = {nullptr}; bool peer_opened[8] = {false}; int p2p = 0;
    unsigned* xepoch = nullptr;                        // [4] exchanges done per kind (att, x1, hd, logits), device memory
    float* att_sc = nullptr;                           // [heads_local][max_seq] scores exchanged between the parts of a split head
    int attn_split = 1;                                // option "attn_split": 1 = spread a head over 4 workgroups from kSplitFrom (128) positions on, 0 = never, >= 2 = always that many
    unsigned* eng_base = nullptr;                      // the token's epoch base (device memory, advanced by k_embed): the tensor-parallel exchanges' flag values count from it
    int resident = 1;                                  // the census at create saw every workgroup of a cu_count-wide launch co-resident
    int fell_back = 0;                                 // a cross-workgroup wait timed out once: fused launches off for good
    int trace_class = -1; unsigned long long* trace = nullptr;   // FLM_ABLATE builds: GEMV timeline of one kernel class
    std::map<int, hipGraphExec_t> graphs;             // key = with_cls*4 + advance
    std::vector<TimedLaunch>* timing = nullptr;
    std::string err;
};

namespace {

#define HIPC(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    if (ctx) (ctx)->err = b_; g_last_error = b_; return FLM_ERR_HIP; } } while (0)
#define NCCLC(ctx, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { \
    char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
    if (ctx) (ctx)->err = b_; g_last_error = b_; return FLM_ERR_COMM; } } while (0)

int fail(flm_ctx* c, int code, const char* msg) { if (c) c->err = msg; g_last_error = msg; return code; }

int esz_of(int qt) { return qt == FLM_QT_INT8 ? 1 : qt == FLM_QT_INT16 ? 2 : 4; }

// balanced contiguous split (split_rows, transformer.cpp:264-287)
void split_even(int total, int parts, int idx, int* begin, int* count) {
    const int itv = total / parts, rem = total % parts;
    if (idx < rem) { *begin = (itv + 1) * idx; *count = itv + 1; }
    else { *begin = (itv + 1) * rem + itv * (idx - rem); *count = itv; }
}

// ---------------------------------------------------------------------------------------------
// GEMV dispatch
// ---------------------------------------------------------------------------------------------
// Pass geometry of k_gemv for one launch (see the kernel's header comment).
//   cb_shift : CB = largest power of two <= 64 dividing K/16, so every 1 KiB wave load is full
//   Rm       : rows (per matrix) per workgroup pass; bounded by LDS (two strip buffers) and by 64 chain
//              lanes; chosen so that the passes divide evenly over `wgs` workgroups (CU-level balance is
//              what matters for an HBM-bound kernel; inside a workgroup the waves draw steps from a counter)
struct GemvPlan { int Rm, cb_shift, grid, nbuf; size_t lds; };
GemvPlan gemv_plan(int n, int esz, int rows, bool two, bool pairs, bool norm, int wgs) {
    GemvPlan P{};
    const int nchunks = n * esz / 16;
    int cbs = 0; while (cbs < 6 && (nchunks % (2 << cbs)) == 0) ++cbs;       // nchunks % 4 == 0 always
    const int RB = 64 >> cbs;
    const int mult = (pairs && RB < 2) ? 2 : RB;                             // ROPE_KV: row pairs stay in one pass
    const int lds_budget = 150 * 1024;                                       // one 1024-thread workgroup per CU out of 160 KiB
    int rmax = 64;                                                           // one chain lane per row
    while (rmax > mult && gemv_lds_layout(n, esz, norm, rmax, RB, two).total > lds_budget) rmax -= mult;
    rmax = rmax / mult * mult; if (rmax < mult) rmax = mult;
    if (rows < 1) rows = 1;
    int ppw = (rows + wgs * rmax - 1) / (wgs * rmax);                        // passes per workgroup
    if (ppw < 1) ppw = 1;
    int Rm = (rows + wgs * ppw - 1) / (wgs * ppw);
    Rm = (Rm + mult - 1) / mult * mult; if (Rm > rmax) Rm = rmax; if (Rm < mult) Rm = mult;
    const int npass = (rows + Rm - 1) / Rm;
    P.Rm = Rm; P.cb_shift = cbs; P.grid = npass < wgs ? npass : wgs; if (P.grid < 1) P.grid = 1;
    P.nbuf = 2;
    P.lds = (size_t)gemv_lds_layout(n, esz, norm, Rm, RB, two, P.nbuf).total;
    return P;
}

// fill the pass geometry of one GEMV into its argument block; returns the LDS bytes and grid it needs
template <int QT, int PRO, int EPI>
int plan_gemv(flm_ctx* c, GemvArgs& a, int wgs, GemvPlan& P) {
    constexpr bool TWO = EPI == EPI_SWIGLU, PAIRS = EPI == EPI_ROPE_KV;
    const int rows = a.items * (PAIRS ? 2 : 1);
    if ((double)rows * a.n * QTraits<QT>::kEsz * (TWO ? 2 : 1) >= 2147483648.0) return fail(c, FLM_ERR_UNSUPPORTED, "gemv: matrix of 2 GiB or more");
    P = gemv_plan(a.n, QTraits<QT>::kEsz, rows, TWO, PAIRS, true, wgs);
    if (P.lds > kLdsMax) return fail(c, FLM_ERR_UNSUPPORTED, "gemv: activation vector does not fit LDS");
    a.rows_per_pass = P.Rm; a.cb_shift = P.cb_shift; a.nbuf = P.nbuf;
    return FLM_OK;
}
template <int QT, int PRO, int EPI, bool COH>
int launch_gemv_xr(flm_ctx* c, hipStream_t st, GemvArgs a, int wgs) {
    GemvPlan P;
    int r = plan_gemv<QT, PRO, EPI>(c, a, wgs, P); if (r) return r;
    const int rounds = (a.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
    if (PRO == PRO_NONE)   hipLaunchKernelGGL((k_gemv<QT, PRO, EPI, 0, COH>), dim3(P.grid), dim3(kGemvBlock), P.lds, st, a);
    else if (rounds <= 1)  hipLaunchKernelGGL((k_gemv<QT, PRO, EPI, 1, COH>), dim3(P.grid), dim3(kGemvBlock), P.lds, st, a);
    else if (rounds <= 3)  hipLaunchKernelGGL((k_gemv<QT, PRO, EPI, 3, COH>), dim3(P.grid), dim3(kGemvBlock), P.lds, st, a);
    else                   hipLaunchKernelGGL((k_gemv<QT, PRO, EPI, 0, COH>), dim3(P.grid), dim3(kGemvBlock), P.lds, st, a);
    HIPC(c, hipGetLastError());
    return FLM_OK;
}
// coh: the activation a.x holds slices written by peer GPUs (tensor parallel, peer-to-peer) -> system-coherent loads
template <int PRO, int EPI>
int launch_gemv(flm_ctx* c, hipStream_t st, int qt, const GemvArgs& a, int wgs, bool coh = false) {
    if (a.n % kGroup != 0 || a.n <= 0) return fail(c, FLM_ERR_INVALID, "gemv: n must be a positive multiple of 64");
    if (qt != FLM_QT_INT8 && qt != FLM_QT_INT16) return fail(c, FLM_ERR_UNSUPPORTED, "gemv: quant type must be INT8 or INT16");
    if constexpr (PRO != PRO_NONE) {
        if (coh) return qt == FLM_QT_INT8 ? launch_gemv_xr<QT_INT8, PRO, EPI, true>(c, st, a, wgs) : launch_gemv_xr<QT_INT16, PRO, EPI, true>(c, st, a, wgs);
    }
    return qt == FLM_QT_INT8 ? launch_gemv_xr<QT_INT8, PRO, EPI, false>(c, st, a, wgs) : launch_gemv_xr<QT_INT16, PRO, EPI, false>(c, st, a, wgs);
}
// workgroups to spread a GEMV over: wg_per_cu per CU
int gemv_grid(int cu_count, int wg_per_cu, int /*items*/, int /*rows_per_item*/) { return cu_count * wg_per_cu; }

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
constexpr int FLM_RETRY = 1;
int xwg_check(flm_ctx* c) {
    if (!c->fuse_attn_o && !c->fuse_ffn && !c->fuse_back && c->attn_split == 0 && !c->p2p) return FLM_OK;
    // (on the context's own stream: a copy on the legacy stream synchronises with every blocking stream of the process -- and fails
    //  outright while another context's thread is capturing its token graph; seen once in ~10 runs of the threaded tensor-parallel tests)
    int e = 0;
    HIPC(c, hipMemcpyAsync(&e, c->xwg_err, 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
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
    c->fuse_attn_o = 0; c->fuse_ffn = 0; c->fuse_qkv = 0; c->fuse_back = 0; c->attn_split = 0; c->fell_back = 1;
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

struct Tick {
    flm_ctx* c; hipStream_t st; int kclass; hipEvent_t e0 = nullptr, e1 = nullptr;
    Tick(flm_ctx* c_, hipStream_t st_, int k) : c(c_), st(st_), kclass(k) {
        if (c->timing) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, st); }
    }
    ~Tick() { if (c->timing) { hipEventRecord(e1, st); c->timing->push_back({kclass, e0, e1}); } }
};

// ---------------------------------------------------------------------------------------------
// One token: ParallelTransformer::forward at bs == 1 (transformer.cpp:105-161).
// Position and token are read from c->state on the device.
//   with_cls  : run the final norm + classifier (+ argmax)
//   advance   : 1 = greedy (tok <- argmax, pos++), 0 = leave state (caller copies logits), 2 = prompt feed
// ---------------------------------------------------------------------------------------------
// peer-to-peer tensor parallelism: where `p` (a pointer into this rank's exchange buffer) lies in every peer's buffer
template <class A> void set_peers(flm_ctx* c, A& a, float* p) {
    a.n_peer = 0;
    if (!c->p2p) return;
    const size_t off = (char*)p - c->xbuf;
    for (int r = 0; r < c->world; ++r) if (r != c->rank) a.out_peer[a.n_peer++] = (float*)(c->peer[r] + off);
}
// argument blocks of the five GEMVs and the attention of layer l (shared by the per-phase launches and k_token)
GemvArgs args_qkv(flm_ctx* c, int l) {
    const auto& d = c->d; LayerW& w = c->layers[l];
    const size_t kv_layer = (size_t)c->heads_local * d.max_seq_len * c->hs;
    GemvArgs a{}; a.ablate = c->ablate;
    a.W = w.qkv.q; a.sW = w.qkv.s; a.n = d.dim; a.items = w.qkv.rows / 2;
    a.x = c->x1; a.norm_w = w.att_norm;
    a.out = c->qbuf; a.kcache = c->kcache + (size_t)l * kv_layer; a.vcache = c->vcache + (size_t)l * kv_layer;
    a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.pos_ptr = &c->state->pos;
    a.dim = c->dim_local; a.kv_dim = c->dim_local; a.max_seq = d.max_seq_len; a.hs = c->hs;
    return a;
}
// Parts per head for a token whose context is T positions: long contexts spread a head's K/V stream over 4 CUs (attn_head, G > 1).
// Below kSplitFrom the exchange of scores between the parts (one more cross-workgroup hand-off) costs more than it saves.
constexpr int kSplitFrom = 128;
int attn_parts(const flm_ctx* c, int T) {
    // every part owns kSplitDims = 32 output dimensions (its whole V slice then fits the registers / LDS of one workgroup)
    const int Gfull = c->hs / kSplitDims;
    const bool can = c->hs % kSplitDims == 0 && Gfull >= 2 && c->hs <= 128 && c->d.max_seq_len <= kSplitMaxSeq && c->heads_local * Gfull + 8 <= c->cu_count && c->heads_local * Gfull <= 256;
    if (!can || c->attn_split == 0) return 1;
    return (c->attn_split >= 2 || T >= kSplitFrom) ? Gfull : 1;
}
AttnArgs args_attn(flm_ctx* c, int l, int G = 1) {
    const auto& d = c->d;
    const size_t kv_layer = (size_t)c->heads_local * d.max_seq_len * c->hs;
    AttnArgs a{};
    a.q = c->qbuf; a.kcache = c->kcache + (size_t)l * kv_layer; a.vcache = c->vcache + (size_t)l * kv_layer;
    a.out = c->att_out + (size_t)c->plan.head_begin * c->hs; a.pos_ptr = &c->state->pos; a.hs = c->hs; a.max_seq = d.max_seq_len;
    a.G = G; a.sc_global = c->att_sc; a.flag_sc = c->flag_lines + 256 * 16; a.epoch = (unsigned)(l + 1); a.err = c->xwg_err;
    set_peers(c, a, a.out);
    return a;
}
GemvArgs args_o(flm_ctx* c, int l) {
    LayerW& w = c->layers[l];
    GemvArgs a{}; a.ablate = c->ablate;
    a.W = w.o.q; a.sW = w.o.s; a.n = c->d.dim; a.items = c->drow_count;
    a.x = c->att_out; a.out = c->x1 + c->drow_begin;
    set_peers(c, a, a.out);
    return a;
}
GemvArgs args_ffn13(flm_ctx* c, int l) {
    LayerW& w = c->layers[l];
    GemvArgs a{}; a.ablate = c->ablate;
    a.W = w.w13.q; a.sW = w.w13.s; a.n = c->d.dim; a.items = c->hidden_local;
    a.x = c->x1; a.norm_w = w.ffn_norm; a.out = c->hd + c->plan.hidden_begin;
    set_peers(c, a, a.out);
    return a;
}
GemvArgs args_ffn2(flm_ctx* c, int l) {
    LayerW& w = c->layers[l];
    GemvArgs a{}; a.ablate = c->ablate;
    a.W = w.w2.q; a.sW = w.w2.s; a.n = c->d.hidden_dim; a.items = c->drow_count;
    a.x = c->hd; a.out = c->x1 + c->drow_begin;
    set_peers(c, a, a.out);
    return a;
}
GemvArgs args_cls(flm_ctx* c) {
    GemvArgs a{}; a.ablate = c->ablate;
    a.W = c->cls.q; a.sW = c->cls.s; a.n = c->d.dim; a.items = c->cls.rows;
    a.x = c->x1; a.norm_w = c->out_norm; a.out = c->logits + (c->world > 1 ? (size_t)c->rank * c->vocab_slot : 0);
    set_peers(c, a, a.out);
    return a;
}


// tensor parallel, peer to peer: the consuming GEMV of exchange (layer l, kind) does the flag round itself (xchg_fold)
void set_fold(flm_ctx* c, GemvArgs& a, int l, int kind) {
    a.xf.world = 0;
    if (!(c->world > 1 && c->p2p && c->fold_xchg)) return;
    a.xf.local_flags = (unsigned*)(c->xbuf + c->x_flags_off);
    for (int r = 0; r < c->world; ++r) a.xf.peer_flags[r] = (unsigned*)(c->peer[r] + c->x_flags_off);
    a.xf.base = c->eng_base; a.xf.add = (unsigned)(4 * l + kind + 1);
    a.xf.rank = c->rank; a.xf.world = c->world; a.xf.slot = 4 + (kind == 3 ? 1 : kind); a.xf.err = c->xwg_err;      // kinds: 0 att, 1 x1 behind Wo, 2 hd, 3 x1 behind FFN2 (the x1 slot again)
}

// attention + Wo GEMV of layer l in one launch (k_attn_o); returns FLM_ERR_UNSUPPORTED when the shape does not allow it
template <int QT>
int launch_attn_o(flm_ctx* c, hipStream_t st, int l, int G) {
    const auto& d = c->d;
    const int parts = c->heads_local * G, wgs = c->cu_count - parts;
    if (wgs < 1 || parts > 256) return FLM_ERR_UNSUPPORTED;
    GemvArgs a = args_o(c, l);
    if (kAblate && c->trace_class == 101 && l == 0) a.trace = c->trace;     // tools/trace_ao.py
    GemvPlan P;
    int r = plan_gemv<QT, PRO_QUANT, EPI_RESIDUAL>(c, a, wgs, P); if (r) return r;
    const int rounds = (a.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
    if (rounds > 3) return FLM_ERR_UNSUPPORTED;
    size_t lds = attn_lds_bytes(d.max_seq_len, c->hs, G > 1); if (P.lds > lds) lds = P.lds;
    AttnArgs aa = args_attn(c, l, G);
    unsigned* flag = c->flag_lines;                              // one 64-byte line per head part, value = layer + 1; k_embed clears them at the start of the token
    const dim3 grid(parts + P.grid), block(kGemvBlock);
    AoTp tp{};
    if (c->world > 1) {
        // across ranks: every rank's head parts raise their lines in every rank's array (in the exchange buffer); the Wo workgroups read the full att vector
        // from this rank's exchange region (the heads' stores went to every rank) with coherent loads and quantize it themselves
        if (d.n_heads * G > 256) return FLM_ERR_UNSUPPORTED;
        tp.world = c->world; tp.line0 = c->plan.head_begin * G; tp.n_lines = d.n_heads * G; tp.base = c->eng_base; tp.add = (unsigned)(4 * l + 1);
        for (int r = 0; r < c->world; ++r) tp.peer_flags[r] = (unsigned*)(c->peer[r] + c->x_hflags_off);
        flag = (unsigned*)(c->xbuf + c->x_hflags_off);
        if (G > 1) {
            if (rounds <= 1) hipLaunchKernelGGL((k_attn_o<QT, 1, false, true>), grid, block, lds, st, aa, a, parts, flag, 0u, c->xwg_err, tp);
            else             hipLaunchKernelGGL((k_attn_o<QT, 3, false, true>), grid, block, lds, st, aa, a, parts, flag, 0u, c->xwg_err, tp);
        }
        else if (rounds <= 1) hipLaunchKernelGGL((k_attn_o<QT, 1, false>), grid, block, lds, st, aa, a, parts, flag, 0u, c->xwg_err, tp);
        else                  hipLaunchKernelGGL((k_attn_o<QT, 3, false>), grid, block, lds, st, aa, a, parts, flag, 0u, c->xwg_err, tp);
        HIPC(c, hipGetLastError());
        return FLM_OK;
    }
    if (c->hs % kGroup == 0 && G == 1) {
        // a head's output is whole quant groups: the head workgroups quantize it themselves (A3 on the 64 values a wave
        // holds), the GEMV workgroups fetch 1 (2) bytes per element and skip the quantize prologue
        aa.oq = c->att_q; aa.os = c->att_qs; aa.oqt = QT;
        a.xq = c->att_q; a.xs = c->att_qs;
        hipLaunchKernelGGL((k_attn_o<QT, 0, true>), grid, block, lds, st, aa, a, parts, flag, (unsigned)(l + 1), c->xwg_err, tp);
    }
    else if (G > 1) {
        if (rounds <= 1) hipLaunchKernelGGL((k_attn_o<QT, 1, false, true>), grid, block, lds, st, aa, a, parts, flag, (unsigned)(l + 1), c->xwg_err, tp);
        else             hipLaunchKernelGGL((k_attn_o<QT, 3, false, true>), grid, block, lds, st, aa, a, parts, flag, (unsigned)(l + 1), c->xwg_err, tp);
    }
    else if (rounds <= 1) hipLaunchKernelGGL((k_attn_o<QT, 1, false>), grid, block, lds, st, aa, a, parts, flag, (unsigned)(l + 1), c->xwg_err, tp);
    else                  hipLaunchKernelGGL((k_attn_o<QT, 3, false>), grid, block, lds, st, aa, a, parts, flag, (unsigned)(l + 1), c->xwg_err, tp);
    HIPC(c, hipGetLastError());
    return FLM_OK;
}

// QKV + attention + Wo GEMV of layer l in one launch (k_qkv_attn_o); returns FLM_ERR_UNSUPPORTED when the shape does not allow it
template <int QT>
int launch_qkv_attn_o(flm_ctx* c, hipStream_t st, int l, int G) {
    const auto& d = c->d;
    const int parts = c->heads_local * G, all = c->cu_count < 256 ? c->cu_count : 256, wgs = all - parts;
    if (wgs < 1 || parts > 256) return FLM_ERR_UNSUPPORTED;
    GemvArgs aq = args_qkv(c, l), a = args_o(c, l);
    GemvPlan Pq, P;
    int r = plan_gemv<QT, PRO_RMSNORM_QUANT, EPI_ROPE_KV>(c, aq, all, Pq); if (r) return r;
    r = plan_gemv<QT, PRO_QUANT, EPI_RESIDUAL>(c, a, wgs, P); if (r) return r;
    const int rq = (aq.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4), rounds = (a.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
    if (rq > 1 || rounds > 3) return FLM_ERR_UNSUPPORTED;
    size_t lds = attn_lds_bytes(d.max_seq_len, c->hs, G > 1); if (P.lds > lds) lds = P.lds; if (Pq.lds > lds) lds = Pq.lds;
    AttnArgs aa = args_attn(c, l, G);
    unsigned* flag = c->flag_lines;                              // heads' lines (as k_attn_o)
    unsigned* flagq = c->flag_lines + 768 * 16;                  // the QKV workgroups' lines; value = layer + 1, cleared by k_embed
    const int gridx = parts + P.grid > Pq.grid ? parts + P.grid : Pq.grid;
    const dim3 grid(gridx), block(kGemvBlock);
    const unsigned tgt = (unsigned)(l + 1);
    AoTp tp{};
    if (c->world > 1) {
        // across ranks (see launch_attn_o); the QKV phase consumes the x1 exchange behind the previous layer's FFN2 (kind 3 of layer l - 1; layer 0 reads the embedding)
        if (d.n_heads * G > 256) return FLM_ERR_UNSUPPORTED;
        if (l > 0) set_fold(c, aq, l - 1, 3);
        tp.world = c->world; tp.line0 = c->plan.head_begin * G; tp.n_lines = d.n_heads * G; tp.base = c->eng_base; tp.add = (unsigned)(4 * l + 1);
        for (int r = 0; r < c->world; ++r) tp.peer_flags[r] = (unsigned*)(c->peer[r] + c->x_hflags_off);
        flag = (unsigned*)(c->xbuf + c->x_hflags_off);
        if (G > 1) {
            if (rounds <= 1) hipLaunchKernelGGL((k_qkv_attn_o<QT, 1, false, true, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
            else             hipLaunchKernelGGL((k_qkv_attn_o<QT, 3, false, true, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
        }
        else if (rounds <= 1) hipLaunchKernelGGL((k_qkv_attn_o<QT, 1, false, false, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
        else                  hipLaunchKernelGGL((k_qkv_attn_o<QT, 3, false, false, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
        HIPC(c, hipGetLastError());
        return FLM_OK;
    }
    if (c->hs % kGroup == 0 && G == 1) {
        aa.oq = c->att_q; aa.os = c->att_qs; aa.oqt = QT;
        a.xq = c->att_q; a.xs = c->att_qs;
        hipLaunchKernelGGL((k_qkv_attn_o<QT, 0, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
    }
    else if (G > 1) {
        if (rounds <= 1) hipLaunchKernelGGL((k_qkv_attn_o<QT, 1, false, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
        else             hipLaunchKernelGGL((k_qkv_attn_o<QT, 3, false, true>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
    }
    else if (rounds <= 1) hipLaunchKernelGGL((k_qkv_attn_o<QT, 1, false>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
    else                  hipLaunchKernelGGL((k_qkv_attn_o<QT, 3, false>), grid, block, lds, st, aq, aa, a, Pq.grid, parts, P.grid, flagq, flag, tgt, c->xwg_err, tp);
    HIPC(c, hipGetLastError());
    return FLM_OK;
}

// FFN13 + FFN2 of layer l in one launch (k_ffn); returns FLM_ERR_UNSUPPORTED when the shape does not allow it
template <int QT>
int launch_ffn(flm_ctx* c, hipStream_t st, int l) {
    GemvArgs a13 = args_ffn13(c, l), a2 = args_ffn2(c, l);
    const int wgs = c->cu_count < 256 ? c->cu_count : 256;       // every workgroup resident, one flag line each
    GemvPlan P13, P2;
    int r = plan_gemv<QT, PRO_RMSNORM_QUANT, EPI_SWIGLU>(c, a13, wgs, P13); if (r) return r;
    r = plan_gemv<QT, PRO_QUANT, EPI_RESIDUAL>(c, a2, wgs, P2); if (r) return r;
    const int r13 = (a13.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4), r2 = (a2.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
    if (r13 > 1 || r2 > 3) return FLM_ERR_UNSUPPORTED;
    const size_t lds = P13.lds > P2.lds ? P13.lds : P2.lds;
    const int grid = P13.grid > P2.grid ? P13.grid : P2.grid;
    unsigned* flag = c->flag_lines + 512 * 16;                    // value = layer + 1; k_embed clears the lines at the start of the token
    FfnTp tp{};
    if (c->world > 1) {
        // across ranks: FFN13 consumes the x1 exchange behind the Wo launch (folded flag round, kind 1); one line per RANK for hd (in the exchange buffer, behind the head lines)
        set_fold(c, a13, l, 1);
        tp.world = c->world; tp.rank = c->rank; tp.base = c->eng_base; tp.add = (unsigned)(4 * l + 3); tp.counter = c->ffn_counter;
        for (int r = 0; r < c->world; ++r) tp.peer_flags[r] = (unsigned*)(c->peer[r] + c->x_hflags_off) + 256 * 16;
        flag = (unsigned*)(c->xbuf + c->x_hflags_off) + 256 * 16;
        if (r2 <= 1) hipLaunchKernelGGL((k_ffn<QT, 1, true>), dim3(grid), dim3(kGemvBlock), lds, st, a13, a2, P13.grid, P2.grid, flag, 0u, c->xwg_err, tp);
        else         hipLaunchKernelGGL((k_ffn<QT, 3, true>), dim3(grid), dim3(kGemvBlock), lds, st, a13, a2, P13.grid, P2.grid, flag, 0u, c->xwg_err, tp);
        HIPC(c, hipGetLastError());
        return FLM_OK;
    }
    if (r2 <= 1) hipLaunchKernelGGL((k_ffn<QT, 1>), dim3(grid), dim3(kGemvBlock), lds, st, a13, a2, P13.grid, P2.grid, flag, (unsigned)(l + 1), c->xwg_err, tp);
    else         hipLaunchKernelGGL((k_ffn<QT, 3>), dim3(grid), dim3(kGemvBlock), lds, st, a13, a2, P13.grid, P2.grid, flag, (unsigned)(l + 1), c->xwg_err, tp);
    HIPC(c, hipGetLastError());
    return FLM_OK;
}


// attention + Wo + FFN13 + FFN2 of layer l in one launch (k_attn_ffn, flm_layer.h); returns FLM_ERR_UNSUPPORTED when the shape does not allow it
template <int QT>
int launch_attn_ffn(flm_ctx* c, hipStream_t st, int l, bool with_qkv) {
    const auto& d = c->d;
    constexpr int esz = QTraits<QT>::kEsz;
    const int all = c->cu_count < 256 ? c->cu_count : 256, parts = c->heads_local, wgs_o = all - parts;
    if (c->world != 1 || c->hs % kGroup != 0 || wgs_o < 1 || parts > 256) return FLM_ERR_UNSUPPORTED;
    GemvArgs aq = args_qkv(c, l), ao = args_o(c, l), a13 = args_ffn13(c, l), a2 = args_ffn2(c, l);
    GemvPlan Pq{}, Po, P13, P2;
    int r = plan_gemv<QT, PRO_QUANT, EPI_RESIDUAL>(c, ao, wgs_o, Po); if (r) return r;
    if (with_qkv) {
        r = plan_gemv<QT, PRO_RMSNORM_QUANT, EPI_ROPE_KV>(c, aq, all, Pq); if (r) return r;
        if ((aq.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4) > 1) return FLM_ERR_UNSUPPORTED;
    }
    r = plan_gemv<QT, PRO_RMSNORM_QUANT, EPI_SWIGLU>(c, a13, all, P13); if (r) return r;
    r = plan_gemv<QT, PRO_QUANT, EPI_RESIDUAL>(c, a2, all, P2); if (r) return r;
    const int r13 = (a13.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4), r2 = (a2.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
    if (r13 > 1 || r2 > 3) return FLM_ERR_UNSUPPORTED;
    // a workgroup with a single pass needs one strip buffer: the LDS above the phases' own layouts is the stash
    auto one_pass = [&](GemvArgs& a, GemvPlan& P, bool two, int rows_per_item = 1) {
        const int rows = a.items * rows_per_item, npass = (rows + P.Rm - 1) / P.Rm;
        if (npass <= P.grid) { a.nbuf = 1; P.nbuf = 1; P.lds = (size_t)gemv_lds_layout(a.n, esz, true, P.Rm, 64 >> P.cb_shift, two, 1).total; }
    };
    one_pass(ao, Po, false); one_pass(a13, P13, true); one_pass(a2, P2, false);
    size_t own = Po.lds; if (P13.lds > own) own = P13.lds; if (P2.lds > own) own = P2.lds;
    if (with_qkv) { one_pass(aq, Pq, false, 2); if (Pq.lds > kLdsMax) return FLM_ERR_UNSUPPORTED; }     // (the QKV phase is over before the first stash request: its layout may overlap the slots)
    own = (own + 255) & ~(size_t)255;
    const size_t lds_attn = attn_lds_bytes(d.max_seq_len, c->hs, false);
    if (own > kLdsMax || lds_attn > kLdsMax) return FLM_ERR_UNSUPPORTED;
    const int slot = kStepBlk * 1024 + 256, fit = (int)((kLdsMax - own) / slot);
    auto slots = [&](int want) { int n = want < 0 ? fit : want; if (n > fit) n = fit; if (n > 32) n = 32; return n < 0 ? 0 : n; };
    AttnArgs aa = args_attn(c, l, 1);
    aa.oq = c->att_q; aa.os = c->att_qs; aa.oqt = QT;
    ao.xq = c->att_q; ao.xs = c->att_qs;
    BackArgs p{};
    p.n_heads = parts; p.grido = Po.grid; p.grid13 = P13.grid; p.grid2 = P2.grid;
    p.flag_h = c->flag_lines; p.flag_hd = c->flag_lines + 512 * 16; p.flag_x = c->flag_lines + 1024 * 16;
    p.gridq = with_qkv ? Pq.grid : 0; p.flag_q = c->flag_lines + 768 * 16;
    p.target = (unsigned)(l + 1); p.err = c->xwg_err;
    p.st_base = (unsigned)own; p.nst13 = slots(c->back_nst13); p.nst13_head = slots(c->back_nst13_head); p.nst2 = slots(c->back_nst2); p.pre13 = c->back_pre13 < 0 ? 0 : c->back_pre13 > 16 ? 16 : c->back_pre13;
    if (kAblate && c->trace_class == 102 && l == 0) { p.trace = c->trace; a13.trace = c->trace + 256 * 16; a2.trace = c->trace + 2 * 256 * 16; }   // tools/trace_back.py
    int grid = parts + Po.grid; if (P13.grid > grid) grid = P13.grid; if (P2.grid > grid) grid = P2.grid; if (with_qkv && Pq.grid > grid) grid = Pq.grid;
    if (grid > all) return FLM_ERR_UNSUPPORTED;
    {   // the stash takes the rest of the CU's 160 KiB: raise the kernels' dynamic-LDS limit, once per device
        static std::mutex mu; static bool done[64] = {false};
        std::lock_guard<std::mutex> lk(mu);
        if (c->device >= 0 && c->device < 64 && !done[c->device]) {
            const void* fns[] = {(const void*)&k_attn_ffn<QT_INT8, 1, false>, (const void*)&k_attn_ffn<QT_INT8, 3, false>, (const void*)&k_attn_ffn<QT_INT16, 1, false>, (const void*)&k_attn_ffn<QT_INT16, 3, false>,
                                 (const void*)&k_attn_ffn<QT_INT8, 1, true>, (const void*)&k_attn_ffn<QT_INT8, 3, true>, (const void*)&k_attn_ffn<QT_INT16, 1, true>, (const void*)&k_attn_ffn<QT_INT16, 3, true>};
            for (const void* f : fns) HIPC(c, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
            done[c->device] = true;
        }
    }
    const dim3 g3(grid), b3(kGemvBlock);
    if (with_qkv) {
        if (r2 <= 1) hipLaunchKernelGGL((k_attn_ffn<QT, 1, true>), g3, b3, kLdsMax, st, aq, aa, ao, a13, a2, p);
        else         hipLaunchKernelGGL((k_attn_ffn<QT, 3, true>), g3, b3, kLdsMax, st, aq, aa, ao, a13, a2, p);
    }
    else if (r2 <= 1) hipLaunchKernelGGL((k_attn_ffn<QT, 1, false>), g3, b3, kLdsMax, st, aq, aa, ao, a13, a2, p);
    else              hipLaunchKernelGGL((k_attn_ffn<QT, 3, false>), g3, b3, kLdsMax, st, aq, aa, ao, a13, a2, p);
    HIPC(c, hipGetLastError());
    return FLM_OK;
}


// one activation exchange between the tensor-parallel ranks (the reference's threads share the vector in memory instead):
// peer-to-peer (the producer already stored its slice everywhere: flag round only) or an RCCL all-gather
enum XKind { XK_ATT = 0, XK_X1 = 1, XK_HD = 2, XK_LOGITS = 3 };
int exchange(flm_ctx* c, hipStream_t st, int kind, float* full, float* mine, int count) {
    Tick t(c, st, KC_ALLREDUCE);
    if (c->p2p) {
        XchgArgs x{};
        x.local_flags = (unsigned*)(c->xbuf + c->x_flags_off);
        for (int r = 0; r < c->world; ++r) x.peer_flags[r] = (unsigned*)(c->peer[r] + c->x_flags_off);
        x.epoch = c->xepoch + kind; x.err = c->xwg_err; x.rank = c->rank; x.world = c->world; x.kind = kind;
        hipLaunchKernelGGL(k_xchg, dim3(1), dim3(64), 0, st, x);
        HIPC(c, hipGetLastError());
        return FLM_OK;
    }
    if (!c->comm) return fail(c, FLM_ERR_STATE, "tensor parallel: neither flm_p2p_import was called nor an RCCL id was given");
    NCCLC(c, ncclAllGather(mine, full, count, ncclFloat, c->comm, st));
    return FLM_OK;
}

int enqueue_token(flm_ctx* c, hipStream_t st, bool with_cls, int advance, int G) {
    const auto& d = c->d;
    const int qt = d.quant_type, hs = c->hs, L = d.n_layers;
    const bool tp = c->world > 1 || c->comm != nullptr, coh = tp && c->p2p;
    {
        Tick t(c, st, KC_EMBED);
        hipLaunchKernelGGL(k_embed, dim3((d.dim + 255) / 256), dim3(256), 0, st, c->x1, (const void*)c->emb, (const float*)c->emb_s, c->emb_qt, d.dim, (const int*)&c->state->tok, c->flag_lines, c->eng_base);
        HIPC(c, hipGetLastError());
    }
    const int wgs = gemv_grid(c->cu_count, c->wg_per_cu, 0, 0);
    auto traced = [&](GemvArgs a, int kc, int l) { if (kAblate && c->trace_class == kc && l == 0) a.trace = c->trace; return a; };
    int r;
    // tensor parallel, peer to peer: the flag rounds of the att / x1 / hd exchanges happen inside the launches that consume them (xchg_fold), not in
    // launches of their own; what stays a k_xchg is the logits' exchange and, for a token without classifier, the last x1 exchange (the next token's
    // k_embed rewrites x1: every peer's stores into it must have landed first)
    // (ranks sharing a device need a CU partition each -- "cu_parts" -- or a consumer that fills the device while it polls keeps its peers' producers out)
    const bool fold = tp && c->p2p && c->fold_xchg && c->world > 1 && c->cu_parts >= c->ranks_on_device;
    auto folded = [&](GemvArgs a, int l, int kind) { if (fold && l >= 0) set_fold(c, a, l, kind); return a; };
    // launches that span the ranks wait across workgroups of one launch too: only where the census found one workgroup per CU resident (a CU partition
    // made for the tests is sized for it: launches are cut to the partition)
    const bool span = fold && (c->resident || c->cu_parts > 1);
    for (int l = 0; l < L; ++l) {
        bool fused = false;
        const bool back_ok = !tp && c->fuse_back && c->fuse_attn_o && c->fuse_ffn && G == 1 && !c->timing && (c->trace_class < 0 || c->trace_class == 102);
        if (back_ok && c->fuse_layer) {   // the whole layer in one launch
            r = qt == FLM_QT_INT8 ? launch_attn_ffn<QT_INT8>(c, st, l, true) : launch_attn_ffn<QT_INT16>(c, st, l, true);
            if (r == FLM_OK) continue; else if (r != FLM_ERR_UNSUPPORTED) return r;
        }
        if (((!tp && c->fuse_attn_o && (c->fuse_qkv >= 2 || (c->fuse_qkv && G > 1))) || (span && c->tp_fuse_attn >= 2)) && !c->timing && c->trace_class < 0) {   // QKV + attention + ATTN_O in one launch (tensor parallel: "tp_fuse_attn" 2)
            r = qt == FLM_QT_INT8 ? launch_qkv_attn_o<QT_INT8>(c, st, l, G) : launch_qkv_attn_o<QT_INT16>(c, st, l, G);
            if (r == FLM_OK) fused = true; else if (r != FLM_ERR_UNSUPPORTED) return r;
        }
        if (!fused) {   // QKV task + RoPE + KV append (transformer.cpp:132-135, execute_qkv :386-395, execute_attn :431-439): this rank's heads
            Tick t(c, st, KC_QKV);
            r = launch_gemv<PRO_RMSNORM_QUANT, EPI_ROPE_KV>(c, st, qt, folded(traced(args_qkv(c, l), KC_QKV, l), l - 1, 3), wgs, coh); if (r) return r;
        }
        if (!fused && back_ok) {   // attention + ATTN_O + FFN13 + FFN2 in one launch
            r = qt == FLM_QT_INT8 ? launch_attn_ffn<QT_INT8>(c, st, l, false) : launch_attn_ffn<QT_INT16>(c, st, l, false);
            if (r == FLM_OK) continue; else if (r != FLM_ERR_UNSUPPORTED) return r;
        }
        if (!fused && ((!tp && c->fuse_attn_o) || (span && c->tp_fuse_attn)) && !c->timing && (c->trace_class < 0 || c->trace_class == 101)) {   // attention + ATTN_O in one launch (tensor parallel: across the ranks)
            r = qt == FLM_QT_INT8 ? launch_attn_o<QT_INT8>(c, st, l, G) : launch_attn_o<QT_INT16>(c, st, l, G);
            if (r == FLM_OK) fused = true; else if (r != FLM_ERR_UNSUPPORTED) return r;
        }
        if (!fused) {   // ATTN task (execute_attn :441-449): local heads write their slice of the full att_out vector (on every rank, peer to peer)
            Tick t(c, st, KC_ATTN);
            AttnArgs aa = args_attn(c, l, G); if (kAblate && c->trace_class == KC_ATTN && l == 0) aa.trace = c->trace;
            if (G > 1) hipLaunchKernelGGL(k_attn_decode<true>, dim3(c->heads_local * G), dim3(kAttnBlock), attn_lds_bytes(d.max_seq_len, hs, true), st, aa);
            else       hipLaunchKernelGGL(k_attn_decode<false>, dim3(c->heads_local), dim3(kAttnBlock), attn_lds_bytes(d.max_seq_len, hs, false), st, aa);
            HIPC(c, hipGetLastError());
        }
        // every rank needs all heads' outputs: the reference's threads share x2 in memory (transformer.cpp:451-454)
        if (tp && !fold) { r = exchange(c, st, XK_ATT, c->att_out, c->att_out + (size_t)c->plan.head_begin * hs, c->dim_local); if (r) return r; }
        if (!fused) {   // ATTN_O task + residual (transformer.cpp:138-139, execute_attn_o :457-466): this rank's rows of Wo
            Tick t(c, st, KC_ATTN_O);
            r = launch_gemv<PRO_QUANT, EPI_RESIDUAL>(c, st, qt, folded(traced(args_o(c, l), KC_ATTN_O, l), l, 0), wgs, coh); if (r) return r;
        }
        if (tp && !fold) { r = exchange(c, st, XK_X1, c->x1, c->x1 + c->drow_begin, c->drow_count); if (r) return r; }
        if (((!tp && c->fuse_ffn) || (span && c->tp_fuse_ffn)) && !c->timing && c->trace_class < 0) {   // FFN13 + FFN2 in one launch (tensor parallel: across the ranks)
            r = qt == FLM_QT_INT8 ? launch_ffn<QT_INT8>(c, st, l) : launch_ffn<QT_INT16>(c, st, l);
            if (r == FLM_OK) {
                if (tp && l == L - 1 && !with_cls) { r = exchange(c, st, XK_X1, c->x1, c->x1 + c->drow_begin, c->drow_count); if (r) return r; }   // (see below)
                continue;
            } else if (r != FLM_ERR_UNSUPPORTED) return r;
        }
        {   // FFN13 task + SwiGLU (transformer.cpp:144-147, execute_ffn13 :468-483): this rank's rows of W1/W3
            Tick t(c, st, KC_FFN13);
            r = launch_gemv<PRO_RMSNORM_QUANT, EPI_SWIGLU>(c, st, qt, folded(traced(args_ffn13(c, l), KC_FFN13, l), l, 1), wgs, coh); if (r) return r;
        }
        if (tp && !fold) { r = exchange(c, st, XK_HD, c->hd, c->hd + c->plan.hidden_begin, c->hidden_local); if (r) return r; }
        {   // FFN2 task + residual (transformer.cpp:149-150, execute_ffn2 :485-494): this rank's rows of W2
            Tick t(c, st, KC_FFN2);
            r = launch_gemv<PRO_QUANT, EPI_RESIDUAL>(c, st, qt, folded(traced(args_ffn2(c, l), KC_FFN2, l), l, 2), wgs, coh); if (r) return r;
        }
        if (tp && (!fold || (l == L - 1 && !with_cls))) { r = exchange(c, st, XK_X1, c->x1, c->x1 + c->drow_begin, c->drow_count); if (r) return r; }
    }
    if (with_cls) {
        {   // final norm + CLS task (transformer.cpp:154-160, execute_cls :496-505): this rank's rows of the classifier
            Tick t(c, st, KC_CLS);
            r = launch_gemv<PRO_RMSNORM_QUANT, EPI_STORE>(c, st, qt, folded(traced(args_cls(c), KC_CLS, 0), L - 1, 3), wgs, coh); if (r) return r;
        }
        if (tp) { r = exchange(c, st, XK_LOGITS, c->logits, c->logits + (size_t)c->rank * c->vocab_slot, c->vocab_slot); if (r) return r; }
        if (advance != 0) {
            Tick t(c, st, KC_ARGMAX);
            hipLaunchKernelGGL(k_argmax_advance, dim3(1), dim3(1024), 0, st, (const float*)c->logits, d.vocab_size, c->state, c->out_tokens_dev, 1, c->out_cap);
            HIPC(c, hipGetLastError());
        }
    } else if (advance == 2) {
        hipLaunchKernelGGL(k_advance_prompt, dim3(1), dim3(64), 0, st, c->state, (const int*)c->prompt_dev);
        HIPC(c, hipGetLastError());
    }
    return FLM_OK;
}

// run one token, through a cached hipGraph when enabled.  T = positions the token's attention covers (known to the host:
// it picks how many workgroups a head is spread over; the graphs are keyed by it)
int run_token(flm_ctx* c, bool with_cls, int advance, int T) {
    const int G = attn_parts(c, T);
    if (!c->use_graph || c->timing || ((c->world > 1 || c->comm) && !c->p2p)) return enqueue_token(c, c->stream, with_cls, advance, G);   // (RCCL collectives stay eager)
    const int key = (with_cls ? 4 : 0) + advance + 8 * G;
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        HIPC(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        int r = enqueue_token(c, c->stream, with_cls, advance, G);
        hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (r) { if (g) hipGraphDestroy(g); return r; }
        HIPC(c, e);
        HIPC(c, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        HIPC(c, hipGraphDestroy(g));
        it = c->graphs.emplace(key, ge).first;
    }
    HIPC(c, hipGraphLaunch(it->second, c->stream));
    return FLM_OK;
}

// Everything a forward needs is allocated at flm_ctx_create ("zero allocations during inference", reference README and
// transformer.cpp:110-130): the prompt / output id buffers and the batched-prefill activations are sized by max_seq_len.
int alloc_run_bufs(flm_ctx* c) {
    const auto& d = c->d;
    c->prompt_cap = d.max_seq_len; c->out_cap = d.max_seq_len;
    HIPC(c, hipMalloc((void**)&c->prompt_dev, sizeof(int) * c->prompt_cap));
    HIPC(c, hipMalloc((void**)&c->out_tokens_dev, sizeof(int) * c->out_cap));
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
    if (!model_complete(c)) return fail(c, FLM_ERR_STATE, "forward before all tensors were uploaded");
    if (n < 1 || pos < 0 || pos + n > c->d.max_seq_len) return fail(c, FLM_ERR_INVALID, "tokens/pos outside [0, max_seq_len]");
    HIPC(c, hipSetDevice(c->device));
    return FLM_OK;
}

// ---------------------------------------------------------------------------------------------
// Batched prefill of B prompt tokens at positions pos .. pos+B-1 (single GPU): leaves their K/V rows in the cache, exactly
// the rows the token-by-token path would write (flm_kernels.h, "Batched prefill").  The prompt's LAST token is not part
// of the batch: it runs through the decode kernels and produces the logits.
// ---------------------------------------------------------------------------------------------
constexpr int kPrefillMin = 4;

template <int QT, int PRO>
int launch_rows(flm_ctx* c, hipStream_t st, const RowsArgs& r, int B, bool coh = false) {
    const size_t lds = (size_t)gemv_lds_layout(r.n, QTraits<QT>::kEsz, true, 4, 4, false).total;
    const int rounds = (r.n + kGemvBlock * 4 - 1) / (kGemvBlock * 4);
    if (coh) {   // the rows lie in the exchange buffer and were partly written by peer GPUs
        if (rounds <= 1)      hipLaunchKernelGGL((k_rows_prologue<QT, PRO, 1, true>), dim3(B), dim3(kGemvBlock), lds, st, r);
        else if (rounds <= 3) hipLaunchKernelGGL((k_rows_prologue<QT, PRO, 3, true>), dim3(B), dim3(kGemvBlock), lds, st, r);
        else                  hipLaunchKernelGGL((k_rows_prologue<QT, PRO, 0, true>), dim3(B), dim3(kGemvBlock), lds, st, r);
    }
    else if (rounds <= 1) hipLaunchKernelGGL((k_rows_prologue<QT, PRO, 1>), dim3(B), dim3(kGemvBlock), lds, st, r);
    else if (rounds <= 3) hipLaunchKernelGGL((k_rows_prologue<QT, PRO, 3>), dim3(B), dim3(kGemvBlock), lds, st, r);
    else                  hipLaunchKernelGGL((k_rows_prologue<QT, PRO, 0>), dim3(B), dim3(kGemvBlock), lds, st, r);
    HIPC(c, hipGetLastError());
    return FLM_OK;
}
// use_mfma (int8): 1 tile shape by size, 2 (and 0) always 64 x 64, 3 always 128 x 128 matrix-core tiles; int16: always the hi / lo byte planes on the int8 matrix cores
template <int QT, int EPI>
int launch_gemm(flm_ctx* c, hipStream_t st, const GemmArgs& g, int use_mfma) {
    const int tiles = ((g.rows + 63) / 64) * ((g.B + 63) / 64);
    if (use_mfma == 0) use_mfma = 2;
    if (QT == QT_INT8) {   // exact int32 group dots on v_mfma_i32_32x32x32_i8
        using Big = GemmTile<4, 2, 2>; using Small = GemmTile<2, 2, 1>;
        {   // the 128 x 128 tiles stage 76 KiB of LDS: raise the kernels' dynamic-LDS limit, once per device
            static std::mutex mu; static bool done[64] = {false};
            int dev = 0; HIPC(c, hipGetDevice(&dev));
            std::lock_guard<std::mutex> lk(mu);
            if (dev >= 0 && dev < 64 && !done[dev]) {
                HIPC(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_q8_mfma<EPI_STORE, 4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, Big::kLds));
                HIPC(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_q8_mfma<EPI_RESIDUAL, 4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, Big::kLds));
                HIPC(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_q8_mfma<EPI_SWIGLU, 4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, Big::kLds));
                HIPC(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_q8_mfma<EPI_ROPE_KV, 4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, Big::kLds));
                done[dev] = true;
            }
        }
        if constexpr (EPI == EPI_SWIGLU) {   // g.rows = hidden: a tile = 64 rows of W1 and of W3 (gemm_fuses_swiglu decides)
            const int tiles = ((g.rows + Big::TR / 2 - 1) / (Big::TR / 2)) * ((g.B + Big::TT - 1) / Big::TT);
            hipLaunchKernelGGL((k_gemm_q8_mfma<EPI_SWIGLU, 4, 2, 2>), dim3(tiles), dim3(Big::NT), Big::kLds, st, g);
            HIPC(c, hipGetLastError());
            return FLM_OK;
        } else {
        const int tiles128 = ((g.rows + Big::TR - 1) / Big::TR) * ((g.B + Big::TT - 1) / Big::TT);
        // 128 x 128 tiles move 0.6x the LDS cycles and half the bytes per product; they pay once every CU has one (measured, 7B width:
        // 512 tokens qkv / ffn13 100.8 vs 115.3 us, Wo / ffn2 (128 tiles) 95.4 vs 65.4; 1000 tokens 171 vs 221 and 110 vs 117)
        if (use_mfma == 3 || (use_mfma == 1 && tiles128 >= 256)) hipLaunchKernelGGL((k_gemm_q8_mfma<EPI, 4, 2, 2>), dim3(tiles128), dim3(Big::NT), Big::kLds, st, g);
        else hipLaunchKernelGGL((k_gemm_q8_mfma<EPI, 2, 2, 1>), dim3(tiles), dim3(Small::NT), Small::kLds, st, g);
        }
    }
    else if constexpr (EPI == EPI_SWIGLU || EPI == EPI_ROPE_KV) return fail(c, FLM_ERR_INVALID, "launch_gemm: the SwiGLU / RoPE epilogues exist for the int8 matrix-core tiles only");
    else hipLaunchKernelGGL((k_gemm_q16_mfma<EPI>), dim3(tiles), dim3(256), Gemm16Tile::kLds, st, g);   // hi / lo byte planes on the int8 matrix cores
    HIPC(c, hipGetLastError());
    return FLM_OK;
}

template <int QT>
int prefill_batched(flm_ctx* c, int B, int pos) {
    const auto& d = c->d;
    const int L = d.n_layers, dim = d.dim, hid = d.hidden_dim, hs = c->hs;
    hipStream_t st = c->stream;
    int r = B <= c->pf_cap ? FLM_OK : fail(c, FLM_ERR_INVALID, "prefill: more tokens than max_seq_len"); if (r) return r;
    const size_t kv_layer = (size_t)c->heads_local * d.max_seq_len * hs;
    if (!c->st_ready) {   // once per set of weights: the scales group-major (no allocation: the copies' memory came with the matrices)
        for (auto& w : c->layers)
            for (QMat* m : {&w.qkv, &w.o, &w.w13, &w.w2}) {
                const size_t n = (size_t)m->rows * (m->cols / kGroup);
                hipLaunchKernelGGL(k_transpose_scales, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)m->s, m->st, m->rows, m->cols / kGroup);
            }
        HIPC(c, hipGetLastError());
        c->st_ready = true;
    }
    // Tensor parallel (peer-to-peer; the matrix-core kernels): this rank's heads / rows / hidden slice of every step, as in the decode path
    // (split_rows, transformer.cpp:264-287); the attention output, the residual stream and hd are full-width in every rank's exchange
    // region: the kernel that produces a column slice stores it into all of them, a flag round (k_xchg) closes the step, the row
    // prologues read with coherent loads.  Single GPU: dimL == dim, slices = everything, no peers.
    const bool tp = c->world > 1;
    const int dimL = c->dim_local, hidL = c->hidden_local, col_o = c->drow_begin, rows_o = c->drow_count, col_h = c->plan.hidden_begin, col_a = c->plan.head_begin * hs;
    auto peers = [&](GemmArgs& g, float* p) { g.n_peer = 0; if (tp) { const size_t off = (char*)p - c->xbuf; for (int r2 = 0; r2 < c->world; ++r2) if (r2 != c->rank) g.out_peer[g.n_peer++] = (float*)(c->peer[r2] + off); } };
    hipLaunchKernelGGL(k_embed_rows, dim3(B), dim3(256), 0, st, c->pf_x, (const void*)c->emb, (const float*)c->emb_s, c->emb_qt, dim, (const int*)c->prompt_dev);
    HIPC(c, hipGetLastError());
    for (int l = 0; l < L; ++l) {
        LayerW& w = c->layers[l];
        // x2 = rmsnorm(x1); qx = quantize(x2); q,k,v = W x; RoPE; cache rows   (transformer.cpp:132-135, 386-395, 431-439)
        RowsArgs ra{c->pf_x, w.att_norm, c->pf_xq, c->pf_xs, dim, c->pf_xst};
        r = launch_rows<QT, PRO_RMSNORM_QUANT>(c, st, ra, B, tp); if (r) return r;
        GemmArgs g{w.qkv.q, w.qkv.s, c->pf_xq, c->pf_xs, c->pf_qkv, 3 * dimL, dim, 3 * dimL, B, c->pf_xst, w.qkv.st};
        if (QT == QT_INT8 && c->use_mfma && dimL % 32 == 0 && hs % 2 == 0) {
            // RoPE and the cache rows as the epilogue of the matrix-core tiles: no [tokens][3 dim] round trip, no k_rope_kv_rows
            g.qout = c->pf_q; g.kcache = c->kcache + (size_t)l * kv_layer; g.vcache = c->vcache + (size_t)l * kv_layer;
            g.rope_cos = c->rope_cos; g.rope_sin = c->rope_sin; g.dim = dimL; g.hs = hs; g.max_seq = d.max_seq_len; g.pos0 = pos;
            r = launch_gemm<QT, EPI_ROPE_KV>(c, st, g, c->use_mfma); if (r) return r;
        } else {
            r = launch_gemm<QT, EPI_STORE>(c, st, g, c->use_mfma); if (r) return r;
            hipLaunchKernelGGL(k_rope_kv_rows, dim3(B), dim3(256), 0, st, (const float*)c->pf_qkv, c->pf_q, c->kcache + (size_t)l * kv_layer, c->vcache + (size_t)l * kv_layer,
                               (const float*)c->rope_cos, (const float*)c->rope_sin, dimL, hs, d.max_seq_len, pos);
            HIPC(c, hipGetLastError());
        }
        if (l == L - 1) break;                            // the batch only has to fill the cache: nothing downstream of the last layer's K/V is needed
        // attention of every query over the cache rows 0 .. its own position   (execute_attn :441-449): the local heads' columns of att
        AttnArgs aa{}; aa.q = c->pf_q; aa.kcache = c->kcache + (size_t)l * kv_layer; aa.vcache = c->vcache + (size_t)l * kv_layer;
        aa.out = c->pf_att + col_a; aa.pos_ptr = &c->state->pos; aa.hs = hs; aa.max_seq = d.max_seq_len;
        if (tp) { const size_t off = (char*)aa.out - c->xbuf; for (int r2 = 0; r2 < c->world; ++r2) if (r2 != c->rank) aa.out_peer[aa.n_peer++] = (float*)(c->peer[r2] + off); }
        // which kernels: the exps of a tile of queries (weighted sum on the matrix cores) or the scores of 8 queries (VALU) must fit the LDS;
        // one query per workgroup needs 4 bytes per position and always fits (flm_ctx_create checked max_seq_len against it)
        const bool mq_fits = attn_mq_lds_bytes(d.max_seq_len, hs) <= kLdsMax;
        const bool pv_mfma = c->use_pv_mfma && (hs & 1) == 0;
        if (hs <= 128 && c->use_prefill_mq && c->use_qk_mfma && c->pf_scores && (pv_mfma || mq_fits)) {
            // scores on the matrix cores (fp32 MFMA = the reference's chains, bit for bit), then softmax + weighted sum per tile of queries
            aa.sc_global = c->pf_scores;
            const dim3 gq(c->heads_local, (B + kQkQ - 1) / kQkQ);
            switch (hs >> 5) {
            case 1: hipLaunchKernelGGL(k_qk_mfma<1>, gq, dim3(256), 0, st, aa, pos, dimL, B); break;
            case 2: hipLaunchKernelGGL(k_qk_mfma<2>, gq, dim3(256), 0, st, aa, pos, dimL, B); break;
            case 3: hipLaunchKernelGGL(k_qk_mfma<3>, gq, dim3(256), 0, st, aa, pos, dimL, B); break;
            default: hipLaunchKernelGGL(k_qk_mfma<4>, gq, dim3(256), 0, st, aa, pos, dimL, B); break;
            }
            HIPC(c, hipGetLastError());
            if (pv_mfma) {   // ... and the weighted sum too (an accumulator element = the reference's chain of one (query, dimension))
                const int qw = pv_mfma_queries(pos + B, kLdsMax);     // 16 queries per workgroup up to ~2500 positions, fewer beyond
                hipLaunchKernelGGL(k_attn_pv_mfma, dim3(c->heads_local, (B + qw - 1) / qw), dim3(256), pv_mfma_lds_bytes(pos + B, qw), st, aa, pos, dim, B, qw);
            } else
                hipLaunchKernelGGL(k_attn_prefill_mq<true>, dim3(c->heads_local, (B + kMqQueries - 1) / kMqQueries), dim3(kAttnBlock), attn_mq_lds_bytes(d.max_seq_len, hs), st, aa, pos, dim, B);
        }
        else if (hs <= 128 && c->use_prefill_mq && mq_fits)      // kMqQueries queries per workgroup share every K/V tile
            hipLaunchKernelGGL(k_attn_prefill_mq<false>, dim3(c->heads_local, (B + kMqQueries - 1) / kMqQueries), dim3(kAttnBlock), attn_mq_lds_bytes(d.max_seq_len, hs), st, aa, pos, dim, B);
        else
            hipLaunchKernelGGL(k_attn_prefill, dim3(c->heads_local, B), dim3(kAttnBlock), attn_lds_bytes(d.max_seq_len, hs), st, aa, pos, dim);
        HIPC(c, hipGetLastError());
        if (tp) { r = exchange(c, st, XK_ATT, nullptr, nullptr, 0); if (r) return r; }
        // x1 += Wo quantize(att)   (transformer.cpp:138-139, 457-466): this rank's rows of Wo = its columns of x1
        RowsArgs rq{c->pf_att, nullptr, c->pf_xq, c->pf_xs, dim, c->pf_xst};
        r = launch_rows<QT, PRO_QUANT>(c, st, rq, B, tp); if (r) return r;
        GemmArgs go{w.o.q, w.o.s, c->pf_xq, c->pf_xs, c->pf_x + col_o, dim, dim, rows_o, B, c->pf_xst, w.o.st};
        peers(go, go.out);
        r = launch_gemm<QT, EPI_RESIDUAL>(c, st, go, c->use_mfma); if (r) return r;
        if (tp) { r = exchange(c, st, XK_X1, nullptr, nullptr, 0); if (r) return r; }
        // hd = swiglu(W1 qx, W3 qx) with qx = quantize(rmsnorm(x1))   (transformer.cpp:144-147, 468-483): this rank's slice of hd
        RowsArgs rf{c->pf_x, w.ffn_norm, c->pf_xq, c->pf_xs, dim, c->pf_xst};
        r = launch_rows<QT, PRO_RMSNORM_QUANT>(c, st, rf, B, tp); if (r) return r;
        if (QT == QT_INT8 && c->use_mfma && (tp || c->use_mfma == 3 || (c->use_mfma == 1 && ((hidL + 63) / 64) * ((B + 127) / 128) >= 256))) {
            // 128 x 128 tiles of 64 gate + 64 up rows: the GEMM's epilogue is the SwiGLU
            GemmArgs g13{w.w13.q, w.w13.s, c->pf_xq, c->pf_xs, c->pf_hd + col_h, hid, dim, hidL, B, c->pf_xst, w.w13.st};
            peers(g13, g13.out);
            r = launch_gemm<QT, EPI_SWIGLU>(c, st, g13, c->use_mfma); if (r) return r;
        } else {
            GemmArgs g13{w.w13.q, w.w13.s, c->pf_xq, c->pf_xs, c->pf_gu, 2 * hidL, dim, 2 * hidL, B, c->pf_xst, w.w13.st};
            r = launch_gemm<QT, EPI_STORE>(c, st, g13, c->use_mfma); if (r) return r;
            SwigluPeers sp{}; { GemmArgs t{}; peers(t, c->pf_hd + col_h); sp.n = t.n_peer; for (int i = 0; i < t.n_peer; ++i) sp.p[i] = t.out_peer[i]; }
            hipLaunchKernelGGL(k_swiglu_rows, dim3(B), dim3(256), 0, st, c->pf_hd + col_h, (const float*)c->pf_gu, hidL, hid, sp);
            HIPC(c, hipGetLastError());
        }
        if (tp) { r = exchange(c, st, XK_HD, nullptr, nullptr, 0); if (r) return r; }
        // x1 += W2 quantize(hd)   (transformer.cpp:149-150, 485-494)
        RowsArgs rh{c->pf_hd, nullptr, c->pf_xq, c->pf_xs, hid, c->pf_xst};
        r = launch_rows<QT, PRO_QUANT>(c, st, rh, B, tp); if (r) return r;
        GemmArgs g2{w.w2.q, w.w2.s, c->pf_xq, c->pf_xs, c->pf_x + col_o, dim, hid, rows_o, B, c->pf_xst, w.w2.st};
        peers(g2, g2.out);
        r = launch_gemm<QT, EPI_RESIDUAL>(c, st, g2, c->use_mfma); if (r) return r;
        if (tp) { r = exchange(c, st, XK_X1, nullptr, nullptr, 0); if (r) return r; }
    }
    return FLM_OK;
}

// can this rank run the batched prompt path under tensor parallelism?  (rank-local: a failed score-buffer allocation, options)
bool tp_prefill_capable(const flm_ctx* c) {
    return c->pf_in_xbuf && c->use_mfma && c->use_qk_mfma && c->use_pv_mfma && c->use_prefill_mq &&
           c->pf_scores && c->hs <= 128 && c->hs % 2 == 0 && c->dim_local % 32 == 0;
}
// feed tokens[0..n) sequentially (row i of the reference's batched prefill depends only on rows
// <= i through the KV cache, so token-by-token evaluation performs the same per-row arithmetic).
int feed(flm_ctx* c, const int32_t* tokens, int n, int pos, int final_advance) {
    int r;
    if (n > c->prompt_cap) return fail(c, FLM_ERR_INVALID, "more tokens than max_seq_len");
    for (int i = 0; i < n; ++i) if (tokens[i] < 0 || tokens[i] >= c->d.vocab_size) return fail(c, FLM_ERR_INVALID, "token id out of range");
    HIPC(c, hipMemcpyAsync(c->prompt_dev, tokens, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));   // (the caller's buffer outlives the call: every entry point synchronises)
    // batched: single GPU always; tensor parallel over the peer-to-peer exchange with the matrix-core kernels (the kernels that store their
    // column slices into the peers' buffers)
    const bool tp_ok = c->world > 1 && c->p2p && c->tp_prefill;          // agreed by all ranks at flm_p2p_import
    if (c->use_prefill && (c->world == 1 || tp_ok) && n - 1 >= kPrefillMin) {
        // all tokens but the last in one batch (cache rows only), then the last one through the decode kernels
        r = c->d.quant_type == FLM_QT_INT8 ? prefill_batched<QT_INT8>(c, n - 1, pos) : prefill_batched<QT_INT16>(c, n - 1, pos);
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

} // namespace

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, n ? n : 4) == hipSuccess ? 0 : 1; }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};
#define OPC(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_last_error = std::string(#expr " failed: ") + hipGetErrorString(e_); return FLM_ERR_HIP; } } while (0)
}

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
    const size_t kvn = (size_t)L * c->heads_local * d.max_seq_len * hs;
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
        const size_t total = world > 1 ? up(o_ph + pcap * d.hidden_dim * 4) : o_fl + (kXchgSlots * 8 + 1) * 64;      // (flags: + the abort line)
        hipError_t ae = hipErrorUnknown;
        if (world > 1) { ae = hipExtMallocWithFlags((void**)&c->xbuf, total, hipDeviceMallocFinegrained); c->xbuf_fine = ae == hipSuccess; }   // written by peer GPUs
        if (ae != hipSuccess) { (void)hipGetLastError(); HIPB(hipMalloc((void**)&c->xbuf, total)); }
        c->xbuf_bytes = total; c->x_flags_off = o_fl; c->x_hflags_off = o_hf;
        HIPB(hipMemsetAsync(c->xbuf, 0, total, c->stream));
        c->att_out = (float*)(c->xbuf + o_att); c->x1 = (float*)(c->xbuf + o_x1); c->hd = (float*)(c->xbuf + o_hd); c->logits = (float*)(c->xbuf + o_lg);
        c->peer[rank] = c->xbuf;
        if (world > 1) { c->pf_x = (float*)(c->xbuf + o_px); c->pf_att = (float*)(c->xbuf + o_pa); c->pf_hd = (float*)(c->xbuf + o_ph); c->pf_in_xbuf = true; }
        HIPB(hipMalloc((void**)&c->xepoch, 64)); HIPB(hipMemsetAsync(c->xepoch, 0, 64, c->stream));
        HIPB(hipMalloc((void**)&c->ffn_counter, 64)); HIPB(hipMemsetAsync(c->ffn_counter, 0, 64, c->stream));
    }
    HIPB(hipMalloc((void**)&c->flag_lines, 1536 * 64)); HIPB(hipMalloc((void**)&c->xwg_err, 64));   // lines 0..255: k_attn_o's heads, 256..511: split heads' scores, 512..767: k_ffn, 768..1023: k_qkv_attn_o's QKV rows, 1024..1279: k_attn_ffn's x1 rows (k_embed clears all 1536)
    HIPB(hipMemsetAsync(c->flag_lines, 0, 1536 * 64, c->stream)); HIPB(hipMemsetAsync(c->xwg_err, 0, 64, c->stream));
    HIPB(hipMalloc((void**)&c->eng_base, 64)); HIPB(hipMemsetAsync(c->eng_base, 0, 64, c->stream));   // the token's epoch base
    HIPB(hipMalloc(&c->att_q, (size_t)d.dim * c->esz)); HIPB(hipMalloc((void**)&c->att_qs, (size_t)(d.dim / kGroup) * 4));
    HIPB(hipMalloc((void**)&c->att_sc, (size_t)c->heads_local * d.max_seq_len * 4));
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
        unsigned* cnt = (unsigned*)((char*)c->xwg_err + 32); int* okp = c->xwg_err + 4;
        int one = 1;
        HIPB(hipMemsetAsync(cnt, 0, 4, c->stream)); HIPB(hipMemcpyAsync(okp, &one, 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_census, dim3(c->cu_count), dim3(1024), 150 * 1024, c->stream, cnt, (unsigned)c->cu_count, okp);
        int ok = 0;
        if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&ok, okp, 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess) c->resident = ok ? 1 : 0;
        else { (void)hipGetLastError(); c->resident = 0; }
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
                    c->flag_lines, c->xwg_err, c->att_q, c->att_qs, c->att_sc, c->trace, c->eng_base, c->ffn_counter,
                    c->pf_in_xbuf ? nullptr : c->pf_x, c->pf_qkv, c->pf_q, c->pf_in_xbuf ? nullptr : c->pf_att, c->pf_gu, c->pf_in_xbuf ? nullptr : c->pf_hd, c->pf_xs, c->pf_xq, c->pf_scores};
    for (void* p : ptrs) if (p) hipFree(p);
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
    b.caps = tp_prefill_capable(c) ? 1 : 0;
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
    c->tp_prefill = true;
    for (int r = 0; r < n; ++r) if (!(b[r].caps & 1)) c->tp_prefill = false;
    c->ranks_on_device = 0;
    for (int r = 0; r < n; ++r) if (b[r].device == c->device) ++c->ranks_on_device;
    if (!tp_prefill_capable(c)) c->tp_prefill = false;
    for (int r = 0; r < n; ++r) {
        if (b[r].magic != kP2pMagic || b[r].rank != r || b[r].world != c->world || b[r].bytes != c->xbuf_bytes) return fail(c, FLM_ERR_INVALID, "p2p_import: blobs are not those of this tensor-parallel group, in rank order");
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
    return FLM_OK;
}

int flm_set_option(flm_ctx* c, const char* key, int value) {
    if (!c || !key) return FLM_ERR_INVALID;
    std::string k(key);
    if (c->world > 1 && c->p2p && (k == "use_mfma" || k == "use_pv_mfma" || k == "use_prefill_mq" || k == "use_qk_mfma"))
        return fail(c, FLM_ERR_STATE, "set_option: which prompt kernels a tensor-parallel group runs is agreed at flm_p2p_import; set this option on every rank before importing (\"use_prefill\" may be switched later, on every rank alike)");
    if (!c->resident && value != 0 && (k == "fuse_attn_o" || k == "fuse_ffn" || k == "fuse_qkv" || k == "fuse_back" || k == "attn_split"))
        return fail(c, FLM_ERR_UNSUPPORTED, "set_option: this device does not keep one workgroup per CU resident (census at flm_ctx_create); the fused launches stay off");
    if (k == "wg_per_cu") { c->wg_per_cu = value > 0 ? value : 1; }
    else if (k == "use_graph") c->use_graph = value;
    else if (k == "use_prefill") c->use_prefill = value;
    else if (k == "use_mfma") c->use_mfma = value;
    else if (k == "use_pv_mfma") c->use_pv_mfma = value;
    else if (k == "fuse_attn_o") c->fuse_attn_o = value;
    else if (k == "fuse_ffn") c->fuse_ffn = value;
    else if (k == "fuse_qkv") c->fuse_qkv = value;
    else if (k == "fuse_back") c->fuse_back = value;
    else if (k == "fuse_layer") c->fuse_layer = value;
    else if (k == "back_nst13") c->back_nst13 = value;
    else if (k == "back_nst13_head") c->back_nst13_head = value;
    else if (k == "back_nst2") c->back_nst2 = value;
    else if (k == "back_pre13") c->back_pre13 = value;
    else if (k == "use_prefill_mq") c->use_prefill_mq = value;
    else if (k == "attn_split") c->attn_split = value;
    else if (k == "fold_xchg") c->fold_xchg = value;
    else if (k == "tp_fuse_attn") c->tp_fuse_attn = value;
    else if (k == "tp_fuse_ffn") c->tp_fuse_ffn = value;
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
    return FLM_OK;
}


int flm_query(flm_ctx* c, const char* key, int* value) {
    if (!c || !key || !value) return FLM_ERR_INVALID;
    const std::string k(key);
    const struct { const char* k; int v; } tab[] = {
        {"wg_per_cu", c->wg_per_cu}, {"use_graph", c->use_graph}, {"use_prefill", c->use_prefill}, {"use_mfma", c->use_mfma}, {"use_pv_mfma", c->use_pv_mfma},
        {"fuse_attn_o", c->fuse_attn_o}, {"fuse_ffn", c->fuse_ffn}, {"fuse_qkv", c->fuse_qkv}, {"fuse_back", c->fuse_back}, {"fuse_layer", c->fuse_layer}, {"back_nst13", c->back_nst13}, {"back_nst13_head", c->back_nst13_head}, {"back_nst2", c->back_nst2}, {"back_pre13", c->back_pre13}, {"use_prefill_mq", c->use_prefill_mq}, {"attn_split", c->attn_split},
        {"use_qk_mfma", c->use_qk_mfma}, {"use_p2p", c->p2p}, {"fold_xchg", c->fold_xchg}, {"tp_fuse_attn", c->tp_fuse_attn}, {"tp_fuse_ffn", c->tp_fuse_ffn}, {"cu_parts", c->cu_parts}, {"fold_active", (c->world > 1 && c->p2p && c->fold_xchg && c->cu_parts >= c->ranks_on_device) ? 1 : 0}, {"resident", c->resident}, {"fallback", c->fell_back},
        {"token_path", (c->world == 1 ? ((c->fuse_attn_o ? 1 : 0) | (c->fuse_ffn ? 2 : 0) | (c->fuse_attn_o && c->fuse_qkv == 1 ? 4 : 0) | (c->fuse_attn_o && c->fuse_qkv >= 2 ? 8 : 0) | (c->fuse_back && c->fuse_attn_o && c->fuse_ffn ? (c->fuse_layer ? 128 + 256 : 128) : 0)) : 0) | (c->attn_split ? 64 : 0)},
    };
    for (const auto& t : tab) if (k == t.k) { *value = t.v; return FLM_OK; }
    return fail(c, FLM_ERR_INVALID, "query: unknown key");
}

int flm_upload_tensor(flm_ctx* c, int kind, int layer, int src_qt, const void* values, const float* scales, int rows, int cols) {
    if (c) c->st_ready = false;
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

int flm_reset_kv(flm_ctx* c) {
    if (!c) return FLM_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    const size_t kvn = (size_t)c->d.n_layers * c->heads_local * c->d.max_seq_len * c->hs;
    HIPC(c, hipMemsetAsync(c->kcache, 0, kvn * 4, c->stream)); HIPC(c, hipMemsetAsync(c->vcache, 0, kvn * 4, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return FLM_OK;
}

// debugging tap (tests): copy an internal device buffer to the host. what: 0 x1, 1 q, 2 att_out, 3 hd, 4 kcache(layer), 5 vcache(layer), 6 logits
int flm_debug_read(flm_ctx* c, int what, int layer, float* out, size_t n) {
    if (!c || !out) return FLM_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    const size_t kvl = (size_t)c->heads_local * c->d.max_seq_len * c->hs;
    const float* src = nullptr; size_t cap = 0;
    switch (what) {
    case 0: src = c->x1; cap = c->d.dim; break;
    case 1: src = c->qbuf; cap = c->dim_local; break;
    case 2: src = c->att_out; cap = c->d.dim; break;
    case 3: src = c->hd; cap = c->d.hidden_dim; break;
    case 4: src = c->kcache + (size_t)layer * kvl; cap = kvl; break;
    case 5: src = c->vcache + (size_t)layer * kvl; cap = kvl; break;
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
        for (size_t i = nrt; i < n; ++i) { const size_t b = i - i % 16; out[i] = i % 16 == 15 ? (float)t[i] : ((t[i] && t[b]) ? (float)(long long)(t[i] - t[b]) : -1.f); }
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
        HIPC(c, hipMemcpyAsync(logits_host, c->logits, (size_t)c->d.vocab_size * 4, hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        r = xwg_check(c); if (r != FLM_RETRY) return r;
    }
    return fail(c, FLM_ERR_HIP, "cross-workgroup wait timed out twice");
}

int flm_forward_argmax(flm_ctx* c, const int32_t* tokens, int n, int pos, int32_t* next_token) {
    if (!tokens || !next_token) return FLM_ERR_INVALID;
    int r = check_ready(c, n, pos); if (r) return r;
    for (int attempt = 0; attempt < 2; ++attempt) {
        r = feed(c, tokens, n, pos, 1); if (r) return r;
        HIPC(c, hipMemcpyAsync(next_token, c->out_tokens_dev, 4, hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        r = xwg_check(c); if (r != FLM_RETRY) return r;
    }
    return fail(c, FLM_ERR_HIP, "cross-workgroup wait timed out twice");
}

static int decode_loop(flm_ctx* c, int32_t first_token, int pos, int n_steps, hipEvent_t e0, hipEvent_t e1) {
    int r = check_ready(c, n_steps, pos); if (r) return r;
    if (first_token < 0 || first_token >= c->d.vocab_size) return fail(c, FLM_ERR_INVALID, "token id out of range");
    if (n_steps > c->out_cap) return fail(c, FLM_ERR_INVALID, "more steps than max_seq_len");
    r = set_state(c, pos, first_token, 0); if (r) return r;
    if (e0) HIPC(c, hipEventRecord(e0, c->stream));
    for (int i = 0; i < n_steps; ++i) { r = run_token(c, true, 1, pos + i + 1); if (r) return r; }
    if (e1) HIPC(c, hipEventRecord(e1, c->stream));
    return FLM_OK;
}

int flm_decode_greedy(flm_ctx* c, int32_t first_token, int pos, int n_steps, int32_t* out_tokens) {
    if (!out_tokens) return FLM_ERR_INVALID;
    for (int attempt = 0; attempt < 2; ++attempt) {
        int r = decode_loop(c, first_token, pos, n_steps, nullptr, nullptr); if (r) return r;
        HIPC(c, hipMemcpyAsync(out_tokens, c->out_tokens_dev, sizeof(int) * n_steps, hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        r = xwg_check(c); if (r != FLM_RETRY) return r;
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
    return r == FLM_RETRY ? fail(c, FLM_ERR_HIP, "cross-workgroup wait timed out twice") : r;
}

// the ids the last flm_decode_greedy / flm_decode_timed* call generated (still in device memory): out[n]
int flm_last_tokens(flm_ctx* c, int n, int32_t* out) {
    if (!c || !out || n < 1 || n > c->out_cap) return FLM_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemcpyAsync(out, c->out_tokens_dev, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return FLM_OK;
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
int flm_kernel_times(flm_ctx* c, int pos, int iters, float* avg_us, int32_t* count) {
    if (!avg_us || !count || iters < 1) return FLM_ERR_INVALID;
    int r = check_ready(c, 1, pos); if (r) return r;
    double tot[FLM_KCLASSES] = {0}; long cnt[FLM_KCLASSES] = {0};
    if (c->world > 1) {
        for (int it = 0; it < iters + 1; ++it) {
            r = set_state(c, pos, 1 % c->d.vocab_size, 0); if (r) return r;
            std::vector<TimedLaunch> tl; c->timing = &tl;
            r = enqueue_token(c, c->stream, true, 1, attn_parts(c, pos + 1));
            c->timing = nullptr;
            hipStreamSynchronize(c->stream);
            for (auto& t : tl) {
                float ms = 0.f;
                if (!r && it > 0 && hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) { tot[t.kclass] += ms * 1000.0; cnt[t.kclass] += 1; }
                hipEventDestroy(t.e0); hipEventDestroy(t.e1);
            }
            if (r) return r;
        }
        for (int k = 0; k < FLM_KCLASSES; ++k) { avg_us[k] = cnt[k] ? (float)(tot[k] / cnt[k]) : 0.f; count[k] = (int32_t)(cnt[k] / iters); }
        return FLM_OK;
    }
    const auto& d = c->d;
    const int qt = d.quant_type, L = d.n_layers, wgs = gemv_grid(c->cu_count, c->wg_per_cu, 0, 0);
    hipStream_t st = c->stream;
    r = set_state(c, pos, 1 % d.vocab_size, 0); if (r) return r;
    EvPair ev; HIPC(c, hipEventCreate(&ev.e0)); HIPC(c, hipEventCreate(&ev.e1));
    const hipEvent_t e0 = ev.e0, e1 = ev.e1;
    auto launch = [&](int kc, int l) -> int {
        switch (kc) {
        case KC_EMBED:  hipLaunchKernelGGL(k_embed, dim3((d.dim + 255) / 256), dim3(256), 0, st, c->x1, (const void*)c->emb, (const float*)c->emb_s, c->emb_qt, d.dim, (const int*)&c->state->tok, c->flag_lines, c->eng_base); return FLM_OK;
        case KC_QKV:    return launch_gemv<PRO_RMSNORM_QUANT, EPI_ROPE_KV>(c, st, qt, args_qkv(c, l), wgs);
        case KC_ATTN:   { const int G = attn_parts(c, pos + 1);
                          if (G > 1) hipLaunchKernelGGL(k_attn_decode<true>, dim3(c->heads_local * G), dim3(kAttnBlock), attn_lds_bytes(d.max_seq_len, c->hs, true), st, args_attn(c, l, G));
                          else       hipLaunchKernelGGL(k_attn_decode<false>, dim3(c->heads_local), dim3(kAttnBlock), attn_lds_bytes(d.max_seq_len, c->hs, false), st, args_attn(c, l, 1));
                          return FLM_OK; }
        case KC_ATTN_O: return launch_gemv<PRO_QUANT, EPI_RESIDUAL>(c, st, qt, args_o(c, l), wgs);
        case KC_FFN13:  return launch_gemv<PRO_RMSNORM_QUANT, EPI_SWIGLU>(c, st, qt, args_ffn13(c, l), wgs);
        case KC_FFN2:   return launch_gemv<PRO_QUANT, EPI_RESIDUAL>(c, st, qt, args_ffn2(c, l), wgs);
        case KC_CLS:    return launch_gemv<PRO_RMSNORM_QUANT, EPI_STORE>(c, st, qt, args_cls(c), wgs);
        case KC_ARGMAX: hipLaunchKernelGGL(k_argmax_advance, dim3(1), dim3(1024), 0, st, (const float*)c->logits, d.vocab_size, c->state, (int*)nullptr, 0, 0); return FLM_OK;   // (no id is recorded: the step counter runs on)
        // the fused launches the token path uses on a single GPU (FLM_ERR_UNSUPPORTED: this shape / option setting runs the phases separately)
        case KC_ATTN_WO: if (!c->fuse_attn_o || c->world > 1) return FLM_ERR_UNSUPPORTED;      // (across ranks the launch waits for its peers' heads: not timed in isolation)
                         return qt == FLM_QT_INT8 ? launch_attn_o<QT_INT8>(c, st, l, attn_parts(c, pos + 1)) : launch_attn_o<QT_INT16>(c, st, l, attn_parts(c, pos + 1));
        case KC_FFN:     if (!c->fuse_ffn || c->world > 1) return FLM_ERR_UNSUPPORTED;
                         return qt == FLM_QT_INT8 ? launch_ffn<QT_INT8>(c, st, l) : launch_ffn<QT_INT16>(c, st, l);
        case KC_QKV_ATTN_WO: { const int G = attn_parts(c, pos + 1);
                         if (c->world > 1) return FLM_ERR_UNSUPPORTED;
                         if (!c->fuse_attn_o || !(c->fuse_qkv >= 2 || (c->fuse_qkv && G > 1))) return FLM_ERR_UNSUPPORTED;
                         return qt == FLM_QT_INT8 ? launch_qkv_attn_o<QT_INT8>(c, st, l, G) : launch_qkv_attn_o<QT_INT16>(c, st, l, G); }
        case KC_LAYER: case KC_BACK: {
                         if (c->world > 1 || !c->fuse_back || !c->fuse_attn_o || !c->fuse_ffn || attn_parts(c, pos + 1) != 1 || (kc == KC_LAYER) != (c->fuse_layer != 0)) return FLM_ERR_UNSUPPORTED;
                         return qt == FLM_QT_INT8 ? launch_attn_ffn<QT_INT8>(c, st, l, kc == KC_LAYER) : launch_attn_ffn<QT_INT16>(c, st, l, kc == KC_LAYER); }
        default: return FLM_OK;
        }
    };
    const int classes[] = {KC_EMBED, KC_QKV, KC_ATTN, KC_ATTN_O, KC_FFN13, KC_FFN2, KC_CLS, KC_ARGMAX, KC_ATTN_WO, KC_FFN, KC_QKV_ATTN_WO, KC_LAYER, KC_BACK};
    for (int kc : classes) {
        const bool fused = kc == KC_ATTN_WO || kc == KC_FFN || kc == KC_QKV_ATTN_WO || kc == KC_LAYER || kc == KC_BACK;
        const bool per_layer = (kc >= KC_QKV && kc <= KC_FFN2) || fused;
        const int n = per_layer ? L : 8;
        for (int it = 0; it < iters + 1 && !r; ++it) {          // first round: warm-up
            if (kc == KC_ATTN || fused) r = launch(KC_EMBED, 0);   // (clears the flag lines the workgroups of a fused launch / the parts of a split head wait on)
            HIPC(c, hipEventRecord(e0, st));
            for (int i = 0; i < n && !r; ++i) r = launch(kc, per_layer ? i : 0);
            if (fused && r == FLM_ERR_UNSUPPORTED) { r = FLM_OK; cnt[kc] = 0; break; }
            HIPC(c, hipEventRecord(e1, st));
            HIPC(c, hipEventSynchronize(e1));
            float ms = 0.f; HIPC(c, hipEventElapsedTime(&ms, e0, e1));
            if (it > 0) { tot[kc] += ms * 1000.0 / n; cnt[kc] += 1; }
        }
        avg_us[kc] = cnt[kc] ? (float)(tot[kc] / cnt[kc]) : 0.f;
        count[kc] = cnt[kc] ? (per_layer ? L : 1) : 0;
    }
    avg_us[KC_ALLREDUCE] = 0.f; count[KC_ALLREDUCE] = 0;
    if (r) return r;
    r = xwg_check(c); if (r == FLM_RETRY) return fail(c, FLM_ERR_HIP, "cross-workgroup wait timed out while timing");
    if (r) return r;
    return flm_reset_kv(c);
}

int flm_kernel_bytes(flm_ctx* c, int kclass, int pos, double* bytes) {
    if (!c || !bytes) return FLM_ERR_INVALID;
    const auto& d = c->d; const double e = c->esz, sb = 4.0 / kGroup;
    auto mat = [&](double rows, double cols) { return rows * cols * (e + sb); };
    switch (kclass) {
    case KC_EMBED:  *bytes = d.dim * 4.0; break;
    case KC_QKV:    *bytes = mat(3.0 * c->dim_local, d.dim) + d.dim * 4.0; break;               // + rmsnorm weight
    case KC_ATTN:   *bytes = 2.0 * c->heads_local * c->hs * 4.0 * (pos + 1); break;             // fp32 K and V rows
    case KC_ATTN_O: *bytes = mat(c->drow_count, d.dim); break;
    case KC_FFN13:  *bytes = 2.0 * mat(c->hidden_local, d.dim) + d.dim * 4.0; break;
    case KC_FFN2:   *bytes = mat(c->drow_count, d.hidden_dim); break;
    case KC_ATTN_WO: *bytes = 2.0 * c->heads_local * c->hs * 4.0 * (pos + 1) + mat(c->drow_count, d.dim); break;
    case KC_FFN:    *bytes = 2.0 * mat(c->hidden_local, d.dim) + d.dim * 4.0 + mat(c->drow_count, d.hidden_dim); break;
    case KC_QKV_ATTN_WO: *bytes = mat(3.0 * c->dim_local, d.dim) + d.dim * 4.0 + 2.0 * c->heads_local * c->hs * 4.0 * (pos + 1) + mat(c->drow_count, d.dim); break;
    case KC_CLS:    *bytes = mat(c->cls.rows, d.dim) + d.dim * 4.0; break;
    case KC_BACK:   *bytes = 2.0 * c->heads_local * c->hs * 4.0 * (pos + 1) + mat(c->drow_count, d.dim) + 2.0 * mat(c->hidden_local, d.dim) + d.dim * 4.0 + mat(c->drow_count, d.hidden_dim); break;
    case KC_LAYER:  *bytes = mat(3.0 * c->dim_local, d.dim) + d.dim * 4.0 + 2.0 * c->heads_local * c->hs * 4.0 * (pos + 1) + mat(c->drow_count, d.dim) + 2.0 * mat(c->hidden_local, d.dim) + d.dim * 4.0 + mat(c->drow_count, d.hidden_dim); break;
    default:        *bytes = 0; break;
    }
    return FLM_OK;
}

// ---------------------------------------------------------------------------------------------
// op-level exports
// ---------------------------------------------------------------------------------------------

int flm_op_quantize(int qt, void* qx, float* qs, const float* x, size_t n, int gs) {
    if (!qx || !qs || !x || gs != kGroup || n % kGroup) return FLM_ERR_INVALID;
    if (qt != FLM_QT_INT8 && qt != FLM_QT_INT16) return FLM_ERR_UNSUPPORTED;
    const int e = esz_of(qt);
    DevBuf dx, dq, ds;
    if (dx.alloc(n * 4) || dq.alloc(n * e) || ds.alloc(n / kGroup * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(dx.p, x, n * 4, hipMemcpyHostToDevice));
    if (n <= 16384) {
        // through the fused path's prologue (PRO_QUANT), tapped
        GemvArgs a{}; a.n = (int)n; a.items = 0; a.x = dx.as<float>(); a.dbg_xq = dq.p; a.dbg_xs = ds.as<float>();
        int r = launch_gemv<PRO_QUANT, EPI_STORE>(nullptr, 0, qt, a, 1); if (r) return r;
    } else {
        int r = quantize_flat(nullptr, 0, qt, dq.p, ds.as<float>(), dx.as<float>(), n); if (r) return r;
    }
    OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(qx, dq.p, n * e, hipMemcpyDeviceToHost));
    OPC(hipMemcpy(qs, ds.p, n / kGroup * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

// square_sum (x86_simd.cpp:942-960) of x[n], n a multiple of 16: out6 = { speculative wave evaluation (sq_chain_spec), sequential total, the 4 strided lanes }
int flm_op_square_sum(const float* x, size_t n, float* out6) {
    if (!x || !out6 || n % 16 || n == 0 || n > 16384) return FLM_ERR_INVALID;
    DevBuf dx, dout;
    if (dx.alloc(n * 4) || dout.alloc(16 * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(dx.p, x, n * 4, hipMemcpyHostToDevice));
    const size_t lds = ((size_t)4 * chain_strip_floats((int)n) + 4 * (n / 4 + 8)) * 4;
    hipLaunchKernelGGL(k_op_square_sum, dim3(1), dim3(256), lds, 0, dout.as<float>(), dx.as<float>(), (int)n);
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(out6, dout.p, 6 * 4, hipMemcpyDeviceToHost));
    if (getenv("FLM_SQ_ITERS")) { float it[6]; hipMemcpy(it, (char*)dout.p + 24, 24, hipMemcpyDeviceToHost); fprintf(stderr, "sq_chain_spec rounds per chain (-1: plain chain): %g %g %g %g; shader-clock ticks: speculative %g, plain %g\n", it[0], it[1], it[2], it[3], it[4], it[5]); }
    return FLM_OK;
}

int flm_op_rmsnorm(float* o, const float* x, const float* w, size_t n) {
    if (!o || !x || !w || n % kGroup || n > 16384 || n == 0) return FLM_ERR_INVALID;
    DevBuf dx, dw, dn, dq, ds;
    if (dx.alloc(n * 4) || dw.alloc(n * 4) || dn.alloc(n * 4) || dq.alloc(n) || ds.alloc(n / kGroup * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(dx.p, x, n * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(dw.p, w, n * 4, hipMemcpyHostToDevice));
    GemvArgs a{}; a.n = (int)n; a.items = 0; a.x = dx.as<float>(); a.norm_w = dw.as<float>();
    a.dbg_xn = dn.as<float>(); a.dbg_xq = dq.p; a.dbg_xs = ds.as<float>();
    int r = launch_gemv<PRO_RMSNORM_QUANT, EPI_STORE>(nullptr, 0, FLM_QT_INT8, a, 1); if (r) return r;
    OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(o, dn.p, n * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

int flm_op_matmul_q(int qt, float* out, const void* W, const float* sW, const void* X, const float* sX, int m, int n, int w, int gs) {
    if (!out || !W || !sW || !X || !sX || m < 1 || n < 1 || w < 1 || gs != kGroup || n % kGroup) return FLM_ERR_INVALID;
    if (qt != FLM_QT_INT8 && qt != FLM_QT_INT16) return FLM_ERR_UNSUPPORTED;
    const size_t e = esz_of(qt), sn = n / kGroup;
    DevBuf dW, dsW, dX, dsX, dsXT, dsWT, dO;
    if (dW.alloc((size_t)m * n * e) || dsW.alloc((size_t)m * sn * 4) || dX.alloc((size_t)w * n * e) || dsX.alloc((size_t)w * sn * 4) || dsXT.alloc((size_t)w * sn * 4 + 64) || dsWT.alloc((size_t)m * sn * 4) || dO.alloc((size_t)w * m * 4)) return FLM_ERR_OOM;
    {   // the activation scales once more, group-major (k_rows_prologue writes both layouts on the prompt path)
        std::vector<float> t((size_t)w * sn);
        for (int b = 0; b < w; ++b) for (size_t g = 0; g < sn; ++g) t[g * w + b] = sX[(size_t)b * sn + g];
        OPC(hipMemcpy(dsXT.p, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> tw((size_t)m * sn);
        for (int r = 0; r < m; ++r) for (size_t g = 0; g < sn; ++g) tw[g * m + r] = sW[(size_t)r * sn + g];
        OPC(hipMemcpy(dsWT.p, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
    }
    OPC(hipMemcpy(dW.p, W, (size_t)m * n * e, hipMemcpyHostToDevice)); OPC(hipMemcpy(dsW.p, sW, (size_t)m * sn * 4, hipMemcpyHostToDevice));
    OPC(hipMemcpy(dX.p, X, (size_t)w * n * e, hipMemcpyHostToDevice)); OPC(hipMemcpy(dsX.p, sX, (size_t)w * sn * 4, hipMemcpyHostToDevice));
    int dev = 0, cus = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const char* gv = getenv("FLM_OP_GEMM");                     // tests: 0 .. 3 = launch_gemm's use_mfma, "gemv" = a GEMV per batch row
    if (w >= 16 && !(gv && !strcmp(gv, "gemv"))) {
        // the batched path the prompt takes (quant::matmul with w > 1, quant_operators.cpp:252-284): one tile kernel
        GemmArgs g{dW.p, dsW.as<float>(), dX.p, dsX.as<float>(), dO.as<float>(), m, n, m, w, dsXT.as<float>(), dsWT.as<float>()};
        const int um = gv ? atoi(gv) : 1;
        int r = qt == FLM_QT_INT8 ? launch_gemm<QT_INT8, EPI_STORE>(nullptr, 0, g, um) : launch_gemm<QT_INT16, EPI_STORE>(nullptr, 0, g, um);
        if (r) return r;
    } else {
        for (int b = 0; b < w; ++b) {
            GemvArgs a{}; a.W = dW.p; a.sW = dsW.as<float>(); a.n = n; a.items = m;
            a.xq = (const char*)dX.p + (size_t)b * n * e; a.xs = dsX.as<float>() + (size_t)b * sn; a.out = dO.as<float>() + (size_t)b * m;
            int r = launch_gemv<PRO_NONE, EPI_STORE>(nullptr, 0, qt, a, gemv_grid(cus, 1, m, 1)); if (r) return r;
        }
    }
    OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(out, dO.p, (size_t)w * m * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

/* sample_argmax (sampler.cpp:36-47) as k_argmax_advance evaluates it: first maximum wins */
int flm_op_argmax(const float* logits, int n, int32_t* idx) {
    if (!logits || !idx || n < 1) return FLM_ERR_INVALID;
    DevBuf dl, dst, dout;
    if (dl.alloc((size_t)n * 4) || dst.alloc(sizeof(DecodeState)) || dout.alloc(16)) return FLM_ERR_OOM;
    OPC(hipMemcpy(dl.p, logits, (size_t)n * 4, hipMemcpyHostToDevice));
    OPC(hipMemset(dst.p, 0, sizeof(DecodeState)));
    hipLaunchKernelGGL(k_argmax_advance, dim3(1), dim3(1024), 0, 0, (const float*)dl.as<float>(), n, dst.as<DecodeState>(), dout.as<int>(), 0, 4);
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(idx, dout.p, 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

int flm_op_swiglu(float* xo, const float* xr, size_t n) {
    if (!xo || !xr || n == 0) return FLM_ERR_INVALID;
    DevBuf a, b; if (a.alloc(n * 4) || b.alloc(n * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(a.p, xo, n * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(b.p, xr, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_swiglu, dim3(256), dim3(256), 0, 0, a.as<float>(), (const float*)b.as<float>(), n);
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(xo, a.p, n * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

int flm_op_rope(float* o, const float* x, int n_dims, int pos) {
    if (!o || !x || n_dims < 2 || n_dims % 2 || pos < 0) return FLM_ERR_INVALID;
    std::vector<float> cs, sn; build_rope_table(n_dims, pos + 1, cs, sn);
    DevBuf dx, dout, dc, dsn; const size_t h = n_dims / 2;
    if (dx.alloc(n_dims * 4) || dout.alloc(n_dims * 4) || dc.alloc(h * 4) || dsn.alloc(h * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(dx.p, x, n_dims * 4, hipMemcpyHostToDevice));
    OPC(hipMemcpy(dc.p, cs.data() + (size_t)pos * h, h * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(dsn.p, sn.data() + (size_t)pos * h, h * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_rope, dim3((unsigned)((h + 63) / 64)), dim3(64), 0, 0, dout.as<float>(), (const float*)dx.as<float>(), n_dims, (const float*)dc.as<float>(), (const float*)dsn.as<float>());
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(o, dout.p, n_dims * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

int flm_op_softmax(float* x, int n) {
    if (!x || n < 1) return FLM_ERR_INVALID;
    DevBuf d; if (d.alloc((size_t)n * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(d.p, x, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_softmax, dim3(1), dim3(kBlock), 0, 0, d.as<float>(), n);
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(x, d.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

int flm_op_attention(float* out, float* kc, float* vc, const float* q, const float* k, const float* v,
                     int n_heads, int hs, int max_seq, int pos) {
    if (!out || !kc || !vc || !q || !k || !v || n_heads < 1 || hs < 32 || hs > 256 || hs % 8 || pos < 0 || pos >= max_seq) return FLM_ERR_INVALID;
    const size_t nd = (size_t)n_heads * hs, nc = (size_t)n_heads * max_seq * hs, h2 = hs / 2;
    std::vector<float> cs, sn; build_rope_table(hs, pos + 1, cs, sn);
    DevBuf dq, dk, dv, dkc, dvc, dout, dc, dsn, dpos;
    if (dq.alloc(nd * 4) || dk.alloc(nd * 4) || dv.alloc(nd * 4) || dkc.alloc(nc * 4) || dvc.alloc(nc * 4) || dout.alloc(nd * 4) ||
        dc.alloc(h2 * 4) || dsn.alloc(h2 * 4) || dpos.alloc(4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(dq.p, q, nd * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(dk.p, k, nd * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(dv.p, v, nd * 4, hipMemcpyHostToDevice));
    OPC(hipMemcpy(dkc.p, kc, nc * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(dvc.p, vc, nc * 4, hipMemcpyHostToDevice));
    OPC(hipMemcpy(dc.p, cs.data() + (size_t)pos * h2, h2 * 4, hipMemcpyHostToDevice)); OPC(hipMemcpy(dsn.p, sn.data() + (size_t)pos * h2, h2 * 4, hipMemcpyHostToDevice));
    OPC(hipMemcpy(dpos.p, &pos, 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_kv_append, dim3((unsigned)((nd / 2 + 255) / 256)), dim3(256), 0, 0, dq.as<float>(), (const float*)dk.as<float>(), (const float*)dv.as<float>(),
                       dkc.as<float>(), dvc.as<float>(), (const float*)dc.as<float>(), (const float*)dsn.as<float>(), n_heads, hs, max_seq, pos);
    OPC(hipGetLastError());
    AttnArgs a{}; a.q = dq.as<float>(); a.kcache = dkc.as<float>(); a.vcache = dvc.as<float>(); a.pos_ptr = dpos.as<int>(); a.hs = hs; a.max_seq = max_seq;
    a.out = dout.as<float>();
    // tests: FLM_OP_ATTN_PARTS = G spreads every head over G workgroups (the long-context path of the decode loop)
    int G = (getenv("FLM_OP_ATTN_PARTS") && atoi(getenv("FLM_OP_ATTN_PARTS")) > 1) ? hs / kSplitDims : 1;
    if (G < 2 || hs % kSplitDims || hs > 128 || n_heads * G > 256 || max_seq > kSplitMaxSeq) G = 1;
    DevBuf dsc, dfl, derr;
    if (dsc.alloc((size_t)n_heads * max_seq * 4) || dfl.alloc(256 * 64) || derr.alloc(64)) return FLM_ERR_OOM;
    OPC(hipMemset(dfl.p, 0, 256 * 64)); OPC(hipMemset(derr.p, 0, 64));
    a.G = G; a.sc_global = dsc.as<float>(); a.flag_sc = dfl.as<unsigned>(); a.epoch = 1; a.err = derr.as<int>();
    if (G > 1) hipLaunchKernelGGL(k_attn_decode<true>, dim3(n_heads * G), dim3(kAttnBlock), attn_lds_bytes(max_seq, hs, true), 0, a);
    else       hipLaunchKernelGGL(k_attn_decode<false>, dim3(n_heads), dim3(kAttnBlock), attn_lds_bytes(max_seq, hs, false), 0, a);
    OPC(hipGetLastError());
    OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(out, dout.p, nd * 4, hipMemcpyDeviceToHost));
    OPC(hipMemcpy(kc, dkc.p, nc * 4, hipMemcpyDeviceToHost)); OPC(hipMemcpy(vc, dvc.p, nc * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}

/* elementary functions exactly as the kernels evaluate them (tests pin them against the host's IEEE results) */
int flm_op_math(int fn, float* x, const float* y, size_t n) {
    if (!x || n == 0 || fn < 0 || fn > 3 || (fn >= 2 && !y)) return FLM_ERR_INVALID;
    DevBuf d, e; if (d.alloc(n * 4) || e.alloc(n * 4)) return FLM_ERR_OOM;
    OPC(hipMemcpy(d.p, x, n * 4, hipMemcpyHostToDevice));
    if (y) OPC(hipMemcpy(e.p, y, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_math, dim3(1024), dim3(256), 0, 0, fn, d.as<float>(), (const float*)e.as<float>(), n);
    OPC(hipGetLastError()); OPC(hipDeviceSynchronize());
    OPC(hipMemcpy(x, d.p, n * 4, hipMemcpyDeviceToHost));
    return FLM_OK;
}
int flm_op_expf(float* x, size_t n) { return flm_op_math(0, x, nullptr, n); }

} // extern "C"
