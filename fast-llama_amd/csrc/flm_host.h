// flm_host.h -- what the translation units of libflm_gpu.so share on the HOST side: the context, the launch planning of the GEMV and the entry points of one
// translation unit that another one calls.
//   flm_gpu.hip          context, upload, options, p2p bootstrap, the model-level C ABI (forward / decode / debug taps), run_token
//   flm_token.hip        the per-token launch sequence (enqueue_token), the per-phase and fused decode launches, per-kernel timing (flm_kernel_times / _bytes)
//   flm_layerlaunch.hip  the whole-layer launch k_attn_ffn (flm_layer.h)
//   flm_prompt.hip       the batched prompt path (GEMM tiles on the matrix cores, prompt attention)
//   flm_ops.hip          the op-level exports of the parity tests (flm_op_*)
#pragma once
#include "flm_gpu.h"
#include "flm_kernels.h"
#include "flm_tuning.h"

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

namespace fh {
using namespace flm;
extern thread_local std::string g_last_error;

struct QMat { void* q = nullptr; float* s = nullptr; int rows = 0, cols = 0; float* st = nullptr; /* s group-major [cols / 64][rows]: the prompt path's GEMM tiles */ };
struct LayerW {
    QMat qkv, o, w13, w2;     // w13 = [W1 (gate) ; W3 (up)] back to back: the SwiGLU GEMV walks them as one matrix
    float* att_norm = nullptr; float* ffn_norm = nullptr;
    unsigned got = 0;     // bitmask of uploaded kinds
};

enum KClass { KC_EMBED = 0, KC_QKV, KC_ATTN, KC_ATTN_O, KC_FFN13, KC_FFN2, KC_CLS, KC_ARGMAX, KC_ALLREDUCE, KC_ATTN_WO /* k_attn_o: attention + Wo */, KC_FFN /* k_ffn: FFN13 + FFN2 */, KC_QKV_ATTN_WO /* k_qkv_attn_o */, KC_LAYER /* k_attn_ffn with the QKV GEMV in front: the whole layer */, KC_BACK /* k_attn_ffn: attention + Wo + FFN13 + FFN2 */, KC_LAYERS /* k_layers: all layers of the token in one launch */, KC_TOKEN /* k_layers<.., TAIL>: the whole greedy token in one launch */ };

struct TimedLaunch { int kclass; hipEvent_t e0, e1; };
// owners that release on every exit path (the error macros return from the middle of a function)
struct DevMem { void* p = nullptr; ~DevMem() { if (p) hipFree(p); } };
struct EvPair { hipEvent_t e0 = nullptr, e1 = nullptr; ~EvPair() { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); } };


} // namespace fh
using namespace fh;

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

    float *kcache = nullptr, *vcache = nullptr;       // [L][heads_local][kv_rows][hs]
    int kv_rows = 0;                                   // max_seq_len + 8: rows per head in the caches (a head's rows start 4 KiB (hs 128) off the power-of-two stride: DESIGN.md section 7c)
    float *x1 = nullptr, *qbuf = nullptr, *att_out = nullptr, *hd = nullptr;
    float *logits = nullptr;
    float *rope_cos = nullptr, *rope_sin = nullptr;
    DecodeState* state = nullptr; int* prompt_dev = nullptr; int* out_tokens_dev = nullptr;
    int prompt_cap = 0, out_cap = 0;
    // the context's own page-locked bounce buffer: logits, ids and prompt tokens cross the bus through it (allocated at create).  A caller's pageable buffer handed to
    // hipMemcpyAsync is pinned and mapped by the runtime on every call -- ~100 us of host time per call and, the first time an address range is seen, page tables in device memory
    // (2 MiB steps: tools/alloc_diag.py), i.e. an allocation inside flm_forward.
    char* bounce = nullptr; size_t bounce_bytes = 0; int err_word = 0; bool err_word_fresh = false;
    bool warmed = false;                               // flm_gpu.hip warm_up: the runtime's lazily built launch resources were set up at load time

    // options
    bool tuning = false;                               // option "tuning": the experiment dials (flm_tuning.h) may be set
    int wg_per_cu = 1; int use_graph = 1; int ablate = 0;
    int graph_chunks = 1;                              // option "graph_chunks": a greedy decode loop replays graphs of up to 16 tokens (0: one graph launch per token)
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
    int fuse_token = 1;                                // option "fuse_token": ALL layers of a token in one launch (k_layers: the edge between two layers is a flag round in front of which [Wq; Wk; Wv] streams)
    int tok_nstq = 4, tok_preq = 99 /* 99: by launch (plan_layer) */;                   // options "tok_nstq" / "tok_preq": its stash slots / early waves of [Wq; Wk; Wv]
    void* la_dev[2] = {nullptr, nullptr}; bool la_valid[2] = {false, false}, la_ok[2] = {false, false}; flm::BackArgs la_p[2]; int la_grid[2] = {0, 0}, la_r2[2] = {0, 0};   // k_layers' argument blocks (flm_layers.hip)
    int fuse_tail = 1;                                 // option "fuse_tail": a greedy decode token is ONE launch (k_layers<.., TAIL>: embedding row, all layers, classifier, argmax + state advance); fp32 embedding tables
    void* tail_dev[2] = {nullptr, nullptr}; bool tail_ok[2] = {false, false};   // its argument block per head split (flm_layers.hip)
    unsigned* tail_mem = nullptr;                      // [0] the epoch base of the one-launch token's flag values, [16 ..) one flag line per classifier workgroup, then their argmax slots
    int fuse_layer = 1;                                // option "fuse_layer": ... with the QKV GEMV in front: the whole layer in one launch
    int back_nst13 = -1, back_nst13_head = -1, back_nst2 = 0, back_pre13 = 99 /* 99: by launch (plan_layer) */, back_pre2 = 16;   // options "back_*": k_attn_ffn's stash slots (-1: as many as the LDS holds) and early register set (flm_layer.h)
    int attn_kpre = 1;                                 // tuning dial "attn_kpre": split heads' first two K tiles by LDS-DMA under the QKV phase (BackArgs::kpre_off)
    int back_nwo = 0;                                  // tuning dial "back_nwo": waves that hold the arrival-order Wo's steps (0: ceil(steps / 2); 16: every wave, the stash issued by all)
    int gr_edges = 1;                                  // tuning dial "gr_edges" (round 6): the one-launch token's x / x1 hand-offs as data-tagged granules (flm_gemv.h: granule_t; BackArgs::gr); 0: flag rounds
    granule_t* xg = nullptr; size_t x_gran_off = 0;    // ... the granule vectors [x: dim][x1: dim][att: dim][hd: hidden] (tensor parallel: a region of the exchange buffer, at x_gran_off in every rank's)
    int back_ao = 3, back_ao2 = 2;                     // options "back_ao" (bit 0: Wo, bit 1: FFN2 consume their activation in arrival order, GemvCtx::run_ao) / "back_ao2" (what of W2 is requested in front of the first look)
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
    int tp_fuse_layers = 1;                            // option "tp_fuse_layers" (round 6): ALL layers of a sharded token in one launch that spans the ranks (k_layers<.., TP>: the four hand-offs of a layer are
                                                       // flag rounds between the ranks' workgroups, the next phase's weights requested in front of each, exactly as on one GPU); needs what the
                                                       // rank-spanning launches need (grp_span) and ranks of identical geometry
    int tp_fence = -1;                                 // option "tp_fence" (k_layers<.., TP>): bit 0 a system-scope release fence in front of a cross-rank line, bit 1 an acquire fence behind a cross-rank poll;
                                                       // -1 (default): none where every rank of the group lives on THIS device (one memory system), both between distinct devices
    size_t x_tlines_off = 0;                           // ... its flag region in the exchange buffer: [heads: 256 lines][x1, hd, x, cls: world x 256 lines each]
    char* peer[8] = {nullptr}; bool peer_opened[8] = {false}; int p2p = 0;
    // what the tensor-parallel GROUP runs, agreed at flm_p2p_import from every rank's blob (the ranks' hand-off protocols must match or they wait on flags nobody raises):
    // exchanges folded into the consuming launches / launches that span the ranks / tp_fuse_attn / tp_fuse_ffn / attn_split, each the weakest any rank can do.
    // Options set after the import take effect at the next flm_p2p_export + flm_p2p_import round of the whole group.
    bool grp_fold = false, grp_span = false, grp_can_split = false, grp_tpl = false, grp_gr = false /* every rank has "gr_edges": the rank-spanning launch's vectors are granules */; int grp_tpfa = 0, grp_tpff = 0, grp_split = 0;
    int force_tp = 0;                                  // option "force_tp": a context created with an RCCL id but world == 1 takes the sharded token path (RCCL exchanges over a 1-rank communicator: tests)
    int tp_trust_fused = 0;                            // option "tp_trust_fused": ranks on DISTINCT devices run the folded / rank-spanning launches too (validated only between CU partitions of one GPU)
    unsigned* xepoch = nullptr;                        // [4] exchanges done per kind (att, x1, hd, logits), device memory
    float* att_sc = nullptr;                           // [heads_local][max_seq] scores exchanged between the parts of a split head
    int attn_split = 1;                                // option "attn_split": 1 = spread a head over 4 workgroups from kSplitFrom (128) positions on, 0 = never, >= 2 = always that many
    unsigned* eng_base = nullptr;                      // the token's epoch base (device memory, advanced by k_embed): the tensor-parallel exchanges' flag values count from it
    int resident = 1;                                  // the census at create saw every workgroup of a cu_count-wide launch co-resident
    int fell_back = 0;                                 // how many times a cross-workgroup wait timed out and the context went to one kernel per phase ("fallback")
    bool fb_active = false; int fb_tokens = 0; int fb_saved[6] = {0, 0, 0, 0, 0, 0};   // ... it is there now / tokens since / the launch structure it had (restored when the census passes again: maybe_recover)
    int trace_class = -1; unsigned long long* trace = nullptr;   // FLM_ABLATE builds: GEMV timeline of one kernel class
    std::map<int, hipGraphExec_t> graphs;             // key = with_cls*4 + advance
    std::vector<TimedLaunch>* timing = nullptr;
    std::string err;
};

namespace fh {

#define HIPC(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    if (ctx) (ctx)->err = b_; g_last_error = b_; return FLM_ERR_HIP; } } while (0)
#define NCCLC(ctx, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { \
    char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
    if (ctx) (ctx)->err = b_; g_last_error = b_; return FLM_ERR_COMM; } } while (0)

int fail(flm_ctx* c, int code, const char* msg);
int esz_of(int qt);
void split_even(int total, int parts, int idx, int* begin, int* count);

// Pass geometry of k_gemv for one launch (flm_token.hip)
struct GemvPlan { int Rm, cb_shift, grid, nbuf; size_t lds; };
GemvPlan gemv_plan(int n, int esz, int rows, bool two, bool pairs, bool norm, int wgs);
// fill the pass geometry of one GEMV into its argument block; returns the LDS bytes and grid it needs
template <int QT, int PRO, int EPI>
int plan_gemv(flm_ctx* c, GemvArgs& a, int wgs, GemvPlan& P) {
    constexpr bool TWO = EPI == EPI_SWIGLU, PAIRS = EPI == EPI_ROPE_KV;
    const int rows = a.items * (PAIRS ? 2 : 1);
    if ((double)rows * a.n * QTraits<QT>::kEsz * (TWO ? 2 : 1) >= 2147483648.0) return fail(c, FLM_ERR_UNSUPPORTED, "gemv: matrix of 2 GiB or more");
    P = gemv_plan(a.n, QTraits<QT>::kEsz, rows, TWO, PAIRS, PRO == PRO_RMSNORM_QUANT, wgs);     // (the chain staging area only where an rmsnorm prologue runs)
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
inline int gemv_grid(int cu_count, int wg_per_cu, int /*items*/, int /*rows_per_item*/) { return cu_count * wg_per_cu; }
int quantize_flat(flm_ctx* c, hipStream_t st, int qt, void* q, float* s, const float* x, size_t n);

// k_attn_o & co. report a cross-workgroup wait that never completed through *xwg_err (flm_gpu.hip)
constexpr int FLM_RETRY = 1;
int xwg_check(flm_ctx* c);

struct Tick {
    flm_ctx* c; hipStream_t st; int kclass; hipEvent_t e0 = nullptr, e1 = nullptr;
    Tick(flm_ctx* c_, hipStream_t st_, int k) : c(c_), st(st_), kclass(k) {
        if (c->timing) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, st); }
    }
    ~Tick() { if (c->timing) { hipEventRecord(e1, st); c->timing->push_back({kclass, e0, e1}); } }
};

// peer-to-peer tensor parallelism: where `p` (a pointer into this rank's exchange buffer) lies in every peer's buffer
template <class A> void set_peers(flm_ctx* c, A& a, float* p) {
    a.n_peer = 0;
    if (!c->p2p) return;
    const size_t off = (char*)p - c->xbuf;
    for (int r = 0; r < c->world; ++r) if (r != c->rank) a.out_peer[a.n_peer++] = (float*)(c->peer[r] + off);
}
// argument blocks of the five GEMVs and the attention of layer l (flm_token.hip)
GemvArgs args_qkv(flm_ctx* c, int l);
constexpr int kSplitFrom = 128;
constexpr int kTpLinesPerRank = 256;                 // k_layers<.., TP>: flag lines per rank and kind in the exchange buffer's region (one per workgroup of a launch)
int attn_parts(const flm_ctx* c, int T);
AttnArgs args_attn(flm_ctx* c, int l, int G = 1);
GemvArgs args_o(flm_ctx* c, int l);
GemvArgs args_ffn13(flm_ctx* c, int l);
GemvArgs args_ffn2(flm_ctx* c, int l);
GemvArgs args_cls(flm_ctx* c);
void set_fold(flm_ctx* c, GemvArgs& a, int l, int kind);
// the whole decoder layer (with_qkv) / attention .. FFN2 of layer l in one launch (flm_layerlaunch.hip); FLM_ERR_UNSUPPORTED when the shape does not allow it
int launch_layer(flm_ctx* c, hipStream_t st, int qt, int l, bool with_qkv, int G = 1);
// all layers of a token in one launch (flm_layers.hip): the argument blocks are built outside any capture
int layers_prepare(flm_ctx* c, int G);
int launch_layers(flm_ctx* c, hipStream_t st, int l0, int l1, int G, bool tail = false);   // G: workgroups per head (attn_parts); tail: the whole greedy token (FLM_ERR_UNSUPPORTED where that launch does not exist)
// one activation exchange between the tensor-parallel ranks (flm_token.hip)
enum XKind { XK_ATT = 0, XK_X1 = 1, XK_HD = 2, XK_LOGITS = 3 };
int exchange(flm_ctx* c, hipStream_t st, int kind, float* full, float* mine, int count);
int enqueue_token(flm_ctx* c, hipStream_t st, bool with_cls, int advance, int G);
int run_token(flm_ctx* c, bool with_cls, int advance, int T);
int set_state(flm_ctx* c, int pos, int tok, int step);
int check_ready(flm_ctx* c, int n, int pos);
constexpr int kPrefillMin = 4;
int prefill_batched_qt(flm_ctx* c, int B, int pos);      // (flm_prompt.hip; by the model's quant type)
int launch_gemm_store(flm_ctx* c, hipStream_t st, int qt, const GemmArgs& g, int use_mfma);   // (flm_prompt.hip: one GEMM tile launch, plain store epilogue: flm_op_matmul_q)
void build_rope_table(int hs, int max_seq, std::vector<float>& cs, std::vector<float>& sn);  // (flm_gpu.hip)

} // namespace fh
