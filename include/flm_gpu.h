/* flm_gpu.h -- C ABI of the MI355X (gfx950) implementation of fast-llama's per-token hot path.
 *
 * The reference (CoderLSF/fast-llama) has no plugin/FFI seam; its path sits behind the C++ class
 * ParallelTransformer (src/transformer/transformer.h:76-99) and the raw-pointer operator set
 * cpuft::quant::* (src/blas/quant_operators.h:37-82) / cpuft::simd::* (src/platforms/arch/simd.h:13-63).
 * This header is the boundary a maintainer would bind instead: one device-resident forward per
 * token (model level) plus 1:1 mirrors of the operator seam (op level, used by the parity tests).
 * Plain pointers and sizes only; no C++/torch types; no exceptions cross it.  Every entry point
 * cites the reference interface it replaces (paths relative to the reference repo root).
 *
 * Threading: one caller thread per ctx at a time (as ParallelTransformer::forward, single caller,
 * transformer.h:101-110).  Independent ctxs (replicas / tensor-parallel ranks) are independent.
 * Ownership: the caller owns every host pointer (copied during the call); the ctx owns all device
 * memory, ALL of it allocated at flm_ctx_create / flm_upload_tensor time (prompt, output-id and batched-prefill buffers are
 * sized by max_seq_len; the launches' argument blocks and every token graph the entry points replay are built when the model's
 * last tensor arrives, at flm_p2p_import, or by flm_prepare) -- nothing is allocated inside flm_forward* / flm_decode_* (the
 * reference's zero-allocation contract, transformer.cpp:110-130; tests/test_gpu_configs.py brackets the first calls with
 * hipMemGetInfo and a hipMalloc interposer).  Exceptions, both off the steady path: after flm_set_option the graphs are
 * re-instantiated by the next call (or by flm_prepare), and so they are when a context returns from a fallback.
 * Robustness: the fused launches hand data between workgroups of one kernel; if a hand-off ever times out (a workgroup not
 * resident because another process holds CUs) the call re-runs its work on one kernel per phase and still returns FLM_OK with
 * correct results; the context stays on that path for 64 tokens, then takes the census again and, if every workgroup is
 * resident, returns to the launch structure it had ("fallback" counts the episodes, "fallback_active" = on that path now).
 */
#ifndef FLM_GPU_H
#define FLM_GPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* status codes (reference: bool + printf, src/utils/log.h; here: int + flm_last_error) */
#define FLM_OK               0
#define FLM_ERR_INVALID     -1   /* bad argument / shape */
#define FLM_ERR_UNSUPPORTED -2   /* e.g. n_kv_heads != n_heads (reference GQA path is broken, transformer.cpp:449) */
#define FLM_ERR_HIP         -3   /* HIP runtime error, see flm_last_error */
#define FLM_ERR_OOM         -4
#define FLM_ERR_STATE       -5   /* e.g. forward before all tensors were uploaded */
#define FLM_ERR_COMM        -6   /* RCCL error */

/* QuantType numbering == cpuft::quant::QuantType (src/blas/quant_operators.h:18-25) */
#define FLM_QT_NONE  0
#define FLM_QT_INT16 1
#define FLM_QT_INT8  2

/* tensor kinds == .flm TensorType (src/model_loaders/flm_loader.cpp:50-67) */
#define FLM_T_TOKEN_EMBD  1
#define FLM_T_OUTPUT_NORM 2
#define FLM_T_CLASSIFIER  3
#define FLM_T_INPUT_NORM 17
#define FLM_T_ATTN_Q     18
#define FLM_T_ATTN_K     19
#define FLM_T_ATTN_V     20
#define FLM_T_ATTN_O     21
#define FLM_T_MLP_GATE   22   /* ffn_1 */
#define FLM_T_MLP_UP     23   /* ffn_3 */
#define FLM_T_MLP_DOWN   24   /* ffn_2 */
#define FLM_T_POST_NORM  25

typedef struct flm_ctx flm_ctx;

/* == TransformerConfig (src/model_loaders/model_loader.h:46-68), the fields the path uses.
 * rope_freq_base / rms_norm_eps are NOT here on purpose: the reference ops hard-code 10000 and
 * 1e-5 (tf_operators.cpp:353, x86_simd.cpp:1755) and ignore the file's values. */
typedef struct flm_model_desc {
    int32_t dim;
    int32_t hidden_dim;
    int32_t n_layers;
    int32_t n_heads;
    int32_t n_kv_heads;       /* must equal n_heads (see FLM_ERR_UNSUPPORTED) */
    int32_t vocab_size;
    int32_t max_seq_len;      /* reference clamps to 1024 (transformer.cpp:32); here any length whose scores (4 bytes per position) fit the
                               * LDS beside a head's K/V tiles: ~22 000 positions at head size 128 (FLM_ERR_UNSUPPORTED beyond) */
    int32_t quant_type;       /* FLM_QT_INT8 | FLM_QT_INT16: type of the linear layers and activations */
    int32_t quant_group_size; /* 64 */
} flm_model_desc;

/* ---- lifecycle: replaces ParallelTransformer::load (transformer.cpp:23-42) + parallel_*_init
 *      (transformer.cpp:209-384).  device_id = HIP ordinal.  rank/world/comm_id: tensor-parallel
 *      group.  comm_id = 128 bytes from flm_comm_unique_id() on rank 0, distributed by the caller (RCCL all-gathers, the
 *      fallback exchange), or NULL: then the ranks must be connected peer to peer (flm_p2p_export / flm_p2p_import). */
int  flm_comm_unique_id(void* out128);
int  flm_ctx_create(const flm_model_desc* desc, int device_id, int rank, int world,
                    const void* comm_id, flm_ctx** out);
void flm_ctx_destroy(flm_ctx* ctx);
/* Tensor parallel, peer to peer (world > 1): every rank exports a blob describing its exchange buffer, the caller gathers all
 * ranks' blobs in rank order (any transport) and imports them on every rank; activation slices then travel as direct stores
 * over xGMI plus a flag round instead of RCCL all-gathers, and comm_id may be NULL at create.  The reference's threads share
 * these vectors in memory (transformer.cpp:394,465,482,493,504); this is the same picture across GPUs. */
#define FLM_P2P_BLOB_BYTES 128
int  flm_p2p_export(flm_ctx* ctx, void* blob128);
int  flm_p2p_import(flm_ctx* ctx, const void* blobs /* [world][128] */, int world);
const char* flm_last_error(const flm_ctx* ctx);   /* ctx may be NULL: last create error */

/* Hand one tensor (one layer of it) to the device: what load_tensor (flm_loader.cpp:493-559) +
 * copy_layers (transformer.cpp:289-304) do for the CPU threads.  `values` is row-major
 * [rows][cols] of src_qtype (fp32 when FLM_QT_NONE) with fp32 `scales` [rows][cols/gs] when
 * quantized.  fp32 linear-layer tensors are quantized on the device with the reference's
 * quantizer (A13).  Under tensor parallelism the FULL tensor is passed on every rank; the ctx
 * keeps its row shard. */
int  flm_upload_tensor(flm_ctx* ctx, int kind, int layer, int src_qtype,
                       const void* values, const float* scales, int rows, int cols);
/* Build everything the token entry points use beyond the buffers -- the launches' argument blocks and the hipGraphs of every token form (parallel_thread_init's
 * arena carving, transformer.cpp:110-130,253-384) -- so that the calls below allocate nothing.  Done automatically when the last tensor arrives and at flm_p2p_import;
 * call it after flm_set_option if the next forward must not pay for it.  FLM_ERR_STATE before the model is complete. */
int  flm_prepare(flm_ctx* ctx);

/* ---- the hot path: replaces ParallelTransformer::forward (transformer.h:99, transformer.cpp:105-161).
 * tokens[n] enter at absolute position pos (pos = tokens already in the KV cache);
 * logits_host[vocab] = logits of the LAST token.  n > 5: the first n-1 tokens are evaluated as one batch (MFMA int8 GEMM,
 * causal attention per query) that leaves the same cache rows as feeding them one by one -- bit-identical results. */
int  flm_forward(flm_ctx* ctx, const int32_t* tokens, int n, int pos, float* logits_host);
/* same + sample_argmax (src/transformer/sampler.cpp:36-47, first maximum wins) on the device */
int  flm_forward_argmax(flm_ctx* ctx, const int32_t* tokens, int n, int pos, int32_t* next_token);
/* Device-resident greedy loop == the body of ParallelTransformer::generate (transformer.cpp:92-101)
 * at temperature 0: feeds `first_token` at `pos`, then n_steps-1 further argmax tokens, no host
 * round trip between tokens.  out_tokens[n_steps] receives every sampled token.  Does not stop
 * on token 0 (the caller truncates). */
int  flm_decode_greedy(flm_ctx* ctx, int32_t first_token, int pos, int n_steps, int32_t* out_tokens);
/* Same loop, nothing copied back; *ms = device time of the n_steps tokens measured with HIP events
 * on the ctx's stream (bench.py's timed region; call flm_sync afterwards is not needed). */
int  flm_decode_timed(flm_ctx* ctx, int32_t first_token, int pos, int n_steps, float* ms);
/* the same with an event after every token: ms_each[n_steps] (medians; the events cost a few us per token) */
int  flm_decode_timed_each(flm_ctx* ctx, int32_t first_token, int pos, int n_steps, float* ms_each);
/* the ids generated by the last flm_decode_greedy / flm_decode_timed* call: out[n] (n <= its n_steps) */
int  flm_last_tokens(flm_ctx* ctx, int n, int32_t* out);
int  flm_reset_kv(flm_ctx* ctx);
int  flm_sync(flm_ctx* ctx);

/* Per-kernel timing at position pos with HIP events on the ctx's stream, averaged over `iters` rounds.
 * Classes: 0 embed, 1 qkv, 2 attn, 3 attn_o, 4 ffn13, 5 ffn2, 6 cls, 7 argmax, 8 allreduce (tensor parallel), and the two fused launches
 * the single-GPU token path runs instead of (2, 3) and (4, 5): 9 attn_wo (attention + Wo), 10 ffn (FFN13 + FFN2); count 0 = not in use.
 * 11 qkv_attn_wo: QKV + attention + Wo in one launch, what the token path runs instead of (1, 9) at long contexts ("fuse_qkv").
 * 12 layer: the whole decoder layer in one launch (k_attn_ffn: QKV, attention, Wo, FFN13, FFN2 -- what the token path runs instead of (1, 9, 10) where a head is
 * one workgroup and the head size a multiple of 64; option "fuse_layer"), 13 back: the same without the QKV GEMV ("fuse_layer" 0: instead of (9, 10)).
 * 14 layers: ALL layers of the token in one launch (k_layers: what the token path runs instead of L launches of class 12; option "fuse_token"): ONE launch per token,
 * avg_us = the duration of that launch, flm_kernel_bytes = L layers' bytes.
 * 15 token: a greedy decode token as ONE launch (k_layers<.., TAIL>: the embedding row read by the first layer, the L layers, the classifier, the argmax and the state's advance;
 * option "fuse_tail"; what flm_decode_* runs instead of (0, 14, 6, 7) for fp32 embedding tables where the arrival-order launch runs); bytes = the layers' + the classifier's.
 * avg_us[c] = mean duration of ONE launch of class c (single GPU: the class's launches of one token are
 * enqueued back to back between one pair of events, so the figure is launch duration + dispatch gap and
 * agrees with a rocprofv3 kernel trace), count[c] = launches of that class per token.
 * Side effect: the KV cache is cleared and the decode state is undefined afterwards. */
#define FLM_KCLASSES 16
int  flm_kernel_times(flm_ctx* ctx, int pos, int iters, float* avg_us, int32_t* count);
/* weight + scale bytes one launch of class c streams (the algorithmic bytes of DESIGN.md) */
int  flm_kernel_bytes(flm_ctx* ctx, int kclass, int pos, double* bytes);

/* debugging tap for the parity tests: copy an internal fp32 device buffer to the host.
 * what: 0 residual x1[dim], 1 q[dim], 2 attention output[dim], 3 hd[hidden], 4 K cache of `layer`
 * [heads][max_seq][hs], 5 V cache of `layer`, 6 logits. */
int  flm_debug_read(flm_ctx* ctx, int what, int layer, float* out, size_t n);

/* Structure switches: which launches a token runs.  None of them changes a result bit; the defaults are what was measured fastest.
 *   "use_graph"      0 = launches enqueued eagerly (default 1: a token is one hipGraph replay)
 *   "graph_chunks"   0 = one graph launch per greedy token (default 1: flm_decode_* replay graphs of up to 16 tokens -- the device idles ~10 us between two graph launches, ~1.5 between two nodes)
 *   "fuse_attn_o"    0 = attention and the Wo GEMV as two launches (default 1: one launch, single GPU)
 *   "fuse_ffn"       0 = FFN13 and FFN2 as two launches (default 1)
 *   "fuse_qkv"       QKV in the same launch as attention + Wo: 0 never, 1 (default) where a head is spread over several workgroups, 2 always
 *   "fuse_back"      0 = attention + Wo and FFN13 + FFN2 as two launches (k_attn_o, k_ffn) instead of one (k_attn_ffn; default 1)
 *   "fuse_layer"     0 = the QKV GEMV as its own launch in front of k_attn_ffn (default 1: the whole decoder layer in one launch)
 *   "fuse_token"     0 = one launch per layer instead of one for all layers (k_layers; default 1: the edge between two layers is a flag round)
 *   "fuse_tail"      0 = a greedy decode token as four launches (embedding row, k_layers, classifier, argmax) instead of one (default 1)
 *   "back_ao"        0 = inside k_layers, Wo and FFN2 wait for ALL producers of their activation (round 4); default 3: consumed in arrival order (a wave waits
 *                    for the producers of its own steps' column blocks only)
 *   "attn_split"     0 = one workgroup per head at every context length (default 1: hs / 32 workgroups per head from 128 positions on; n >= 2: always n)
 *   "use_prefill"    0 = prompts token by token (default 1: batched; under tensor parallelism once the peers are mapped with flm_p2p_import)
 *   "use_prefill_mq" 0 = batched attention with one query per workgroup (default 1: eight)
 *   "use_qk_mfma" / "use_pv_mfma"  0 = prefill scores / softmax x V on VALU chains (default 1: v_mfma_f32_16x16x4_f32, the same bits)
 * Tensor parallel (set on every rank alike, before flm_p2p_export where noted):
 *   "use_p2p"        0 = exchanges by RCCL all-gathers although the peers are mapped (1: peer to peer again)
 *   "fold_xchg"      0 = every peer-to-peer exchange's flag round as a launch of its own (k_xchg); default 1: inside the launch that consumes the vector
 *   "tp_fuse_attn"   folded exchanges: 1 = attention + Wo in ONE launch across the ranks, 2 (default) = with the QKV GEMV in front, 0 = separate launches
 *   "tp_fuse_ffn"    the same for FFN13 + FFN2 (default 0)
 *   "tp_fuse_layers" 1 (default) = ALL layers of a sharded token in one launch per rank that spans the ranks (k_layers<.., TP>: the single-GPU persistent launch with the
 *                    reference's row split across the ranks; the four hand-offs of a layer are flag rounds between the ranks' workgroups); where the group can span
 *                    (folded exchanges, every rank's workgroups resident, identical geometry), else the per-layer launches above; before flm_p2p_export
 *   "tp_fence"       that launch's system-scope fences: bit 0 release in front of a cross-rank flag line, bit 1 acquire behind a cross-rank poll; -1 (default) = none between
 *                    ranks of ONE device, both between distinct devices (every cross-rank access is itself a system-scope atomic or a coherent load: the fences are belt and braces,
 *                    and cost ~20 us per hand-off)
 *   "gr_edges"       1 (default) = inside the one-launch token (one GPU) and inside the rank-spanning launch (tensor parallel, where every rank says so; before flm_p2p_export) the
 *                    vectors that cross workgroups / ranks -- the residual stream behind Wo and behind FFN2; across ranks also the heads' output and FFN13's hd -- travel as
 *                    8-byte {value, tag} granules: ONE aligned store per element, the tag = the hand-off's flag value.  The data is its own flag: no drained stores, no flag
 *                    line, no fence, nothing inferred from the ORDER of stores (over xGMI a granule is one write); the consumers re-read their own granules until the tags
 *                    match.  0 = flag rounds (rounds 4-5).  Same results bit for bit
 *   "tp_trust_fused" 1 = between DISTINCT devices too, run the folded exchanges / rank-spanning launches (default 0: the k_xchg launches; before flm_p2p_export)
 *   "cu_parts"       n = confine the context's stream to 1/n of the device's CUs (part rank % n): several ranks on ONE GPU (tests)
 *   "force_tp"       1 = a context created with an RCCL id and world == 1 takes the sharded token path (RCCL exchanges over a 1-rank communicator; tests)
 * Not part of the boundary: the experiment dials whose optimum was measured and fixed (stash slots, early register sets, tile shapes: csrc/flm_tuning.h) are refused
 * until "tuning" 1 has been set; switches that skip work ("ablate", "trace") exist only in -DFLM_ABLATE=1 builds.  Unknown key: FLM_ERR_INVALID. */
int  flm_set_option(flm_ctx* ctx, const char* key, int value);
/* What the context actually runs (bench.py reports it; a caller can see that a fused launch was given up).  Keys: every flm_set_option key
 * (its current value; the dials of csrc/flm_tuning.h too), "tuning", and
 *   "resident"  1 = the census at flm_ctx_create saw one 1024-thread workgroup per CU co-resident (the fused launches wait across workgroups;
 *               0 = they were switched off up front: a masked / partitioned device),
 *   "fallback"  how many times a cross-workgroup wait timed out during a call and the context fell back to one kernel per phase (flm_gpu.hip
 *               xwg_check; the call itself was re-run and returned correct results), "fallback_active" 1 = it is on that path now (after 64 tokens the census
 *               runs again and a context whose workgroups are all resident returns to the launch structure it had),
 *   "token_path" bit 0 attention + Wo fused, bit 1 FFN13 + FFN2 fused, bit 2 QKV joins the attention's launch at long contexts, bit 3 the same
 *               at every context, bit 6 heads split over workgroups at long contexts, bit 7 attention .. FFN2 in one launch (k_attn_ffn), bit 8 with the QKV GEMV in front
 *               (the whole layer in one launch), bit 9 all layers of the token in one launch (k_layers), bit 10 a greedy decode token is ONE launch (embedding row, layers,
 *               classifier, argmax in k_layers<.., TAIL>),
 *   "ao_active" which hand-offs of that launch are consumed in arrival order: bit 0 Wo, bit 1 FFN2 (-1: the launch has not been planned yet).
 * Unknown key: FLM_ERR_INVALID. */
int  flm_query(flm_ctx* ctx, const char* key, int* value);

/* ---- op level: 1:1 mirrors of the reference operator seam, host pointers in / out, running the
 *      same device code as the fused path.  Used by the parity tests. ----------------------- */
/* quant::quantize (quant_operators.cpp:78-97) */
int  flm_op_quantize(int qt, void* qx, float* qs, const float* x, size_t n, int gs);
/* quant::matmul (quant_operators.cpp:571-591), same argument order: out[w][m].  w < 16: one GEMV per batch row (the decode
 * kernel); w >= 16: the tile kernels of the batched prompt path (int8 and int16 on the int8 matrix cores; FLM_OP_GEMM=1|2|3: the tile shape) */
int  flm_op_matmul_q(int qt, float* out, const void* mat1, const float* scales1,
                     const void* mat2, const float* scales2, int m, int n, int w, int gs);
/* simd::rmsnorm(o,x,w,n) (x86_simd.cpp:1754-1764) */
int  flm_op_rmsnorm(float* o, const float* x, const float* w, size_t n);
/* simd::square_sum (x86_simd.cpp:942-960), n % 16 == 0, n <= 16384: out6 = { total from the speculative wave evaluation the
 * rmsnorm prologue uses, total from the plain sequential chains, the 4 strided partial sums }; the two totals must be the same bits */
int  flm_op_square_sum(const float* x, size_t n, float* out6);
/* sample_argmax (sampler.cpp:36-47): first maximum wins */
int  flm_op_argmax(const float* logits, int n, int32_t* idx);
/* simd::swiglu(xo,xr,n) (x86_simd.cpp:1766-1770) */
int  flm_op_swiglu(float* xo, const float* xr, size_t n);
/* rope_v2 (tf_operators.cpp:352-402): one head row of n_dims at position pos */
int  flm_op_rope(float* o, const float* x, int n_dims, int pos);
/* softmax_sisd over the first n entries (tf_operators.cpp:176-186) */
int  flm_op_softmax(float* x, int n);
/* the ATTN task (execute_attn, transformer.cpp:397-455) for n_heads heads of one new token at
 * position pos: q,k,v are [n_heads*hs]; kc,vc are [n_heads][max_seq][hs] caches (updated);
 * out [n_heads*hs]. */
int  flm_op_attention(float* out, float* kc, float* vc, const float* q, const float* k, const float* v,
                      int n_heads, int hs, int max_seq, int pos);
/* libm expf as the device evaluates it (the reference calls glibc expf in softmax_sisd and swiglu);
 * in place over n floats.  Lets the tests pin the device routine against glibc bit for bit. */
int  flm_op_expf(float* x, size_t n);
/* elementary fp32 functions as the kernels evaluate them, in place over x[n]:
 * fn 0 expf(x), 1 sqrtf(x), 2 x / y, 3 rmsnorm scale 1/sqrtf(x/n + 1e-5) with n = (int)y[i],
 * 4 / 5 quant::quantize's element step q(x) = (T)(x / y) (quant_operators.cpp:26-47) the way the prologues evaluate it (4: the group's four divisions share one refined
 *       reciprocal) and as a plain IEEE division (5): x[i] <- q(x) - 1024 q(-x); the two must agree on every input. */
int  flm_op_math(int fn, float* x, const float* y, size_t n);
/* the hand-off protocol of the fused launches on its own (no reference counterpart: it replaces the reference's thread-pool task barrier,
 * src/components/threadparallel.hpp, inside one GPU launch): `rounds` publish -> flag -> poll -> coherent-read rounds between one workgroup per CU,
 * every value read checked.  *wrong_values / *timed_out must both come back 0. */
int  flm_op_handoff_litmus(int rounds, int* wrong_values, int* timed_out);

/* ---- tensor-parallel shard plan (pure host arithmetic, no GPU needed).
 * Every matmul is split by OUTPUT ROWS, as the reference splits them over its worker threads
 * (split_rows, transformer.cpp:264-287): each output row is reduced on one rank in the reference's
 * order, so a sharded run is bit-identical to the single-GPU / CPU run.  Ranks exchange activation
 * slices with all-gathers (attention outputs, the residual stream twice, the FFN hidden vector, the
 * logits), which is why every split is an equal contiguous slice. */
typedef struct flm_shard_plan {
    int32_t head_begin, head_count;        /* attention heads owned: q,k,v rows, KV cache, attention */
    int32_t hidden_begin, hidden_count;    /* rows of W1/W3 owned (slice of the FFN hidden vector) */
    int32_t dim_begin, dim_count;          /* rows of Wo and W2 owned (slice of the residual stream) */
    int32_t vocab_begin, vocab_count;      /* classifier rows */
} flm_shard_plan;
int  flm_plan_shards(const flm_model_desc* desc, int rank, int world, flm_shard_plan* out);

#ifdef __cplusplus
}
#endif
#endif
