/* oracle/flm_oracle.h -- TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by, or called from
 * the product (fast-llama_amd/csrc, the CLI, or the C-ABI library).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may use it, and only as the checker.
 *
 * Plain-C restatement of the reference's (CoderLSF/fast-llama) per-token hot path.  Every function
 * cites the reference file:line it restates (paths relative to /root/reference).
 *
 * Parity pin: validated bit-for-bit / to <=1e-6 against the reference itself (oracle/_ref/libflref.so,
 * built from the reference sources by oracle/Makefile) in tests/test_oracle_vs_reference.py, and
 * against the committed golden vectors in tests/golden/ (generated from libflref.so by
 * tests/golden/make_golden.py).  The reference ships no tests of its own.
 */
#ifndef FLM_ORACLE_H
#define FLM_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* QuantType numbering == cpuft::quant::QuantType (src/blas/quant_operators.h:18-25) */
enum { ORC_QT_NONE = 0, ORC_QT_INT16 = 1, ORC_QT_INT8 = 2 };

/* tensor kinds == .flm TensorType (src/model_loaders/flm_loader.cpp:50-67) */
enum {
    ORC_T_TOKEN_EMBD = 1, ORC_T_OUTPUT_NORM = 2, ORC_T_CLASSIFIER = 3,
    ORC_T_INPUT_NORM = 17, ORC_T_ATTN_Q = 18, ORC_T_ATTN_K = 19, ORC_T_ATTN_V = 20, ORC_T_ATTN_O = 21,
    ORC_T_MLP_GATE = 22 /* ffn_1 */, ORC_T_MLP_UP = 23 /* ffn_3 */, ORC_T_MLP_DOWN = 24 /* ffn_2 */,
    ORC_T_POST_NORM = 25
};

/* ---- operators (A2..A12 of SURVEY.md section 8a) ------------------------------------------- */
void  orc_quantize(int qt, void* qx, float* qs, const float* x, size_t n, int gs);
void  orc_dequantize(int qt, float* out, const void* qx, const float* qs, size_t n, int gs);
void  orc_matmul_q(int qt, float* out, const void* W, const float* sW, const void* X, const float* sX,
                   int m, int n, int w, int gs);
void  orc_matmul_f32(float* out, const float* mat1, const float* mat2, int m, int n, int k);
float orc_dot_f32(const float* a, const float* b, size_t n);
float orc_square_sum(const float* x, size_t n);
void  orc_rmsnorm(float* o, const float* x, const float* w, size_t n);
void  orc_swiglu(float* xo, const float* xr, size_t n);
void  orc_softmax(float* x, int n);
void  orc_rope(float* o, const float* x, int n_dims, int pos);
void  orc_weighted_sum(float* out, const float* matrix, const float* weights, int m, int n, int bs, float min_w);
int   orc_argmax(const float* x, int n);
/* the ATTN task for one head (execute_attn, transformer.cpp:397-455), hgs == 1 */
void  orc_attention_head(float* out /*[bs][hs]*/, float* kc /*[max_seq][hs]*/, float* vc,
                         const float* q /*[bs][hs]*/, const float* k, const float* v,
                         int hs, int pos, int bs, float* scratch /* bs*(pos+bs) floats */);

/* ---- model level (A0: ParallelTransformer::forward, transformer.cpp:105-161) ---------------- */
typedef struct orc_model orc_model;
orc_model* orc_model_create(int dim, int hidden_dim, int n_layers, int n_heads, int n_kv_heads,
                            int vocab_size, int qt, int gs, int max_seq_len);
void       orc_model_free(orc_model* m);
/* data: fp32 (src_qt==NONE) or already-quantized values + scales.  The model keeps its own copy.
 * Linear-layer tensors given as fp32 are quantized here exactly like parallel_thread_init does
 * (transformer.cpp:289-304, A13).  rows/cols describe ONE layer's matrix. */
int        orc_model_set_tensor(orc_model* m, int kind, int layer, int src_qt, const void* data,
                                const float* scales, int rows, int cols);
/* tokens[n] at absolute position pos -> logits[vocab] of the LAST token.  0 on success. */
int        orc_model_forward(orc_model* m, const int32_t* tokens, int n, int pos, float* logits);
void       orc_model_reset(orc_model* m);
/* evaluation of the K-split deviation (tools/ksplit_eval.py; NOT the reference's arithmetic): Wo and W2 as `parts` partial chains summed in rank order */
void       orc_model_set_ksplit(orc_model* m, int parts);
void       orc_matmul_q_ksplit(int qt, float* out, const void* W, const float* sW, const void* X, const float* sX,
                               int m, int n, int w, int gs, int parts);
/* debugging taps: copies of intermediate activations of the last forward (last row) */
const float* orc_model_tap_x(orc_model* m);   /* residual stream after the last layer [dim] */
const float* orc_model_kcache(orc_model* m, int layer);   /* [heads][max_seq][hs] */
const float* orc_model_vcache(orc_model* m, int layer);

#ifdef __cplusplus
}
#endif
#endif
