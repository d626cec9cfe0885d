/* oracle/flm_oracle.c -- TEST INFRASTRUCTURE ONLY (see flm_oracle.h).
 *
 * CPU restatement of the reference hot path (CoderLSF/fast-llama, paths relative to
 * /root/reference).  Written to reproduce the reference's *arithmetic order*, including the FMA
 * contraction GCC applies to the reference build (-O3 -mfma, -ffp-contract=fast), so that it can be
 * pinned against the reference bit-for-bit where the reference is deterministic.
 *
 * Build: oracle/Makefile (gcc -O3 -march=x86-64-v3 -mfma -fopenmp).  OpenMP only parallelises over
 * independent output rows; it never changes a summation order.
 */
#include "flm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_F8  127.0f   /* QUANT8_FACTOR  src/blas/quant_operators.h:33 */
#define ORC_F16 5792.0f  /* QUANT16_FACTOR src/blas/quant_operators.h:32 */

/* array_max_abs(float) -- src/platforms/arch/x86_simd.cpp:321-343,460-477 (max is order-free) */
static float max_abs_f32(const float* x, size_t n) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) {
        float a = fabsf(x[i]);
        if (a > m) m = a;
    }
    return m;
}

/* quant::quantize<T> -- src/blas/quant_operators.cpp:26-47.
 * r = max|x| / F ; q = (T)(x / r): C float->int conversion = truncation toward zero.
 * All-zero group: r = 0, x/r = NaN; the x86 build converts NaN to 0x80000000 and keeps the low
 * byte/word, i.e. q = 0 -- stated explicitly here. */
void orc_quantize(int qt, void* qx, float* qs, const float* x, size_t n, int gs) {
    if (qt == ORC_QT_NONE) { memcpy(qx, x, sizeof(float) * n); return; }
    const float F = (qt == ORC_QT_INT8) ? ORC_F8 : ORC_F16;
    size_t ng = (n + (size_t)gs - 1) / (size_t)gs;
    for (size_t g = 0; g < ng; ++g) {
        const float* xg = x + g * (size_t)gs;
        size_t gn = n - g * (size_t)gs; if (gn > (size_t)gs) gn = (size_t)gs;
        float r = max_abs_f32(xg, gn) / F;
        qs[g] = r;
        for (size_t j = 0; j < gn; ++j) {
            float t = xg[j] / r;
            int v = (r == 0.f || t != t) ? 0 : (int)t;
            if (qt == ORC_QT_INT8) ((int8_t*)qx)[g * (size_t)gs + j] = (int8_t)v;
            else                   ((int16_t*)qx)[g * (size_t)gs + j] = (int16_t)v;
        }
    }
}

/* quant::dequantize_ -- src/blas/quant_operators.cpp:49-65 : out = q * scale */
void orc_dequantize(int qt, float* out, const void* qx, const float* qs, size_t n, int gs) {
    if (qt == ORC_QT_NONE) { memcpy(out, qx, sizeof(float) * n); return; }
    for (size_t i = 0; i < n; ++i) {
        float r = qs[i / (size_t)gs];
        out[i] = (qt == ORC_QT_INT8) ? ((const int8_t*)qx)[i] * r : ((const int16_t*)qx)[i] * r;
    }
}

static inline int dot_i8(const int8_t* a, const int8_t* b, int n) {
    int s = 0;
    for (int i = 0; i < n; ++i) s += (int)a[i] * (int)b[i];
    return s;
}
static inline int dot_i16(const int16_t* a, const int16_t* b, int n) {
    int s = 0;
    for (int i = 0; i < n; ++i) s += (int)a[i] * (int)b[i];
    return s;
}

/* quant::matmul<T> -- src/blas/quant_operators.cpp:252-284:
 *   out[b][j] = sum over groups g ASCENDING of (sW[j,g]*sX[b,g]) * float(int32 dot of the group),
 *   "o[j] += s * dot" is contracted to one FMA by the reference build. Output layout out[b][j]. */
void orc_matmul_q(int qt, float* out, const void* W, const float* sW, const void* X, const float* sX,
                  int m, int n, int w, int gs) {
    const int sn = (n + gs - 1) / gs;
    for (int b = 0; b < w; ++b) {
#pragma omp parallel for schedule(static)
        for (int j = 0; j < m; ++j) {
            float o = 0.f;
            for (int g = 0; g < sn; ++g) {
                int len = n - g * gs; if (len > gs) len = gs;
                float s = sW[(size_t)j * sn + g] * sX[(size_t)b * sn + g];
                int d;
                if (qt == ORC_QT_INT8)
                    d = dot_i8((const int8_t*)X + (size_t)b * n + g * gs, (const int8_t*)W + (size_t)j * n + g * gs, len);
                else
                    d = dot_i16((const int16_t*)X + (size_t)b * n + g * gs, (const int16_t*)W + (size_t)j * n + g * gs, len);
                o = fmaf(s, (float)d, o);
            }
            out[(size_t)b * m + j] = o;
        }
    }
}

/* NOT a reference function -- the evaluation of a DEVIATION from it (tools/ksplit_eval.py): the Megatron-style
 * K-split of a row's chain over `parts` tensor-parallel ranks.  The n / gs quant groups are dealt to the ranks in contiguous runs
 * (the first `rem` ranks one group more: 172 groups over 8 ranks = 22,22,22,22,21,21,21,21; a group never straddles ranks), every rank
 * runs the reference's chain (quant_operators.cpp:252-284) over ITS groups from 0.f, and the partials are added in rank order
 * 0..parts-1 (what a deterministic all-reduce would do).  parts == 1 is orc_matmul_q. */
void orc_matmul_q_ksplit(int qt, float* out, const void* W, const float* sW, const void* X, const float* sX,
                         int m, int n, int w, int gs, int parts) {
    const int sn = (n + gs - 1) / gs;
    if (parts < 1) parts = 1;
    if (parts > sn) parts = sn;
    for (int b = 0; b < w; ++b) {
#pragma omp parallel for schedule(static)
        for (int j = 0; j < m; ++j) {
            float total = 0.f;
            int g = 0;
            for (int r = 0; r < parts; ++r) {
                const int cnt = sn / parts + (r < sn % parts ? 1 : 0);
                float o = 0.f;
                for (int e = g + cnt; g < e; ++g) {
                    int len = n - g * gs; if (len > gs) len = gs;
                    float s = sW[(size_t)j * sn + g] * sX[(size_t)b * sn + g];
                    int d;
                    if (qt == ORC_QT_INT8)
                        d = dot_i8((const int8_t*)X + (size_t)b * n + g * gs, (const int8_t*)W + (size_t)j * n + g * gs, len);
                    else
                        d = dot_i16((const int16_t*)X + (size_t)b * n + g * gs, (const int16_t*)W + (size_t)j * n + g * gs, len);
                    o = fmaf(s, (float)d, o);
                }
                total = r == 0 ? o : total + o;
            }
            out[(size_t)b * m + j] = total;
        }
    }
}

/* simd::dot_product(float) -> dot_product_avx256 -- src/platforms/arch/x86_simd.cpp:1447-1467,1677-1699
 * 8 strided lane accumulators (mul+add contracted to FMA), then partials summed 0..7, then tail.
 * (n >= 32 takes the 8-lane path; 16 <= n < 32 the 4-lane SSE path; smaller: scalar.) */
float orc_dot_f32(const float* a, const float* b, size_t n) {
    if (n >= 32) {
        float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        size_t i = 0;
        for (; i + 7 < n; i += 8)
            for (int k = 0; k < 8; ++k) l[k] = fmaf(a[i + k], b[i + k], l[k]);
        float t = 0.f;
        for (int k = 0; k < 8; ++k) t += l[k];
        for (; i < n; ++i) t = fmaf(a[i], b[i], t);
        return t;
    } else if (n >= 16) {
        float l[4] = {0, 0, 0, 0};
        size_t i = 0;
        for (; i + 3 < n; i += 4)
            for (int k = 0; k < 4; ++k) l[k] = fmaf(a[i + k], b[i + k], l[k]);
        float t = 0.f;
        for (int k = 0; k < 4; ++k) t += l[k];
        for (; i < n; ++i) t = fmaf(a[i], b[i], t);
        return t;
    }
    float t = 0.f;
    for (size_t i = 0; i < n; ++i) t = fmaf(a[i], b[i], t);
    return t;
}

/* float quant::matmul -- src/blas/quant_operators.cpp:340-348: out[m*j+i] = dot(mat1[i], mat2[j]) */
void orc_matmul_f32(float* out, const float* mat1, const float* mat2, int m, int n, int k) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < k; ++j)
            out[(size_t)m * j + i] = orc_dot_f32(mat1 + (size_t)n * i, mat2 + (size_t)n * j, (size_t)n);
}

/* simd::square_sum(float) -- src/platforms/arch/x86_simd.cpp:1089-1106.  The AVX2 branch is guarded
 * by the misspelt "__AVX2" so the SSE 4-lane kernel square_sum_avx128 (:942-960) always runs. */
float orc_square_sum(const float* x, size_t n) {
    if (n < 16) {
        float s = 0.f;
        for (size_t i = 0; i < n; ++i) s = fmaf(x[i], x[i], s);
        return s;
    }
    float l[4] = {0, 0, 0, 0};
    size_t i = 0;
    for (; i + 3 < n; i += 4)
        for (int k = 0; k < 4; ++k) l[k] = fmaf(x[i + k], x[i + k], l[k]);
    float r = 0.f;
    for (int k = 0; k < 4; ++k) r += l[k];
    for (; i < n; ++i) r = fmaf(x[i], x[i], r);
    return r;
}

/* simd::rmsnorm(o,x,w,n) -- src/platforms/arch/x86_simd.cpp:1754-1764; multiply_avx256 :1360-1372:
 *   r = float(1. / sqrtf(ss/n + 1e-5f)) ; o = (x*w)*r      (n % 8 == 0 on this path) */
void orc_rmsnorm(float* o, const float* x, const float* w, size_t n) {
    float ss = orc_square_sum(x, n);
    const float r = (float)(1. / (double)sqrtf(ss / (float)n + 1e-5f));
    for (size_t i = 0; i < n; ++i) o[i] = (x[i] * w[i]) * r;
}

/* simd::swiglu(xo,xr,n) -- src/platforms/arch/x86_simd.cpp:1766-1770, evaluated in double */
void orc_swiglu(float* xo, const float* xr, size_t n) {
    for (size_t i = 0; i < n; ++i)
        xo[i] = (float)((double)xo[i] / (1. + (double)expf(-xo[i])) * (double)xr[i]);
}

/* softmax_sisd -- src/blas/tf_operators.cpp:176-186 */
void orc_softmax(float* x, int n) {
    float mx = x[0];
    for (int i = 1; i < n; ++i) if (x[i] > mx) mx = x[i];
    float sum = 0.f;
    for (int i = 0; i < n; ++i) { x[i] = expf(x[i] - mx); sum += x[i]; }
    for (int i = 0; i < n; ++i) x[i] /= sum;
}

/* rope_v2 -- src/blas/tf_operators.cpp:352-402 with its constants folded (freq_base 10000,
 * freq_scale 1, ext_factor 0, attn_factor 1, no xPos): interleaved pairs, theta by recurrence. */
void orc_rope(float* o, const float* x, int n_dims, int pos) {
    const float theta_scale = powf(10000.f, -2.0f / n_dims);
    float theta = (float)pos;
    for (int i = 0; i < n_dims; i += 2) {
        float c = cosf(theta), s = sinf(theta);
        theta *= theta_scale;
        const float x0 = x[i], x1 = x[i + 1];
        o[i]     = fmaf(x0, c, -(x1 * s));
        o[i + 1] = fmaf(x0, s, x1 * c);
    }
}

/* batch weighted_sum -- src/blas/tf_operators.cpp:325-350 (row 0 always, rows >=1 skipped when
 * |w| <= min_w; "o += row*w" contracted to FMA) */
void orc_weighted_sum(float* out, const float* matrix, const float* weights, int m, int n, int bs, float min_w) {
    for (int k = 0; k < bs; ++k) {
        float w = weights[(size_t)m * k];
        for (int i = 0; i < n; ++i) out[(size_t)n * k + i] = matrix[i] * w;
    }
    for (int i = 1; i < m; ++i) {
        const float* row = matrix + (size_t)n * i;
        for (int k = 0; k < bs; ++k) {
            float w = weights[(size_t)m * k + i];
            if (fabsf(w) <= min_w) continue;
            float* o = out + (size_t)n * k;
            for (int j = 0; j < n; ++j) o[j] = fmaf(row[j], w, o[j]);
        }
    }
}

/* sample_argmax -- src/transformer/sampler.cpp:36-47 (first maximum wins) */
int orc_argmax(const float* x, int n) {
    int bi = 0; float bv = x[0];
    for (int i = 1; i < n; ++i) if (x[i] > bv) { bv = x[i]; bi = i; }
    return bi;
}

/* execute_attn for one kv head, hgs == 1 -- src/transformer/transformer.cpp:397-455 */
void orc_attention_head(float* out, float* kc, float* vc, const float* q_in, const float* k_in, const float* v_in,
                        int hs, int pos, int bs, float* scratch) {
    const int seqlen = pos + bs;
    float* att = scratch;                         /* [bs][seqlen] */
    float* q = (float*)malloc(sizeof(float) * (size_t)bs * hs);
    for (int i = 0; i < bs; ++i) {
        memcpy(kc + (size_t)(pos + i) * hs, k_in + (size_t)i * hs, sizeof(float) * hs);   /* :431 */
        memcpy(vc + (size_t)(pos + i) * hs, v_in + (size_t)i * hs, sizeof(float) * hs);   /* :432 */
        orc_rope(q + (size_t)i * hs, q_in + (size_t)i * hs, hs, pos + i);                 /* :438 */
        orc_rope(kc + (size_t)(pos + i) * hs, kc + (size_t)(pos + i) * hs, hs, pos + i);  /* :439 */
    }
    const float scale = (float)(1. / (double)sqrtf((float)hs));                            /* :418 */
    orc_matmul_f32(att, kc, q, seqlen, hs, bs);                                            /* :442 */
    for (size_t i = 0; i < (size_t)bs * seqlen; ++i) att[i] *= scale;                      /* :443 */
    for (int i = 0; i < bs; ++i) {                                                         /* :444-448 */
        float* row = att + (size_t)i * seqlen;
        orc_softmax(row, pos + i + 1);
        for (int t = pos + i + 1; t < seqlen; ++t) row[t] = 0.f;
    }
    orc_weighted_sum(out, vc, att, seqlen, hs, bs, 1e-15f);                                /* :449 */
    free(q);
}

/* ------------------------------- model ------------------------------------------------------ */
typedef struct { void* q; float* s; int rows, cols; } orc_qmat;
struct orc_model {
    int dim, hidden, L, H, KVH, V, qt, gs, max_seq, hs, kv_dim;
    int emb_qt; void* emb; float* emb_s;              /* [V][dim] */
    float *att_norm, *ffn_norm, *out_norm;            /* [L][dim], [L][dim], [dim] */
    orc_qmat *wq, *wk, *wv, *wo, *w1, *w2, *w3;       /* per layer */
    orc_qmat cls;
    float *kcache, *vcache;                           /* [L][KVH][max_seq][hs] */
    float* tap_x;
    int ksplit;                                       /* > 1: Wo and W2 run orc_matmul_q_ksplit (a deviation under evaluation, never the parity path) */
};

static size_t esz(int qt) { return qt == ORC_QT_INT8 ? 1 : (qt == ORC_QT_INT16 ? 2 : 4); }

orc_model* orc_model_create(int dim, int hidden, int L, int H, int KVH, int V, int qt, int gs, int max_seq) {
    if (H < 1 || KVH != H || dim % H || dim % gs || hidden % gs || (qt != ORC_QT_INT8 && qt != ORC_QT_INT16)) return NULL;
    orc_model* m = (orc_model*)calloc(1, sizeof(*m));
    m->dim = dim; m->hidden = hidden; m->L = L; m->H = H; m->KVH = KVH; m->V = V; m->qt = qt; m->gs = gs;
    m->max_seq = max_seq; m->hs = dim / H; m->kv_dim = m->hs * KVH;
    m->att_norm = (float*)calloc((size_t)L * dim, 4); m->ffn_norm = (float*)calloc((size_t)L * dim, 4);
    m->out_norm = (float*)calloc(dim, 4);
    m->wq = calloc(L, sizeof(orc_qmat)); m->wk = calloc(L, sizeof(orc_qmat)); m->wv = calloc(L, sizeof(orc_qmat));
    m->wo = calloc(L, sizeof(orc_qmat)); m->w1 = calloc(L, sizeof(orc_qmat)); m->w2 = calloc(L, sizeof(orc_qmat));
    m->w3 = calloc(L, sizeof(orc_qmat));
    size_t kvn = (size_t)L * KVH * max_seq * m->hs;
    m->kcache = (float*)calloc(kvn, 4); m->vcache = (float*)calloc(kvn, 4);
    m->tap_x = (float*)calloc(dim, 4);
    return m;
}
static void qmat_free(orc_qmat* q) { if (q) { free(q->q); free(q->s); } }
void orc_model_free(orc_model* m) {
    if (!m) return;
    for (int l = 0; l < m->L; ++l) { qmat_free(&m->wq[l]); qmat_free(&m->wk[l]); qmat_free(&m->wv[l]); qmat_free(&m->wo[l]);
                                     qmat_free(&m->w1[l]); qmat_free(&m->w2[l]); qmat_free(&m->w3[l]); }
    qmat_free(&m->cls);
    free(m->wq); free(m->wk); free(m->wv); free(m->wo); free(m->w1); free(m->w2); free(m->w3);
    free(m->emb); free(m->emb_s); free(m->att_norm); free(m->ffn_norm); free(m->out_norm);
    free(m->kcache); free(m->vcache); free(m->tap_x); free(m);
}
void orc_model_set_ksplit(orc_model* m, int parts) { m->ksplit = parts; }
void orc_model_reset(orc_model* m) {
    size_t kvn = (size_t)m->L * m->KVH * m->max_seq * m->hs;
    memset(m->kcache, 0, kvn * 4); memset(m->vcache, 0, kvn * 4);
}

/* A13: fp32 source weights are quantized row-block-wise with the same quantize (copy_layers,
 * transformer.cpp:289-304); groups never straddle rows since cols % gs == 0. */
static int set_qmat(orc_model* m, orc_qmat* d, int src_qt, const void* data, const float* scales, int rows, int cols) {
    size_t n = (size_t)rows * cols;
    qmat_free(d);
    d->rows = rows; d->cols = cols;
    d->q = malloc(n * esz(m->qt)); d->s = (float*)malloc(n / m->gs * 4);
    if (src_qt == ORC_QT_NONE) orc_quantize(m->qt, d->q, d->s, (const float*)data, n, m->gs);
    else if (src_qt == m->qt) { memcpy(d->q, data, n * esz(m->qt)); memcpy(d->s, scales, n / m->gs * 4); }
    else return -1;
    return 0;
}

int orc_model_set_tensor(orc_model* m, int kind, int layer, int src_qt, const void* data, const float* scales, int rows, int cols) {
    if (layer < 0 || (kind >= 16 && layer >= m->L)) return -1;
    switch (kind) {
    case ORC_T_TOKEN_EMBD: {
        if (rows != m->V || cols != m->dim) return -2;
        size_t n = (size_t)rows * cols;
        free(m->emb); free(m->emb_s); m->emb_s = NULL; m->emb_qt = src_qt;
        m->emb = malloc(n * esz(src_qt)); memcpy(m->emb, data, n * esz(src_qt));
        if (src_qt != ORC_QT_NONE) { m->emb_s = (float*)malloc(n / m->gs * 4); memcpy(m->emb_s, scales, n / m->gs * 4); }
        return 0; }
    case ORC_T_OUTPUT_NORM: memcpy(m->out_norm, data, (size_t)m->dim * 4); return 0;
    case ORC_T_INPUT_NORM:  memcpy(m->att_norm + (size_t)layer * m->dim, data, (size_t)m->dim * 4); return 0;
    case ORC_T_POST_NORM:   memcpy(m->ffn_norm + (size_t)layer * m->dim, data, (size_t)m->dim * 4); return 0;
    case ORC_T_CLASSIFIER:  if (rows != m->V || cols != m->dim) return -2; return set_qmat(m, &m->cls, src_qt, data, scales, rows, cols);
    case ORC_T_ATTN_Q: if (rows != m->dim || cols != m->dim) return -2;    return set_qmat(m, &m->wq[layer], src_qt, data, scales, rows, cols);
    case ORC_T_ATTN_K: if (rows != m->kv_dim || cols != m->dim) return -2; return set_qmat(m, &m->wk[layer], src_qt, data, scales, rows, cols);
    case ORC_T_ATTN_V: if (rows != m->kv_dim || cols != m->dim) return -2; return set_qmat(m, &m->wv[layer], src_qt, data, scales, rows, cols);
    case ORC_T_ATTN_O: if (rows != m->dim || cols != m->dim) return -2;    return set_qmat(m, &m->wo[layer], src_qt, data, scales, rows, cols);
    case ORC_T_MLP_GATE: if (rows != m->hidden || cols != m->dim) return -2; return set_qmat(m, &m->w1[layer], src_qt, data, scales, rows, cols);
    case ORC_T_MLP_UP:   if (rows != m->hidden || cols != m->dim) return -2; return set_qmat(m, &m->w3[layer], src_qt, data, scales, rows, cols);
    case ORC_T_MLP_DOWN: if (rows != m->dim || cols != m->hidden) return -2; return set_qmat(m, &m->w2[layer], src_qt, data, scales, rows, cols);
    default: return -3;
    }
}

const float* orc_model_tap_x(orc_model* m) { return m->tap_x; }
const float* orc_model_kcache(orc_model* m, int layer) { return m->kcache + (size_t)layer * m->KVH * m->max_seq * m->hs; }
const float* orc_model_vcache(orc_model* m, int layer) { return m->vcache + (size_t)layer * m->KVH * m->max_seq * m->hs; }

/* ParallelTransformer::forward -- src/transformer/transformer.cpp:105-161 (line refs inline) */
int orc_model_forward(orc_model* m, const int32_t* tokens, int n, int pos, float* logits) {
    const int dim = m->dim, hid = m->hidden, hs = m->hs, kvd = m->kv_dim, gs = m->gs, qt = m->qt;
    if (n < 1 || pos < 0 || pos + n > m->max_seq || !m->emb || !m->cls.q) return -1;
    int bs = n;
    const int qkv_w = dim + 2 * kvd;
    float* x1  = (float*)malloc(sizeof(float) * (size_t)bs * dim);
    float* x2  = (float*)malloc(sizeof(float) * (size_t)bs * dim);
    float* qkv = (float*)malloc(sizeof(float) * (size_t)bs * qkv_w);
    float* hd  = (float*)malloc(sizeof(float) * (size_t)bs * hid);
    float* h3  = (float*)malloc(sizeof(float) * (size_t)bs * hid);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)bs * (dim > hid ? dim : hid));
    void*  qx  = malloc((size_t)bs * (dim > hid ? dim : hid) * esz(qt));
    float* sx  = (float*)malloc(sizeof(float) * (size_t)bs * (dim > hid ? dim : hid) / gs);
    float* att = (float*)malloc(sizeof(float) * (size_t)bs * (pos + bs));
    float* hq  = (float*)malloc(sizeof(float) * (size_t)bs * hs * 4);

    for (int i = 0; i < bs; ++i) {                                                     /* :115-122 */
        int t = tokens[i];
        if (t < 0 || t >= m->V) return -2;
        if (m->emb_qt == ORC_QT_NONE) memcpy(x1 + (size_t)i * dim, (float*)m->emb + (size_t)t * dim, sizeof(float) * dim);
        else orc_dequantize(m->emb_qt, x1 + (size_t)i * dim, (char*)m->emb + (size_t)t * dim * esz(m->emb_qt),
                            m->emb_s + (size_t)t * dim / gs, dim, gs);
    }
    for (int l = 0; l < m->L; ++l) {
        for (int i = 0; i < bs; ++i) orc_rmsnorm(x2 + (size_t)i * dim, x1 + (size_t)i * dim, m->att_norm + (size_t)l * dim, dim); /* :132 */
        orc_quantize(qt, qx, sx, x2, (size_t)bs * dim, gs);                            /* :134 */
        /* QKV task :135, execute_qkv :386-395 -> qkv[b] = Q | K | V */
        orc_matmul_q(qt, tmp, m->wq[l].q, m->wq[l].s, qx, sx, dim, dim, bs, gs);
        for (int i = 0; i < bs; ++i) memcpy(qkv + (size_t)i * qkv_w, tmp + (size_t)i * dim, sizeof(float) * dim);
        orc_matmul_q(qt, tmp, m->wk[l].q, m->wk[l].s, qx, sx, kvd, dim, bs, gs);
        for (int i = 0; i < bs; ++i) memcpy(qkv + (size_t)i * qkv_w + dim, tmp + (size_t)i * kvd, sizeof(float) * kvd);
        orc_matmul_q(qt, tmp, m->wv[l].q, m->wv[l].s, qx, sx, kvd, dim, bs, gs);
        for (int i = 0; i < bs; ++i) memcpy(qkv + (size_t)i * qkv_w + dim + kvd, tmp + (size_t)i * kvd, sizeof(float) * kvd);
        /* ATTN task :136, execute_attn :397-455 */
        for (int h = 0; h < m->KVH; ++h) {
            float* kc = m->kcache + (((size_t)l * m->KVH + h) * m->max_seq) * hs;
            float* vc = m->vcache + (((size_t)l * m->KVH + h) * m->max_seq) * hs;
            float* hq_q = hq; float* hq_k = hq + (size_t)bs * hs; float* hq_v = hq + 2 * (size_t)bs * hs; float* hq_o = hq + 3 * (size_t)bs * hs;
            for (int i = 0; i < bs; ++i) {
                memcpy(hq_q + (size_t)i * hs, qkv + (size_t)i * qkv_w + (size_t)hs * h, sizeof(float) * hs);
                memcpy(hq_k + (size_t)i * hs, qkv + (size_t)i * qkv_w + dim + (size_t)hs * h, sizeof(float) * hs);
                memcpy(hq_v + (size_t)i * hs, qkv + (size_t)i * qkv_w + dim + kvd + (size_t)hs * h, sizeof(float) * hs);
            }
            orc_attention_head(hq_o, kc, vc, hq_q, hq_k, hq_v, hs, pos, bs, att);
            for (int i = 0; i < bs; ++i) memcpy(x2 + (size_t)i * dim + (size_t)hs * h, hq_o + (size_t)i * hs, sizeof(float) * hs);
        }
        orc_quantize(qt, qx, sx, x2, (size_t)bs * dim, gs);                            /* :138 */
        if (m->ksplit > 1) orc_matmul_q_ksplit(qt, tmp, m->wo[l].q, m->wo[l].s, qx, sx, dim, dim, bs, gs, m->ksplit);
        else orc_matmul_q(qt, tmp, m->wo[l].q, m->wo[l].s, qx, sx, dim, dim, bs, gs);       /* :139, :457-466 */
        for (size_t i = 0; i < (size_t)bs * dim; ++i) x1[i] += tmp[i];
        if (bs > 1 && l == m->L - 1) {                                                 /* :140-142 */
            memmove(x1, x1 + (size_t)(bs - 1) * dim, sizeof(float) * dim);
            bs = 1;
        }
        for (int i = 0; i < bs; ++i) orc_rmsnorm(x2 + (size_t)i * dim, x1 + (size_t)i * dim, m->ffn_norm + (size_t)l * dim, dim); /* :144 */
        orc_quantize(qt, qx, sx, x2, (size_t)bs * dim, gs);                            /* :146 */
        orc_matmul_q(qt, hd, m->w1[l].q, m->w1[l].s, qx, sx, hid, dim, bs, gs);        /* :147, :468-483 */
        orc_matmul_q(qt, h3, m->w3[l].q, m->w3[l].s, qx, sx, hid, dim, bs, gs);
        orc_swiglu(hd, h3, (size_t)bs * hid);
        orc_quantize(qt, qx, sx, hd, (size_t)bs * hid, gs);                            /* :149 */
        if (m->ksplit > 1) orc_matmul_q_ksplit(qt, tmp, m->w2[l].q, m->w2[l].s, qx, sx, dim, hid, bs, gs, m->ksplit);
        else orc_matmul_q(qt, tmp, m->w2[l].q, m->w2[l].s, qx, sx, dim, hid, bs, gs);       /* :150, :485-494 */
        for (size_t i = 0; i < (size_t)bs * dim; ++i) x1[i] += tmp[i];
    }
    float* xl = x1 + (size_t)(bs - 1) * dim;                                           /* :154 */
    memcpy(m->tap_x, xl, sizeof(float) * dim);
    orc_rmsnorm(xl, xl, m->out_norm, dim);                                             /* :155 */
    orc_quantize(qt, qx, sx, xl, dim, gs);                                             /* :156 */
    orc_matmul_q(qt, logits, m->cls.q, m->cls.s, qx, sx, m->V, dim, 1, gs);            /* :160, :496-505 */
    free(x1); free(x2); free(qkv); free(hd); free(h3); free(tmp); free(qx); free(sx); free(att); free(hq);
    return 0;
}
