// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A thin extern "C" shim over the *real* reference (CoderLSF/fast-llama), compiled against the
// sources where they lie under /root/reference by oracle/Makefile -> oracle/_ref/libflref.so.
// It exists so that (a) the C restatement in oracle/flm_oracle.c can be pinned against the
// reference itself, and (b) tests/golden/make_golden.py can generate golden vectors.
// Nothing here is shipped; the product (fast-llama_amd/csrc) never links or loads this.
//
// Every entry point forwards 1:1 to the reference function named in its comment.

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <span>
#include <string>
#include <string_view>
#include <vector>
#include <condition_variable>
#include <thread>
#include <atomic>
#include <iostream>
#include <fstream>
#include <sstream>

#include "quant_operators.h"   // cpuft::quant::*   (src/blas/quant_operators.h:37-82)
#include "tf_operators.h"      // cpuft::softmax_sisd, rope_v2, weighted_sum (src/blas/tf_operators.h)
#include "simd.h"              // cpuft::simd::*    (src/platforms/arch/simd.h:13-63)

// ParallelTransformer::forward is private (src/transformer/transformer.h:99); the harness needs
// per-step logits, so this one TU is compiled with -fno-access-control (see oracle/Makefile).
#include "transformer.h"

using namespace cpuft;

extern "C" {

// quant::quantize  (src/blas/quant_operators.cpp:78-97). qt: 1=INT16, 2=INT8 (QuantType enum)
void ref_quantize(int qt, void* qx, float* qs, const float* x, size_t n, int gs) {
    quant::quantize(quant::QuantType(qt), qx, qs, x, n, gs);
}
// quant::dequantize (src/blas/quant_operators.cpp:99-117)
void ref_dequantize(int qt, float* out, const void* qx, const float* qs, size_t n, int gs) {
    quant::dequantize(quant::QuantType(qt), out, qx, qs, n, gs);
}
// quant::matmul (src/blas/quant_operators.cpp:571-591); qt=0 -> float matmul (:340-348)
void ref_matmul(int qt, float* out, const void* m1, const float* s1, const void* m2, const float* s2,
                int m, int n, int w, int gs) {
    quant::matmul(quant::QuantType(qt), out, m1, s1, m2, s2, m, n, w, gs);
}
// quant::mul (src/blas/quant_operators.cpp:425-442), float path == Tensor::multiply
void ref_mul(float* x, float factor, size_t n) {
    quant::mul(quant::QuantType::NONE, x, factor, nullptr, n, 64);
}
// simd::rmsnorm(o,x,w,n) (src/platforms/arch/x86_simd.cpp:1754-1764)
void ref_rmsnorm(float* o, const float* x, const float* w, size_t n) { simd::rmsnorm(o, x, w, n); }
// simd::swiglu(xo,xr,n) (src/platforms/arch/x86_simd.cpp:1766-1770)
void ref_swiglu(float* xo, const float* xr, size_t n) { simd::swiglu(xo, xr, n); }
// simd::add(x1,x2,n) (src/platforms/arch/x86_simd.cpp:1269-1286)
void ref_add(float* x1, const float* x2, size_t n) { simd::add(x1, x2, n); }
// softmax_sisd (src/blas/tf_operators.cpp:176-186)
void ref_softmax(float* x, int n) { cpuft::softmax_sisd(x, n); }
// rope_v2 (src/blas/tf_operators.cpp:352-402)
void ref_rope_v2(float* o, const float* x, int n_dims, int n_orig_ctx, int pos) {
    cpuft::rope_v2(o, x, n_dims, n_orig_ctx, pos, 0, 1);
}
// batch weighted_sum (src/blas/tf_operators.cpp:325-350)
void ref_weighted_sum(float* out, const float* matrix, const float* weights, int m, int n, int bs, float min_w) {
    cpuft::weighted_sum(out, matrix, weights, m, n, bs, min_w);
}
float ref_dot_f32(const float* a, const float* b, size_t n) { return simd::dot_product(a, b, n); }
float ref_square_sum(const float* a, size_t n) { return simd::square_sum(a, n); }
float ref_array_max(const float* a, size_t n) { return simd::array_max(a, n); }
float ref_array_max_abs(const float* a, size_t n) { return simd::array_max_abs(a, n); }
size_t ref_simd_size() { return simd::get_simd_size(); }

// ---- model level: ParallelTransformer (src/transformer/transformer.h:76-99) --------------------
struct RefModel {
    ParallelTransformer tf{false};
};

void* ref_model_load(const char* ckpt, const char* tknr, int qt, int num_threads, int max_batch) {
    auto* m = new RefModel();
    // load(ckpt, tknr, mft, qt, num_threads, use_numa, max_batch_size, seed) transformer.cpp:23-42
    bool ok = m->tf.load(ckpt, tknr ? tknr : "", ModelFileType::UNKNOWN, quant::QuantType(qt),
                         num_threads, false, max_batch, 0);
    if (!ok) { delete m; return nullptr; }
    return m;
}
void ref_model_free(void* h) { delete reinterpret_cast<RefModel*>(h); }

int ref_model_vocab(void* h)  { return reinterpret_cast<RefModel*>(h)->tf._tfc.vocab_size; }
int ref_model_dim(void* h)    { return reinterpret_cast<RefModel*>(h)->tf._tfc.dim; }
int ref_model_layers(void* h) { return reinterpret_cast<RefModel*>(h)->tf._tfc.n_layers; }
int ref_model_qtype(void* h)  { return int(reinterpret_cast<RefModel*>(h)->tf._tfc.quant_type); }

// forward(tokens, pos, logits) (transformer.cpp:105-161); copies logits[vocab] out.
int ref_model_forward(void* h, const int* tokens, int n, int pos, float* logits_out) {
    auto* m = reinterpret_cast<RefModel*>(h);
    Tensor logits;
    m->tf.forward(std::span<const int>(tokens, size_t(n)), pos, logits);
    memcpy(logits_out, logits.float_data(), sizeof(float) * size_t(m->tf._tfc.vocab_size));
    return 0;
}
// sampler (src/transformer/sampler.cpp:113-137)
int ref_model_sample(void* h, float* logits, float temperature, float topp) {
    auto* m = reinterpret_cast<RefModel*>(h);
    Tensor t;
    t.reset(m->tf._tfc.vocab_size);
    t.manage(logits);
    return m->tf._sampler.sample(t, temperature, topp);
}
// tokenizer (src/transformer/tokenizer.cpp:247-342)
int ref_model_encode(void* h, const char* text, int* out, int cap) {
    auto v = reinterpret_cast<RefModel*>(h)->tf.encode(text);
    int n = int(v.size()) < cap ? int(v.size()) : cap;
    memcpy(out, v.data(), sizeof(int) * size_t(n));
    return int(v.size());
}
int ref_model_decode(void* h, const int* tokens, int n, char* out, int cap) {
    auto s = reinterpret_cast<RefModel*>(h)->tf.decode(std::span<const int>(tokens, size_t(n)));
    snprintf(out, size_t(cap), "%s", s.c_str());
    return int(s.size());
}

} // extern "C"
