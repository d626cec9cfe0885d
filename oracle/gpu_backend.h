// oracle/gpu_backend.h -- the REFERENCE-SIDE binding of include/flm_gpu.h that INTEGRATION.md section 2 describes: the file a
// fast-llama maintainer would add as src/transformer/gpu_backend.h.  It lives under oracle/ because it is compiled only
// against the reference's own headers (oracle/Makefile target `ref`, build container) to PROVE that the boundary binds:
// oracle/gpu_backend_check.cpp loads a model with the reference's loader, runs the reference's forward and this
// binding's forward on the same tokens and compares the logits bit for bit.  Not part of the product.
#pragma once
#include <algorithm>
#include <span>

#include "flm_gpu.h"                 // this repo: include/flm_gpu.h
#include "model_loader.h"            // reference: src/model_loaders/model_loader.h
#include "log.h"                     // reference: src/utils/log.h

namespace cpuft {
class GpuBackend {
public:
    ~GpuBackend() { if (_ctx) flm_ctx_destroy(_ctx); }

    // replaces ThreadParallel init + parallel_global_init / parallel_thread_init (transformer.cpp:209-384).
    // qt: the -q type, used when the file carries no quantization (transformer.cpp:36-38)
    bool init(const TransformerModel& tf, int device, QuantType qt = QuantType::INT8) {
        const auto& c = tf.conf;
        const QuantType use = c.quant_type != QuantType::NONE ? c.quant_type : qt;
        flm_model_desc d{c.dim, c.hidden_dim, c.n_layers, c.n_heads, c.n_kv_heads, c.vocab_size,
                         c.max_seq_len > 1024 ? 1024 : c.max_seq_len /* transformer.cpp:32 */, int(use), c.quant_group_size};
        if (flm_ctx_create(&d, device, 0, 1, nullptr, &_ctx) != FLM_OK) { _err = flm_last_error(nullptr); return false; }
        auto put = [&](int kind, const Tensor& t, int rows, int cols) {       // one call per layer (Tensor::operator[] = layer slice)
            const int L = t.layers() > 1 ? t.layers() : 1;
            for (int l = 0; l < L; ++l) {
                const Tensor s = t.layers() > 1 ? t[l] : t;
                if (flm_upload_tensor(_ctx, kind, l, int(s.quant_type()), s.data(), s.is_quantized() ? s.scales() : nullptr, rows, cols) != FLM_OK) {
                    _err = flm_last_error(_ctx); return false;
                }
            }
            return true;
        };
        const auto& w = tf.weights;
        return put(FLM_T_TOKEN_EMBD, w.token_embedding_table, c.vocab_size, c.dim)
            && put(FLM_T_INPUT_NORM, w.attn_norm, 1, c.dim)      && put(FLM_T_ATTN_Q, w.attn_q, c.dim, c.dim)
            && put(FLM_T_ATTN_K, w.attn_k, c.kv_dim, c.dim)      && put(FLM_T_ATTN_V, w.attn_v, c.kv_dim, c.dim)
            && put(FLM_T_ATTN_O, w.attn_o, c.dim, c.dim)         && put(FLM_T_POST_NORM, w.ffn_norm, 1, c.dim)
            && put(FLM_T_MLP_GATE, w.ffn_1, c.hidden_dim, c.dim) && put(FLM_T_MLP_DOWN, w.ffn_2, c.dim, c.hidden_dim)
            && put(FLM_T_MLP_UP, w.ffn_3, c.hidden_dim, c.dim)   && put(FLM_T_OUTPUT_NORM, w.out_norm, 1, c.dim)
            && put(FLM_T_CLASSIFIER, w.classifier.data() ? w.classifier : w.token_embedding_table, c.vocab_size, c.dim);
    }

    // replaces ParallelTransformer::forward (transformer.cpp:105-161): logits of the last token, on the host
    bool forward(std::span<const int> tokens, int pos, float* logits) {
        if (flm_forward(_ctx, tokens.data(), int(tokens.size()), pos, logits) == FLM_OK) return true;
        _err = flm_last_error(_ctx); return false;
    }
    // temperature 0: the whole decode loop stays on the device (the body of generate, transformer.cpp:92-101)
    bool decode_greedy(int first, int pos, int n, int* out) { return flm_decode_greedy(_ctx, first, pos, n, out) == FLM_OK; }
    const char* error() const { return _err ? _err : ""; }

private:
    flm_ctx* _ctx = nullptr;
    const char* _err = nullptr;
};
} // namespace cpuft
