#include "engine.h"

#include <stdio.h>

#include <iomanip>
#include <iostream>

namespace flmhost {

GpuTransformer::~GpuTransformer() { if (_ctx) flm_ctx_destroy(_ctx); }

bool GpuTransformer::load(const std::string& ckpt, const std::string& tknr, FileType ft, int qtype, int device, uint64_t seed) {
    ModelFile mf;
    if (!load_model_file(ckpt, tknr, ft, false, _debug, mf, _err)) return false;
    _cfg = mf.cfg;
    _cfg.max_seq_len = 1024;                                   // transformer.cpp:32
    if (_cfg.quant_type == 0) _cfg.quant_type = qtype;         // the file's quant type wins over -q (transformer.cpp:36-38)
    _tok.set_vocab(std::move(mf.vocab));
    _sampler.build(_cfg.vocab_size, seed);
    flm_model_desc d{};
    d.dim = _cfg.dim; d.hidden_dim = _cfg.hidden_dim; d.n_layers = _cfg.n_layers; d.n_heads = _cfg.n_heads; d.n_kv_heads = _cfg.n_kv_heads;
    d.vocab_size = _cfg.vocab_size; d.max_seq_len = _cfg.max_seq_len; d.quant_type = _cfg.quant_type; d.quant_group_size = _cfg.quant_group_size;
    int rc = flm_ctx_create(&d, device, 0, 1, nullptr, &_ctx);
    if (rc != FLM_OK) { _err = std::string("GPU context: ") + flm_last_error(nullptr); return false; }
    for (const HostTensor& t : mf.tensors) {
        // a quantized tensor of another type than the model's cannot be multiplied (tensor.cpp:557-561)
        rc = flm_upload_tensor(_ctx, t.kind, t.layer, t.qtype, t.values, t.scales, t.rows, t.cols);
        if (rc != FLM_OK) { _err = std::string("upload: ") + flm_last_error(_ctx); return false; }
    }
    return true;
}

std::vector<int> GpuTransformer::encode(const char* prompt) const {
    if (!prompt || !prompt[0]) return {};
    auto v = _tok.encode(prompt, true);
    if ((int)v.size() > _cfg.max_seq_len) v.resize(_cfg.max_seq_len);
    return v;
}

void print_vector(const char* title, const std::vector<int>& vec) {       // utility.cpp:67-94
    int dw = 1;
    if (!vec.empty()) {
        int mn = vec[0], mx = vec[0];
        for (int v : vec) { if (v > mx) mx = v; else if (v < mn) mn = v; }
        for (int v = 10; v <= mx; v *= 10, ++dw) {}
        if (mn < 0 && -mn > mx) { dw = 2; for (int v = -10; v >= mn; v *= 10, ++dw) {} }
    }
    std::cout << title << "[";
    for (size_t i = 0; i < vec.size(); ++i) { if (i) std::cout << ", "; std::cout << std::setw(dw) << vec[i]; }
    std::cout << "]" << std::endl;
}

bool GpuTransformer::generate(const char* prompt, const std::function<bool(const char*, int, int, bool)>& cb,
                              int max_new_tokens, float temperature, float topp) {
    const std::vector<int> input = encode(prompt);
    if (input.empty()) { fprintf(stderr, "Empty input for generate()\n"); return false; }
    printf("Input prompt:%s\n", prompt);
    print_vector("Input tokens:", input);
    const int n_in = (int)input.size();
    if (n_in >= _cfg.max_seq_len) { fprintf(stderr, "Input is too long for generate()\n"); return false; }
    if (max_new_tokens > _cfg.max_seq_len - n_in) max_new_tokens = _cfg.max_seq_len - n_in;
    const int max_tokens = n_in + max_new_tokens;

    int prev = -1;
    auto emit = [&](int tok, int index, int n_cur) {
        const std::string piece = _tok.decode(tok, prev);
        prev = tok;
        return cb(piece.c_str(), n_in, index + n_cur - n_in, tok == 0);
    };
    std::vector<float> logits(_cfg.vocab_size);
    // the reference's loop (transformer.cpp:91-101): forward at position i, sample, callback, stop on token 0
    int i = 0, next = -1;
    std::vector<int> cur = input;
    const bool greedy = temperature == 0.0f;
    while (next != 0 && i < max_tokens) {
        if (greedy && (int)cur.size() == 1) {
            // temperature 0: run a chunk of tokens in the device-resident greedy loop (no per-token host round trip)
            int chunk = max_tokens - i; if (chunk > 8) chunk = 8;
            std::vector<int32_t> out(chunk);
            if (flm_decode_greedy(_ctx, cur[0], i, chunk, out.data()) != FLM_OK) { _err = flm_last_error(_ctx); return false; }
            bool stop = false;
            for (int k = 0; k < chunk && !stop; ++k) {
                next = out[k];
                if (!emit(next, i, 1)) stop = true;
                i += 1; cur = {next};
                if (next == 0) stop = true;
            }
            if (stop) break;
            continue;
        }
        int rc;
        if (greedy) { int32_t t; rc = flm_forward_argmax(_ctx, cur.data(), (int)cur.size(), i, &t); next = t; }
        else { rc = flm_forward(_ctx, cur.data(), (int)cur.size(), i, logits.data()); if (rc == FLM_OK) next = _sampler.sample(logits.data(), temperature, topp); }
        if (rc != FLM_OK) { _err = flm_last_error(_ctx); return false; }
        if (!emit(next, i, (int)cur.size())) break;
        i += (int)cur.size();
        cur = {next};
    }
    return true;
}

} // namespace flmhost
