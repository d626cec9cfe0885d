// test_shim.cpp -- C entry points over the host-side classes so that tests/ can drive them through ctypes
// (no GPU needed).  Built as lib/libflm_host.so next to the CLI; the CLI itself does not use it.
#include <string.h>

#include <string>

#include "model_file.h"
#include "sampler.h"
#include "tokenizer.h"

using namespace flmhost;

namespace { struct Handle { ModelFile mf; Tokenizer tok; Vocab vocab_copy; std::string err; }; thread_local std::string g_err; }

extern "C" {

const char* fh_last_error() { return g_err.c_str(); }

void* fh_open(const char* ckpt, const char* tknr, int file_type, int tokenizer_only) {
    Handle* h = new Handle();
    if (!load_model_file(ckpt ? ckpt : "", tknr ? tknr : "", (FileType)file_type, tokenizer_only != 0, false, h->mf, g_err)) { delete h; return nullptr; }
    h->tok.set_vocab(h->mf.vocab);
    return h;
}
void fh_close(void* p) { delete (Handle*)p; }

int fh_detect(const char* path) { return (int)detect_file_type(path, g_err); }

// dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, max_seq_len, quant_type, quant_group_size
void fh_config(void* p, int* out9) {
    const Config& c = ((Handle*)p)->mf.cfg;
    int v[9] = {c.dim, c.hidden_dim, c.n_layers, c.n_heads, c.n_kv_heads, c.vocab_size, c.max_seq_len, c.quant_type, c.quant_group_size};
    memcpy(out9, v, sizeof(v));
}
int fh_n_tensors(void* p) { return (int)((Handle*)p)->mf.tensors.size(); }
// kind, layer, qtype, rows, cols
void fh_tensor(void* p, int i, int* out5, const void** values, const float** scales) {
    const HostTensor& t = ((Handle*)p)->mf.tensors[i];
    int v[5] = {t.kind, t.layer, t.qtype, t.rows, t.cols};
    memcpy(out5, v, sizeof(v)); *values = t.values; *scales = t.scales;
}
int fh_vocab_size(void* p) { return ((Handle*)p)->tok.vocab_size(); }
int fh_encode(void* p, const char* text, int add_bos, int* out, int cap) {
    auto v = ((Handle*)p)->tok.encode(text, add_bos != 0);
    int n = (int)v.size() < cap ? (int)v.size() : cap;
    memcpy(out, v.data(), n * sizeof(int));
    return (int)v.size();
}
int fh_decode(void* p, const int* toks, int n, char* out, int cap) {
    std::string s = ((Handle*)p)->tok.decode(std::vector<int>(toks, toks + n));
    int m = (int)s.size() < cap - 1 ? (int)s.size() : cap - 1;
    memcpy(out, s.data(), m); out[m] = 0;
    return (int)s.size();
}
int fh_decode_one(void* p, int tok, int prev, char* out, int cap) {
    std::string s = ((Handle*)p)->tok.decode(tok, prev);
    int m = (int)s.size() < cap - 1 ? (int)s.size() : cap - 1;
    memcpy(out, s.data(), m); out[m] = 0;
    return (int)s.size();
}
// n_draws samples from the same logits (copied per draw) with one sampler state
void fh_sample(int vocab, unsigned long long seed, const float* logits, float temperature, float topp, int n_draws, int* out) {
    Sampler s; s.build(vocab, seed);
    std::vector<float> l(vocab);
    for (int i = 0; i < n_draws; ++i) { memcpy(l.data(), logits, vocab * sizeof(float)); out[i] = s.sample(l.data(), temperature, topp); }
}
void fh_quantize(const float* x, size_t n, int qtype, void* q, float* scales) { quantize_groups(x, n, qtype, q, scales); }

}
