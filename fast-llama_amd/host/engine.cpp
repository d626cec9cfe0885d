#include "engine.h"
#include <algorithm>

#include <stdio.h>
#include <string.h>

#include <chrono>

#include <iomanip>
#include <iostream>
#include <thread>

namespace flmhost {

GpuTransformer::~GpuTransformer() { for (flm_ctx* c : _ctxs) if (c) flm_ctx_destroy(c); }

int GpuTransformer::on_all(const std::function<int(int)>& f) {
    const int n = (int)_ctxs.size();
    if (n == 1) return f(0);
    std::vector<int> rc(n, 0);
    std::vector<std::thread> th;
    for (int r = 1; r < n; ++r) th.emplace_back([&, r] { rc[r] = f(r); });
    rc[0] = f(0);
    for (auto& t : th) t.join();
    for (int r = 0; r < n; ++r) if (rc[r] != FLM_OK) { _err = std::string("rank ") + std::to_string(r) + ": " + flm_last_error(_ctxs[r]); return rc[r]; }
    return FLM_OK;
}

bool GpuTransformer::load(const std::string& ckpt, const std::string& tknr, FileType ft, int qtype, const std::vector<int>& devices, uint64_t seed) {
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
    const int world = (int)devices.size();
    if (world < 1 || world > 8) { _err = "1 to 8 devices"; return false; }
    _ctxs.assign(world, nullptr);
    for (int r = 0; r < world; ++r) {
        const int rc = flm_ctx_create(&d, devices[r], r, world, nullptr, &_ctxs[r]);
        if (rc != FLM_OK) { _err = std::string("GPU context (rank ") + std::to_string(r) + ", device " + std::to_string(devices[r]) + "): " + flm_last_error(nullptr); return false; }
    }
    for (const HostTensor& t : mf.tensors) {
        // a quantized tensor of another type than the model's cannot be multiplied (tensor.cpp:557-561); every rank is handed the full tensor and keeps its rows
        for (int r = 0; r < world; ++r) {
            const int rc = flm_upload_tensor(_ctxs[r], t.kind, t.layer, t.qtype, t.values, t.scales, t.rows, t.cols);
            if (rc != FLM_OK) { _err = std::string("upload: ") + flm_last_error(_ctxs[r]); return false; }
        }
    }
    if (world > 1) {
        if (!connect_ranks()) return false;
        if (!calibrate_structure()) return false;
    }
    return true;
}

// connect the ranks peer to peer: every rank's exchange buffer mapped into every other (ranks of one process share pointers); also the round at which the group
// agrees on its launch structure from every rank's options
bool GpuTransformer::connect_ranks() {
    const int world = (int)_ctxs.size();
    std::vector<unsigned char> blobs((size_t)world * FLM_P2P_BLOB_BYTES);
    for (int r = 0; r < world; ++r)
        if (flm_p2p_export(_ctxs[r], blobs.data() + (size_t)r * FLM_P2P_BLOB_BYTES) != FLM_OK) { _err = std::string("p2p export: ") + flm_last_error(_ctxs[r]); return false; }
    for (int r = 0; r < world; ++r)
        if (flm_p2p_import(_ctxs[r], blobs.data(), world) != FLM_OK) { _err = std::string("p2p import: ") + flm_last_error(_ctxs[r]); return false; }
    return true;
}

// The group's launch structure.  Default: what the LIBRARY takes by itself -- between distinct devices the conservative structure (a flag round behind a kernel boundary, release /
// acquire fences), on one device the folded / rank-spanning launches.  The faster structures between devices rely on system-scope store / flag ordering over xGMI that the build box could
// only rehearse between CU partitions of one GPU, and a memory-ordering race is rare: one matching token verifies nothing.  They are therefore OPT-IN (FLM_TP_CALIBRATE=1 in the
// environment): every candidate must then reproduce the conservative structure's ids over a 96-token greedy decode AND its last logits bit for bit on every rank, is timed on the
// device (flm_decode_timed: HIP events, median of three runs of 32 tokens) and replaces the incumbent only if at least 2 % faster.  A candidate that fails in any way other than a
// mismatch (a time-out raises the group's abort line on every peer: the contexts cannot be used any more) is fatal -- load() fails instead of returning a poisoned group.
bool GpuTransformer::calibrate_structure() {
    const char* env = getenv("FLM_TP_CALIBRATE");
    if (!env || !env[0] || env[0] == '0') { _tp_structure = "library default"; return true; }
    struct Cand { const char* name; int trust, ffn, layers, fence, gr; };
    const Cand cands[] = {{"exchange launches", 0, 0, 0, -1, 0}, {"folded exchanges, QKV + attention + Wo across ranks", 1, 0, 0, -1, 0},
                          {"all layers in one rank-spanning launch, fenced flags", 1, 0, 1, 3, 0}, {"all layers in one rank-spanning launch, flags", 1, 0, 1, 0, 0},
                          {"all layers in one rank-spanning launch, data-tagged granules (no flag, no fence)", 1, 0, 1, -1, 1},
                          {"folded exchanges, + FFN13 + FFN2 across ranks", 1, 1, 0, -1, 0}};
    const int world = (int)_ctxs.size(), V = _cfg.vocab_size, kVerify = 96 < _cfg.max_seq_len - 2 ? 96 : _cfg.max_seq_len - 2, kTime = kVerify < 32 ? kVerify : 32;
    auto apply = [&](const Cand& c) {
        for (int r = 0; r < world; ++r)
            if (flm_set_option(_ctxs[r], "tp_trust_fused", c.trust) != FLM_OK || flm_set_option(_ctxs[r], "tp_fuse_ffn", c.ffn) != FLM_OK ||
                flm_set_option(_ctxs[r], "tp_fuse_layers", c.layers) != FLM_OK || flm_set_option(_ctxs[r], "tp_fence", c.fence) != FLM_OK || flm_set_option(_ctxs[r], "gr_edges", c.gr) != FLM_OK) { _err = std::string("set_option: ") + flm_last_error(_ctxs[r]); return false; }
        return connect_ranks();
    };
    // BOS at position 0, then kVerify greedy tokens: every rank's ids and its last logits are the verdict; then the device time of kTime tokens (median of three)
    struct Probe { std::vector<int32_t> ids; std::vector<float> logits; double us = 0.0; };
    auto probe = [&](Probe& p) -> int {     // 0 ok, 1 the ranks disagree among themselves, -1 an error (fatal for the group)
        std::vector<std::vector<int32_t>> ids(world, std::vector<int32_t>(kVerify));
        std::vector<std::vector<float>> lg(world, std::vector<float>(V));
        const int32_t tok = 1;
        if (on_all([&](int r) { return flm_reset_kv(_ctxs[r]); }) != FLM_OK) return -1;
        if (on_all([&](int r) { int rc = flm_decode_greedy(_ctxs[r], tok, 0, kVerify, ids[r].data()); if (rc == FLM_OK) rc = flm_forward(_ctxs[r], &ids[r][kVerify - 1], 1, kVerify, lg[r].data()); return rc; }) != FLM_OK) return -1;
        for (int r = 0; r < world; ++r) { int fb = 0; flm_query(_ctxs[r], "fallback", &fb); if (fb) { _err = "a cross-workgroup wait timed out"; return -1; } }
        for (int r = 1; r < world; ++r) if (ids[r] != ids[0] || memcmp(lg[r].data(), lg[0].data(), (size_t)V * 4) != 0) return 1;
        std::vector<double> t;
        for (int rep = 0; rep < 3; ++rep) {
            std::vector<float> ms(world, 0.f);
            if (on_all([&](int r) { return flm_decode_timed(_ctxs[r], tok, 0, kTime, &ms[r]); }) != FLM_OK) return -1;
            double m = 0; for (float x : ms) if (x > m) m = x;
            t.push_back(m * 1000.0 / kTime);
        }
        std::sort(t.begin(), t.end());
        p.ids = ids[0]; p.logits = lg[0]; p.us = t[1];
        return 0;
    };
    Probe want, got;
    int best = 0;
    if (!apply(cands[0])) return false;
    if (probe(want) != 0) { if (_err.empty()) _err = "tensor parallel: the conservative structure does not give the same results on every rank"; return false; }
    double best_us = want.us;
    for (int i = 1; i < (int)(sizeof cands / sizeof cands[0]); ++i) {
        if (!apply(cands[i])) return false;
        const int rc = probe(got);
        if (rc < 0) { if (_err.empty()) _err = std::string("tensor parallel: structure \"") + cands[i].name + "\" failed: " + flm_last_error(_ctxs[0]); return false; }   // (the group may be poisoned: no way back)
        if (rc > 0 || got.ids != want.ids || memcmp(got.logits.data(), want.logits.data(), (size_t)V * 4) != 0) {
            if (_debug) fprintf(stderr, "tensor parallel: structure \"%s\" dropped (its ids / logits differ from the conservative structure's)\n", cands[i].name);
            continue;
        }
        if (_debug) fprintf(stderr, "tensor parallel: structure \"%s\": %.1f us per token against %.1f\n", cands[i].name, got.us, best_us);
        if (got.us < 0.98 * best_us) { best_us = got.us; best = i; }
    }
    if (!apply(cands[best])) return false;
    _tp_structure = cands[best].name;
    return on_all([&](int r) { return flm_reset_kv(_ctxs[r]); }) == FLM_OK;
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
            std::vector<std::vector<int32_t>> outs(_ctxs.size(), std::vector<int32_t>(chunk));
            if (on_all([&](int r) { return flm_decode_greedy(_ctxs[r], cur[0], i, chunk, outs[r].data()); }) != FLM_OK) return false;
            out = outs[0];
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
        if (greedy) {
            std::vector<int32_t> ts(_ctxs.size(), 0);
            rc = on_all([&](int r) { return flm_forward_argmax(_ctxs[r], cur.data(), (int)cur.size(), i, &ts[r]); }); next = ts[0];
        } else {
            std::vector<std::vector<float>> lgs(_ctxs.size());
            for (size_t r = 1; r < lgs.size(); ++r) lgs[r].resize(_cfg.vocab_size);
            rc = on_all([&](int r) { return flm_forward(_ctxs[r], cur.data(), (int)cur.size(), i, r == 0 ? logits.data() : lgs[r].data()); });
            if (rc == FLM_OK) next = _sampler.sample(logits.data(), temperature, topp);
        }
        if (rc != FLM_OK) return false;
        if (!emit(next, i, (int)cur.size())) break;
        i += (int)cur.size();
        cur = {next};
    }
    return true;
}

} // namespace flmhost
