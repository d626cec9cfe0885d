// engine.h -- the MI355X counterpart of ParallelTransformer (src/transformer/transformer.h:76-96): same public
// surface (load / encode / decode / generate / get_quant_type), the per-token forward runs on the GPU through
// the C ABI in include/flm_gpu.h.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "flm_gpu.h"
#include "model_file.h"
#include "sampler.h"
#include "tokenizer.h"

namespace flmhost {

// "title[ a,  b, ...]" with every number right-aligned to the widest one, like the reference's print_vector
void print_vector(const char* title, const std::vector<int>& vec);

class GpuTransformer {
public:
    explicit GpuTransformer(bool debug) : _debug(debug) {}
    ~GpuTransformer();
    // load(ckpt, tokenizer, file type, -q quant type, devices): transformer.cpp:23-42.  More than one device = the reference's parallel width
    // (main.cpp:30,78 `-j`, split_rows transformer.cpp:264-287) on GPUs: ONE sequence, every matmul split by output rows over the devices, one host
    // thread per device (a tensor-parallel rank of the C ABI), activation slices exchanged peer to peer.  A device may be named more than once
    // (ranks sharing a GPU: how the 1-GPU test box exercises the path).
    bool load(const std::string& ckpt, const std::string& tknr, FileType ft, int qtype, const std::vector<int>& devices, uint64_t seed = 0);
    bool load(const std::string& ckpt, const std::string& tknr, FileType ft, int qtype, int device, uint64_t seed = 0) { return load(ckpt, tknr, ft, qtype, std::vector<int>{device}, seed); }
    std::vector<int> encode(const char* prompt) const;
    std::string decode(const std::vector<int>& tokens) const { return _tok.decode(tokens); }
    // generate(prompt, cb(text, n_input, n_output, ended), max_new_tokens, temperature, topp): transformer.cpp:54-103
    bool generate(const char* prompt, const std::function<bool(const char*, int, int, bool)>& cb, int max_new_tokens, float temperature, float topp);
    int get_quant_type() const { return _cfg.quant_type; }
    const std::string& error() const { return _err; }
    // more than one device: which launch structure of the sharded token load() settled on (calibrate_structure), for --detail
    const std::string& tp_structure() const { return _tp_structure; }
private:
    // Sharded over several devices: time ONE token under the conservative launch structure (exchange flag rounds as launches of their own between distinct devices), then
    // under each faster one ("tp_trust_fused" 1; + "tp_fuse_ffn" 1), and keep the fastest whose logits are the conservative structure's bit for bit -- nothing is trusted
    // that was not verified on THIS machine; a structure that gives up or differs is dropped and the group put back.
    bool connect_ranks();
    bool calibrate_structure();
    std::string _tp_structure;
    bool _debug;
    Config _cfg;
    Tokenizer _tok;
    Sampler _sampler;
    std::vector<flm_ctx*> _ctxs;                    // rank r's context (rank 0's results are the ones reported)
    // run f(rank) on every rank at once (the ranks wait for each other's slices inside the launches); first non-zero status wins
    int on_all(const std::function<int(int)>& f);
    std::string _err;
};

} // namespace flmhost
