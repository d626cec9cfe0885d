// model_file.h -- readers for the model files the reference CLI accepts: .flm, llama2.c .bin (+ tokenizer.bin).
// From-scratch host code of the MI355X build; formats follow the reference's loaders
// (src/model_loaders/{model_loader,flm_loader,llama2c_loader}.cpp) so that the same files load.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace flmhost {

enum class FileType { UNKNOWN = 0, FLM, GGUF, LLAMA2C };   // == ModelFileType (model_loader.h:28-35)

struct Config {   // the fields of TransformerConfig (model_loader.h:46-68) the hot path uses
    std::string name;
    int dim = 0, hidden_dim = 0, n_layers = 0, n_heads = 0, n_kv_heads = 0, vocab_size = 0, max_seq_len = 0;
    int quant_type = 0;          // 0 none, 1 int16, 2 int8 (QuantType)
    int quant_group_size = 64;
};

struct VocabEntry { std::string index_text, show_text; int type = 1; float score = 0.f; };
struct Vocab {
    std::vector<VocabEntry> tokens;
    int bos = 1, eos = 2, pad = 0;
    std::string conn_tag = "\xE2\x96\x81";   // "▁"
};

struct HostTensor {
    int kind = 0, layer = 0;     // FLM_T_* numbering (flm_loader.cpp:50-67)
    int qtype = 0;               // of `values`
    int rows = 0, cols = 0;
    const void*  values = nullptr;
    const float* scales = nullptr;
    std::vector<char>  owned_values;   // llama2.c path keeps converted data here
    std::vector<float> owned_scales;
};

struct ModelFile {
    Config cfg;
    Vocab vocab;
    std::vector<HostTensor> tensors;
    // backing store (mmap) of zero-copy tensors
    void* map_base = nullptr; size_t map_size = 0;
    ~ModelFile();
    ModelFile() = default;
    ModelFile(const ModelFile&) = delete;
};

FileType detect_file_type(const std::string& path, std::string& err);
// tokenizer_only: stop after the vocabulary (the -e / -d debug modes, main.cpp:246-286)
bool load_model_file(const std::string& ckpt, const std::string& tokenizer_path, FileType ft, bool tokenizer_only, bool debug,
                     ModelFile& out, std::string& err);

// quant::quantize<T> (src/blas/quant_operators.cpp:26-47) on the host, for the llama2.c path that quantizes
// the embedding table at load (llama2c_loader.cpp:83,117-124)
void quantize_groups(const float* x, size_t n, int qtype, void* q, float* scales);

} // namespace flmhost
