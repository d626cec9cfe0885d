#include "model_file.h"

#include <fcntl.h>
#include <math.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>

namespace flmhost {

namespace {
constexpr uint32_t kFlmTag = 0xFA571AEAu;       // flm_loader.cpp:114
enum { BT_BASE_ITEM = 0, BT_DICT = 1, BT_TENSOR = 2, BT_ARRAY = 3, BT_STRING = 4 };
enum { DT_INT8 = 1, DT_INT16 = 2, DT_INT32 = 3, DT_INT64 = 4, DT_FLOAT32 = 11, DT_FLOAT64 = 12 };

struct Cursor {
    const uint8_t* base; size_t size;
    bool ok(size_t pos, size_t n) const { return pos <= size && n <= size - pos; }
    template <class T> T at(size_t pos) const { T v; memcpy(&v, base + pos, sizeof(T)); return v; }
};

// one block header, both encodings (flm_loader.cpp:132-178): BASE_ITEM packs the value and the name in the
// header; every other block is {type, dtype, header_size, header_data_size, name_offset, name_size,
// tail_pad u16, data_size u64, header_data, name, pad}
struct Block {
    int type = 0, dtype = 0; size_t header_size = 0, data_size = 0, total = 0, pos = 0, hds = 0;
    std::string name;
};
bool parse_block(const Cursor& c, size_t pos, Block& b) {
    if (!c.ok(pos, 8)) return false;
    b.pos = pos; b.type = c.at<uint8_t>(pos); b.dtype = c.at<uint8_t>(pos + 1); b.header_size = c.at<uint8_t>(pos + 2); b.hds = c.at<uint8_t>(pos + 3);
    if (b.header_size < 8 || !c.ok(pos, b.header_size)) return false;
    if (b.type == BT_BASE_ITEM) {
        const size_t noff = b.hds <= 4 ? 8 : 16;
        b.name.assign(reinterpret_cast<const char*>(c.base + pos + noff), strnlen(reinterpret_cast<const char*>(c.base + pos + noff), b.header_size - noff));
        b.data_size = b.hds; b.total = b.header_size;
        return true;
    }
    if (b.header_size < 16) return false;
    const size_t name_off = c.at<uint8_t>(pos + 4), name_size = c.at<uint8_t>(pos + 5), tail = c.at<uint16_t>(pos + 6);
    b.data_size = c.at<uint64_t>(pos + 8);
    if (name_off + name_size > b.header_size) return false;
    b.name.assign(reinterpret_cast<const char*>(c.base + pos + name_off), name_size);
    b.total = b.header_size + b.data_size + tail;
    return c.ok(pos, b.total);
}
double item_number(const Cursor& c, const Block& b) {
    const bool small = b.hds <= 4;
    const size_t p = b.pos + (small ? 4 : 8);
    switch (b.dtype) {
    case DT_FLOAT32: return c.at<float>(p);
    case DT_FLOAT64: return c.at<double>(p);
    case DT_INT8: return c.at<int8_t>(p);
    case DT_INT16: return c.at<int16_t>(p);
    case DT_INT64: return (double)c.at<int64_t>(p);
    default: return small ? (double)c.at<int32_t>(p) : (double)c.at<int64_t>(p);
    }
}

bool load_flm(const std::string& path, bool tokenizer_only, bool debug, ModelFile& m, std::string& err) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { err = "Failed to open model file:" + path; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) { close(fd); err = "Cannot stat model file (or it is too short):" + path; return false; }
    m.map_size = (size_t)st.st_size;
    m.map_base = mmap(nullptr, m.map_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m.map_base == MAP_FAILED) { m.map_base = nullptr; err = "mmap failed:" + path; return false; }
    Cursor c{reinterpret_cast<const uint8_t*>(m.map_base), m.map_size};
    if (!c.ok(0, 8) || c.at<uint32_t>(0) != kFlmTag) { err = "Not a valid model file. Invalid file tag"; return false; }
    if (debug) fprintf(stderr, "FLM version:%d.%d.%d\n", c.at<uint8_t>(4), c.at<uint8_t>(5), c.at<uint16_t>(6));
    Config& cf = m.cfg;
    for (size_t pos = 8; pos < c.size;) {
        Block b;
        if (!parse_block(c, pos, b)) { err = "Reading block header error at file_pos:" + std::to_string(pos); return false; }
        if (b.name == "model_config") {
            // load_config (flm_loader.cpp:390-442)
            for (size_t q = pos + b.header_size, end = q + b.data_size; q < end;) {
                Block it;
                if (!parse_block(c, q, it)) { err = "Reading config block header error at file_pos:" + std::to_string(q); return false; }
                if (it.type == BT_BASE_ITEM) {
                    const double v = item_number(c, it);
                    if (it.name == "vocab_size") cf.vocab_size = (int)v; else if (it.name == "dim") cf.dim = (int)v;
                    else if (it.name == "hidden_dim") cf.hidden_dim = (int)v; else if (it.name == "n_heads") cf.n_heads = (int)v;
                    else if (it.name == "n_kv_heads") cf.n_kv_heads = (int)v; else if (it.name == "n_layers") cf.n_layers = (int)v;
                    else if (it.name == "max_length") cf.max_seq_len = (int)v; else if (it.name == "quant_type") cf.quant_type = (int)v;
                    else if (it.name == "quant_group_size") cf.quant_group_size = (int)v;
                } else if (it.type == BT_STRING && it.name == "name") {
                    cf.name.assign(reinterpret_cast<const char*>(c.base + q + it.header_size), strnlen(reinterpret_cast<const char*>(c.base + q + it.header_size), it.data_size));
                }
                q += it.total;
            }
            if (cf.n_kv_heads < 1) cf.n_kv_heads = cf.n_heads;
            if (cf.n_heads < 1 || cf.dim % cf.n_heads || cf.n_kv_heads > cf.n_heads) { err = "Invalid config value"; return false; }
        } else if (b.name == "tokenizer") {
            // load_tokenizer (flm_loader.cpp:444-491): {vocab_type, conn_tag_pos, special[8], vocab_size, text_size, items, text}
            size_t q = pos + b.header_size;
            if (!c.ok(q, 48)) { err = "Reading tokenizer header error"; return false; }
            const uint32_t conn_pos = c.at<uint32_t>(q + 4);
            int32_t special[8]; memcpy(special, c.base + q + 8, 32);
            const uint32_t n = c.at<uint32_t>(q + 40), tsz = c.at<uint32_t>(q + 44);
            const size_t items = q + 48, text = items + (size_t)16 * n;
            if (!c.ok(text, tsz)) { err = "Reading tokenizer texts error"; return false; }
            // every offset comes from the file: one past the text area is an error, not a read beyond the mapping
            bool bad_off = false;
            auto str_at = [&](uint32_t off) {
                if (off >= tsz) { bad_off = true; return std::string(); }
                const char* p = reinterpret_cast<const char*>(c.base + text + off); return std::string(p, strnlen(p, tsz - off));
            };
            m.vocab.tokens.resize(n);
            for (uint32_t i = 0; i < n; ++i) {
                const size_t it = items + (size_t)16 * i;
                auto& t = m.vocab.tokens[i];
                t.index_text = str_at(c.at<uint32_t>(it)); t.show_text = str_at(c.at<uint32_t>(it + 4));
                t.type = (int)c.at<uint32_t>(it + 8); t.score = c.at<float>(it + 12);
            }
            m.vocab.conn_tag = str_at(conn_pos);
            if (bad_off) { err = "Tokenizer text offset outside the text area"; return false; }
            m.vocab.bos = special[1]; m.vocab.eos = special[2]; m.vocab.pad = special[3];
            if (tokenizer_only) return true;
        } else if (b.type == BT_TENSOR) {
            // tensor header (flm_loader.cpp:165-177): shape[4] u32, tensor_type u16, layer_id u16, scales_size u32
            const size_t hp = pos + 16;
            if (b.header_size < 40 || !c.ok(hp, 24)) { err = "Tensor block header too short:" + b.name; return false; }
            uint32_t shape[4]; memcpy(shape, c.base + hp, 16);
            for (int i = 0; i < 4; ++i) if (shape[i] > 0x7fffffffu) { err = "Tensor dimension out of range in block:" + b.name; return false; }
            HostTensor t;
            t.kind = c.at<uint16_t>(hp + 16); t.layer = c.at<uint16_t>(hp + 18);
            const uint32_t ssz = c.at<uint32_t>(hp + 20);
            int nd = 0; while (nd < 4 && shape[nd] > 0) ++nd;
            if (nd == 1) { t.rows = 1; t.cols = (int)shape[0]; } else if (nd == 2) { t.rows = (int)shape[0]; t.cols = (int)shape[1]; }
            else { err = "Unsupported tensor rank in block:" + b.name; return false; }
            t.qtype = b.dtype == DT_INT8 ? 2 : b.dtype == DT_INT16 ? 1 : 0;
            if (b.dtype != DT_INT8 && b.dtype != DT_INT16 && b.dtype != DT_FLOAT32) { err = "Unsupported tensor data type in block:" + b.name; return false; }
            const size_t esz = t.qtype == 2 ? 1 : t.qtype == 1 ? 2 : 4;
            const unsigned long long elems = (unsigned long long)t.rows * (unsigned long long)t.cols;       // < 2^62
            if (elems > (1ull << 40)) { err = "Tensor too large in block:" + b.name; return false; }
            const size_t vbytes = (size_t)elems * esz;
            const size_t d0 = pos + b.header_size;
            if (vbytes > b.data_size || (size_t)ssz * 4 > b.data_size - vbytes) { err = "Tensor block too small:" + b.name; return false; }
            if (t.qtype && (t.cols % 64 || (size_t)ssz < elems / 64)) { err = "Quantized tensor with too few scales:" + b.name; return false; }
            t.values = c.base + d0;
            t.scales = ssz ? reinterpret_cast<const float*>(c.base + d0 + vbytes) : nullptr;
            if (t.qtype && !ssz) { err = "Quantized tensor without scales:" + b.name; return false; }
            if (debug) fprintf(stderr, "Loading tensor:%-50s type:%d layer_id:%d shape:(%d,%d)\n", b.name.c_str(), t.kind, t.layer, t.rows, t.cols);
            m.tensors.push_back(std::move(t));
        }
        pos += b.total;
    }
    return true;
}

// llama2.c legacy checkpoint (llama2c_loader.cpp:21-29,126-194) + its tokenizer.bin (tokenizer.cpp:173-245)
bool load_llama2c_tokenizer(const std::string& path, int vocab_size, Vocab& v, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "Failed to open tokenizer file " + path; return false; }
    int max_len = 0;
    if (fread(&max_len, 4, 1, f) != 1) { fclose(f); err = "Reading tokenizer file error:" + path; return false; }
    v.tokens.resize(vocab_size);
    for (int i = 0; i < vocab_size; ++i) {
        float score; int len;
        if (fread(&score, 4, 1, f) != 1 || fread(&len, 4, 1, f) != 1 || len < 0 || len > 4096) { fclose(f); err = "Reading tokenizer file error:" + path; return false; }
        std::string s(len, '\0');
        if (len && fread(&s[0], 1, len, f) != (size_t)len) { fclose(f); err = "Reading tokenizer file error:" + path; return false; }
        v.tokens[i].index_text = s; v.tokens[i].show_text = s; v.tokens[i].score = score;
    }
    fclose(f);
    v.bos = 1; v.eos = 2; v.pad = 0;
    v.conn_tag.clear();      // Tokenizer::load never sets _conn_tag (tokenizer.cpp:162-233): " " has no token of its own and falls back to the byte token
    return true;
}

bool load_llama2c(const std::string& ckpt, const std::string& tok_path, bool tokenizer_only, ModelFile& m, std::string& err) {
    FILE* f = fopen(ckpt.c_str(), "rb");
    if (!f) { err = "Failed to open model file:" + ckpt; return false; }
    int h[7];
    if (fread(h, 4, 7, f) != 7) { fclose(f); err = "Reading model file error:" + ckpt; return false; }
    Config& c = m.cfg;
    c.dim = h[0]; c.hidden_dim = h[1]; c.n_layers = h[2]; c.n_heads = h[3]; c.n_kv_heads = h[4]; c.vocab_size = abs(h[5]); c.max_seq_len = h[6];
    // the header is untrusted input (and -f llama2c skips the detector): refuse what the quantizer and the engine cannot handle
    if (c.dim <= 0 || c.hidden_dim <= 0 || c.n_layers <= 0 || c.n_heads <= 0 || c.n_kv_heads <= 0 || c.vocab_size <= 0 || c.max_seq_len <= 0 ||
        c.dim % 64 || c.hidden_dim % 64 || c.dim % c.n_heads || c.n_kv_heads > c.n_heads || c.n_heads % c.n_kv_heads ||
        c.dim > (1 << 20) || c.hidden_dim > (1 << 22) || c.n_layers > 4096 || c.vocab_size > (1 << 24)) {
        fclose(f); err = "Invalid llama2.c header (dim / hidden_dim must be positive multiples of 64, dim a multiple of n_heads, 0 < n_kv_heads <= n_heads):" + ckpt; return false;
    }
    c.quant_type = 2; c.name = "llama2c";   // weights are quantized to INT8 below; the reference leaves NONE and only works with -q int8
    const bool shared = h[5] > 0;
    if (!load_llama2c_tokenizer(tok_path, c.vocab_size, m.vocab, err)) { fclose(f); return false; }
    if (tokenizer_only) { fclose(f); return true; }
    const int hs = c.dim / c.n_heads, kvd = hs * c.n_kv_heads, L = c.n_layers;
    std::vector<float> buf;
    auto read_f32 = [&](size_t n) { buf.resize(n); return fread(buf.data(), 4, n, f) == n; };
    // every 2-D tensor (the embedding table included) is quantized to INT8 / 64 at load (llama2c_loader.cpp:83,117-124)
    auto add_q = [&](int kind, int layer, int rows, int cols, const float* src) {
        HostTensor t; t.kind = kind; t.layer = layer; t.qtype = 2; t.rows = rows; t.cols = cols;
        t.owned_values.resize((size_t)rows * cols); t.owned_scales.resize(((size_t)rows * cols + 63) / 64);
        quantize_groups(src, (size_t)rows * cols, 2, t.owned_values.data(), t.owned_scales.data());
        m.tensors.push_back(std::move(t));
    };
    auto add_f = [&](int kind, int layer, int cols, const float* src) {
        HostTensor t; t.kind = kind; t.layer = layer; t.qtype = 0; t.rows = 1; t.cols = cols;
        t.owned_values.resize((size_t)cols * 4); memcpy(t.owned_values.data(), src, (size_t)cols * 4);
        m.tensors.push_back(std::move(t));
    };
    auto per_layer_q = [&](int kind, int rows, int cols) {
        if (!read_f32((size_t)L * rows * cols)) return false;
        for (int l = 0; l < L; ++l) add_q(kind, l, rows, cols, buf.data() + (size_t)l * rows * cols);
        return true;
    };
    auto per_layer_f = [&](int kind) {
        if (!read_f32((size_t)L * c.dim)) return false;
        for (int l = 0; l < L; ++l) add_f(kind, l, c.dim, buf.data() + (size_t)l * c.dim);
        return true;
    };
    bool ok = read_f32((size_t)c.vocab_size * c.dim);
    if (ok) { add_q(1, 0, c.vocab_size, c.dim, buf.data()); if (shared) add_q(3, 0, c.vocab_size, c.dim, buf.data()); }
    ok = ok && per_layer_f(17) && per_layer_q(18, c.dim, c.dim) && per_layer_q(19, kvd, c.dim) && per_layer_q(20, kvd, c.dim) &&
         per_layer_q(21, c.dim, c.dim) && per_layer_f(25) && per_layer_q(22, c.hidden_dim, c.dim) && per_layer_q(24, c.dim, c.hidden_dim) &&
         per_layer_q(23, c.hidden_dim, c.dim);
    if (ok && (ok = read_f32(c.dim))) add_f(2, 0, c.dim, buf.data());
    if (ok) ok = fseek(f, (long)((size_t)hs * c.max_seq_len / 2 * 2 * 4), SEEK_CUR) == 0;      // freq_cis real+imag: unused (rope_v2 recomputes)
    if (ok && !shared) { ok = read_f32((size_t)c.vocab_size * c.dim); if (ok) add_q(3, 0, c.vocab_size, c.dim, buf.data()); }
    fclose(f);
    if (!ok) { err = "Failed to read weights while loading:" + ckpt; return false; }
    for (auto& t : m.tensors) { t.values = t.owned_values.data(); t.scales = t.owned_scales.empty() ? nullptr : t.owned_scales.data(); }
    return true;
}

// ---------------------------------------------------------------------------------------------
// gguf (v2 / v3), LLaMA architecture, F32 and F16 tensors.  Same meaning as the reference's reader
// (src/model_loaders/gguf_loader.cpp:205-489): the hyper-parameters come from the llama.* keys, the tokenizer from
// tokenizer.ggml.{tokens,scores,token_type,bos/eos/padding_token_id}, tensors are named the llama.cpp way and already
// carry the interleaved q/k row order; fp32 weights are handed to the device as fp32 masters and quantized there with
// the -q type.  Unlike the reference this reader skips keys it does not know instead of refusing the file, and it
// converts F16 tensors (the reference reads them and forgets to convert, :466); Q8_0 (32-element groups, which the
// engine's 64-element operators cannot run) is refused.
// ---------------------------------------------------------------------------------------------
enum { GV_UINT8 = 0, GV_INT8, GV_UINT16, GV_INT16, GV_UINT32, GV_INT32, GV_FLOAT32, GV_BOOL, GV_STRING, GV_ARRAY, GV_UINT64, GV_INT64, GV_FLOAT64 };
int gv_size(int t) { static const int sz[] = {1, 1, 2, 2, 4, 4, 4, 1, -1, -1, 8, 8, 8}; return (t >= 0 && t <= GV_FLOAT64) ? sz[t] : 0; }

struct GgufReader {
    Cursor c; size_t pos = 0; bool bad = false;
    template <class T> T get() { if (!c.ok(pos, sizeof(T))) { bad = true; return T(); } T v = c.at<T>(pos); pos += sizeof(T); return v; }
    std::string str() {
        const uint64_t n = get<uint64_t>();
        if (bad || !c.ok(pos, n)) { bad = true; return std::string(); }
        std::string s(reinterpret_cast<const char*>(c.base + pos), (size_t)n); pos += n; return s;
    }
    double number(int t) {
        switch (t) {
        case GV_UINT8: case GV_BOOL: return get<uint8_t>(); case GV_INT8: return get<int8_t>(); case GV_UINT16: return get<uint16_t>();
        case GV_INT16: return get<int16_t>(); case GV_UINT32: return get<uint32_t>(); case GV_INT32: return get<int32_t>();
        case GV_FLOAT32: return get<float>(); case GV_UINT64: return (double)get<uint64_t>(); case GV_INT64: return (double)get<int64_t>();
        case GV_FLOAT64: return get<double>(); default: bad = true; return 0;
        }
    }
    void skip(int t) {
        if (t == GV_STRING) { str(); return; }
        if (t == GV_ARRAY) { const int et = get<int32_t>(); const uint64_t n = get<uint64_t>(); for (uint64_t i = 0; i < n && !bad; ++i) skip(et); return; }
        const int sz = gv_size(t);
        if (sz <= 0 || !c.ok(pos, (size_t)sz)) { bad = true; return; }
        pos += (size_t)sz;
    }
};

float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { int e = -1; uint32_t m = man; do { ++e; m <<= 1; } while (!(m & 0x400u)); bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13); }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

bool load_gguf(const std::string& path, bool tokenizer_only, bool debug, ModelFile& m, std::string& err) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { err = "Failed to open gguf file:" + path; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) { close(fd); err = "Cannot stat model file (or it is too short):" + path; return false; }
    m.map_size = (size_t)st.st_size;
    m.map_base = mmap(nullptr, m.map_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m.map_base == MAP_FAILED) { m.map_base = nullptr; err = "mmap failed:" + path; return false; }
    GgufReader r{Cursor{reinterpret_cast<const uint8_t*>(m.map_base), m.map_size}};
    const uint32_t magic = r.get<uint32_t>(), version = r.get<uint32_t>();
    const int64_t n_tensors = r.get<int64_t>(), n_kv = r.get<int64_t>();
    if (r.bad || magic != 0x46554747u || n_tensors < 1 || n_kv < 1) { err = "Not a valid gguf file:" + path; return false; }
    if (debug) fprintf(stderr, "gguf version:%u tensors:%ld keys:%ld\n", version, (long)n_tensors, (long)n_kv);
    Config& cf = m.cfg; Vocab& vc = m.vocab;
    cf.quant_type = 0; cf.quant_group_size = 64;
    size_t alignment = 32; int rope_dims = 0;
    std::vector<std::string> texts; std::vector<float> scores; std::vector<int> types;
    for (int64_t i = 0; i < n_kv && !r.bad; ++i) {
        const std::string key = r.str();
        const int t = r.get<int32_t>();
        const bool num = t != GV_STRING && t != GV_ARRAY;
        if (key == "general.architecture" && t == GV_STRING) { const std::string a = r.str(); if (a != "llama") { err = "Unsupported architecture:" + a; return false; } }
        else if (key == "general.name" && t == GV_STRING) cf.name = r.str();
        else if (key == "general.file_type" && num) { const int ft = (int)r.number(t); if (ft != 0 && ft != 1) { err = "Unsupported gguf file type " + std::to_string(ft) + " (F32 and F16 files load; Q8_0 uses 32-element groups the engine cannot run)"; return false; } }
        else if (key == "general.alignment" && num) alignment = (size_t)r.number(t);
        else if (key == "llama.context_length" && num) cf.max_seq_len = (int)r.number(t);
        else if (key == "llama.embedding_length" && num) cf.dim = (int)r.number(t);
        else if (key == "llama.block_count" && num) cf.n_layers = (int)r.number(t);
        else if (key == "llama.feed_forward_length" && num) cf.hidden_dim = (int)r.number(t);
        else if (key == "llama.attention.head_count" && num) cf.n_heads = (int)r.number(t);
        else if (key == "llama.attention.head_count_kv" && num) cf.n_kv_heads = (int)r.number(t);
        else if (key == "llama.rope.dimension_count" && num) rope_dims = (int)r.number(t);
        else if (key == "tokenizer.ggml.bos_token_id" && num) vc.bos = (int)r.number(t);
        else if (key == "tokenizer.ggml.eos_token_id" && num) vc.eos = (int)r.number(t);
        else if (key == "tokenizer.ggml.padding_token_id" && num) vc.pad = (int)r.number(t);
        else if (key == "tokenizer.ggml.tokens" && t == GV_ARRAY) {
            const int et = r.get<int32_t>(); const uint64_t n = r.get<uint64_t>();
            if (et != GV_STRING || n > (1ull << 24)) { err = "Reading tokens error"; return false; }
            texts.resize((size_t)n); for (auto& s : texts) s = r.str();
        } else if (key == "tokenizer.ggml.scores" && t == GV_ARRAY) {
            const int et = r.get<int32_t>(); const uint64_t n = r.get<uint64_t>();
            if (n > (1ull << 24)) { err = "Reading scores error"; return false; }
            scores.resize((size_t)n); for (auto& v : scores) v = (float)r.number(et);
        } else if (key == "tokenizer.ggml.token_type" && t == GV_ARRAY) {
            const int et = r.get<int32_t>(); const uint64_t n = r.get<uint64_t>();
            if (n > (1ull << 24)) { err = "Reading token types error"; return false; }
            types.resize((size_t)n); for (auto& v : types) v = (int)r.number(et);
        } else r.skip(t);                                   // rope.freq_base, rms epsilon (the operators hard-code both), merges, ...
    }
    if (r.bad) { err = "Reading gguf key/value section error:" + path; return false; }
    const int hs = cf.n_heads > 0 ? cf.dim / cf.n_heads : 0;
    if (cf.n_kv_heads < 1) cf.n_kv_heads = cf.n_heads;
    if (cf.n_heads < 1 || cf.dim % cf.n_heads != 0 || (rope_dims > 0 && rope_dims != hs)) { err = "Invalid dim / n_heads / rope dimension count in gguf file"; return false; }
    if (texts.empty()) { err = "gguf file without tokenizer.ggml.tokens"; return false; }
    cf.vocab_size = (int)texts.size();
    vc.tokens.resize(texts.size());
    for (size_t i = 0; i < texts.size(); ++i) {            // Tokenizer::set_token_texts (tokenizer.cpp:73-120): "▁xyz" is shown as " xyz"
        VocabEntry& e = vc.tokens[i];
        e.index_text = texts[i];
        e.show_text = texts[i].compare(0, vc.conn_tag.size(), vc.conn_tag) == 0 ? " " + texts[i].substr(vc.conn_tag.size()) : texts[i];
        e.score = i < scores.size() ? scores[i] : 0.f;
        e.type = i < types.size() ? types[i] : 1;
    }
    if (tokenizer_only) return true;

    struct Info { std::string name; int kind = 0, layer = 0, dtype = 0; int64_t dims[4] = {1, 1, 1, 1}; int nd = 0; uint64_t off = 0; };
    std::vector<Info> infos((size_t)n_tensors);
    for (auto& ti : infos) {
        ti.name = r.str();
        ti.nd = (int)r.get<uint32_t>();
        if (r.bad || ti.nd < 1 || ti.nd > 4) { err = "Invalid shape of tensor:" + ti.name; return false; }
        for (int j = 0; j < ti.nd; ++j) ti.dims[j] = r.get<int64_t>();
        ti.dtype = r.get<int32_t>(); ti.off = r.get<uint64_t>();
        // names -> .flm tensor kinds
        static const struct { const char* n; int kind; } top[] = {{"token_embd.weight", 1}, {"output_norm.weight", 2}, {"output.weight", 3}};
        static const struct { const char* n; int kind; } per[] = {{"attn_norm.weight", 17}, {"attn_q.weight", 18}, {"attn_k.weight", 19}, {"attn_v.weight", 20},
            {"attn_output.weight", 21}, {"ffn_gate.weight", 22}, {"ffn_up.weight", 23}, {"ffn_down.weight", 24}, {"ffn_norm.weight", 25}};
        for (auto& e : top) if (ti.name == e.n) ti.kind = e.kind;
        if (!ti.kind && ti.name.compare(0, 4, "blk.") == 0) {
            const size_t dot = ti.name.find('.', 4);
            if (dot != std::string::npos) {
                ti.layer = atoi(ti.name.c_str() + 4);
                for (auto& e : per) if (ti.name.compare(dot + 1, std::string::npos, e.n) == 0) ti.kind = e.kind;
            }
        }
        if (!ti.kind || ti.layer < 0 || ti.layer >= cf.n_layers) { err = "Invalid tensor name:" + ti.name; return false; }
    }
    if (r.bad) { err = "Reading gguf tensor infos error:" + path; return false; }
    const size_t data0 = (r.pos + alignment - 1) / alignment * alignment;
    bool have_cls = false;
    for (auto& ti : infos) {
        size_t items = 1; for (int j = 0; j < ti.nd; ++j) items *= (size_t)ti.dims[j];
        const size_t bytes = ti.dtype == 0 ? items * 4 : ti.dtype == 1 ? items * 2 : 0;
        if (!bytes) { err = "Tensor " + ti.name + ": this data type is not supported (F32 and F16 are)"; return false; }
        if (!r.c.ok(data0 + ti.off, bytes)) { err = "Tensor data exceeds the file:" + ti.name; return false; }
        HostTensor t; t.kind = ti.kind; t.layer = ti.layer; t.qtype = 0;
        t.cols = (int)ti.dims[0]; t.rows = ti.nd > 1 ? (int)ti.dims[1] : 1;
        const uint8_t* src = r.c.base + data0 + ti.off;
        if (ti.dtype == 0) t.values = src;                                  // zero copy out of the mapping
        else {
            t.owned_values.resize(items * 4);
            float* dst = reinterpret_cast<float*>(t.owned_values.data());
            for (size_t i = 0; i < items; ++i) { uint16_t h; memcpy(&h, src + 2 * i, 2); dst[i] = half_to_float(h); }
        }
        if (ti.kind == 3) have_cls = true;
        m.tensors.push_back(std::move(t));
    }
    for (auto& t : m.tensors) if (!t.owned_values.empty()) t.values = t.owned_values.data();
    if (!have_cls) {                                       // tied embeddings: the classifier is the embedding table
        for (size_t i = 0; i < m.tensors.size(); ++i) if (m.tensors[i].kind == 1) {
            HostTensor t; t.kind = 3; t.layer = 0; t.qtype = 0; t.rows = m.tensors[i].rows; t.cols = m.tensors[i].cols; t.values = m.tensors[i].values;
            m.tensors.push_back(std::move(t)); break;
        }
    }
    return true;
}
} // namespace

ModelFile::~ModelFile() { if (map_base) munmap(map_base, map_size); }

void quantize_groups(const float* x, size_t n, int qtype, void* q, float* scales) {
    const float F = qtype == 2 ? 127.0f : 5792.0f;
    for (size_t g = 0; g * 64 < n; ++g) {
        const float* xg = x + g * 64;
        const size_t gn = std::min<size_t>(64, n - g * 64);
        float mx = 0.f;
        for (size_t j = 0; j < gn; ++j) mx = std::max(mx, fabsf(xg[j]));
        const float r = mx / F;
        scales[g] = r;
        for (size_t j = 0; j < gn; ++j) {
            const float t = xg[j] / r;
            const int v = (r == 0.f || t != t) ? 0 : (int)t;     // truncation; all-zero group -> 0
            if (qtype == 2) reinterpret_cast<int8_t*>(q)[g * 64 + j] = (int8_t)v; else reinterpret_cast<int16_t*>(q)[g * 64 + j] = (int16_t)v;
        }
    }
}

FileType detect_file_type(const std::string& path, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "Cannot open model file:" + path; return FileType::UNKNOWN; }
    unsigned char b[128] = {0};
    const size_t n = fread(b, 1, sizeof b, f);
    fclose(f);
    if (n < 28) { err = "Cannot read model file:" + path; return FileType::UNKNOWN; }
    uint32_t tag; memcpy(&tag, b, 4);
    if (tag == kFlmTag) return FileType::FLM;
    if (memcmp(b, "GGUF", 4) == 0) return FileType::GGUF;
    int h[7]; memcpy(h, b, 28);     // is_valid_llama2c_header (llama2c_loader.cpp:31-40)
    if (h[0] >= 512 && h[0] <= 64000 && h[1] >= 512 && h[1] <= 64000 && h[2] > 0 && h[2] < 512 && h[3] >= 4 && h[3] <= 1024 &&
        h[4] >= 1 && h[4] <= h[3] && h[5] >= 1000 && h[5] < (256 << 10)) return FileType::LLAMA2C;
    err = "Unsupported model file type";
    return FileType::UNKNOWN;
}

bool load_model_file(const std::string& ckpt, const std::string& tokenizer_path, FileType ft, bool tokenizer_only, bool debug,
                     ModelFile& out, std::string& err) {
    if (ft == FileType::UNKNOWN) ft = detect_file_type(ckpt, err);
    switch (ft) {
    case FileType::FLM: return load_flm(ckpt, tokenizer_only, debug, out, err);
    case FileType::LLAMA2C: return load_llama2c(ckpt, tokenizer_path, tokenizer_only, out, err);
    case FileType::GGUF: return load_gguf(ckpt, tokenizer_only, debug, out, err);
    default: if (err.empty()) err = "Unsupported model file type"; return false;
    }
}

} // namespace flmhost
