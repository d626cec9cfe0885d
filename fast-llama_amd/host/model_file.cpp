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
    struct stat st; fstat(fd, &st);
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
            auto str_at = [&](uint32_t off) { const char* p = reinterpret_cast<const char*>(c.base + text + off); return std::string(p, strnlen(p, tsz - off)); };
            m.vocab.tokens.resize(n);
            for (uint32_t i = 0; i < n; ++i) {
                const size_t it = items + (size_t)16 * i;
                auto& t = m.vocab.tokens[i];
                t.index_text = str_at(c.at<uint32_t>(it)); t.show_text = str_at(c.at<uint32_t>(it + 4));
                t.type = (int)c.at<uint32_t>(it + 8); t.score = c.at<float>(it + 12);
            }
            m.vocab.conn_tag = str_at(conn_pos);
            m.vocab.bos = special[1]; m.vocab.eos = special[2]; m.vocab.pad = special[3];
            if (tokenizer_only) return true;
        } else if (b.type == BT_TENSOR) {
            // tensor header (flm_loader.cpp:165-177): shape[4] u32, tensor_type u16, layer_id u16, scales_size u32
            const size_t hp = pos + 16;
            uint32_t shape[4]; memcpy(shape, c.base + hp, 16);
            HostTensor t;
            t.kind = c.at<uint16_t>(hp + 16); t.layer = c.at<uint16_t>(hp + 18);
            const uint32_t ssz = c.at<uint32_t>(hp + 20);
            int nd = 0; while (nd < 4 && shape[nd] > 0) ++nd;
            if (nd == 1) { t.rows = 1; t.cols = (int)shape[0]; } else if (nd == 2) { t.rows = (int)shape[0]; t.cols = (int)shape[1]; }
            else { err = "Unsupported tensor rank in block:" + b.name; return false; }
            t.qtype = b.dtype == DT_INT8 ? 2 : b.dtype == DT_INT16 ? 1 : 0;
            if (b.dtype != DT_INT8 && b.dtype != DT_INT16 && b.dtype != DT_FLOAT32) { err = "Unsupported tensor data type in block:" + b.name; return false; }
            const size_t esz = t.qtype == 2 ? 1 : t.qtype == 1 ? 2 : 4, vbytes = (size_t)t.rows * t.cols * esz;
            const size_t d0 = pos + b.header_size;
            if (vbytes + (size_t)ssz * 4 > b.data_size) { err = "Tensor block too small:" + b.name; return false; }
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
    return true;
}

bool load_llama2c(const std::string& ckpt, const std::string& tok_path, bool tokenizer_only, ModelFile& m, std::string& err) {
    FILE* f = fopen(ckpt.c_str(), "rb");
    if (!f) { err = "Failed to open model file:" + ckpt; return false; }
    int h[7];
    if (fread(h, 4, 7, f) != 7) { fclose(f); err = "Reading model file error:" + ckpt; return false; }
    Config& c = m.cfg;
    c.dim = h[0]; c.hidden_dim = h[1]; c.n_layers = h[2]; c.n_heads = h[3]; c.n_kv_heads = h[4]; c.vocab_size = abs(h[5]); c.max_seq_len = h[6];
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
        t.owned_values.resize((size_t)rows * cols); t.owned_scales.resize((size_t)rows * cols / 64);
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
    case FileType::GGUF: err = "gguf files are not supported by this build yet (F32 gguf is planned; the reference's Q8_0 path is broken, utility.cpp:64)"; return false;
    default: if (err.empty()) err = "Unsupported model file type"; return false;
    }
}

} // namespace flmhost
