// main.cpp -- drop-in for the reference CLI (src/main.cpp): the same flags, the same prompt/output/summary
// lines, the per-token transformer running on an MI355X through include/flm_gpu.h.
//   ./main -c model.flm -q int8 -i "prompt" [-n 512] [-t 1.0] [-p 0.9] [-j N] [--mode gen|chat|bm] [--rounds R]
// Extra flags of this build: --device <hip ordinal>; --devices a,b,... = the reference's parallel width (-j, main.cpp:30,78) on GPUs: one
// sequence sharded over the named devices (split_rows, transformer.cpp:264-287), one host thread per device.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include <chrono>
#include <initializer_list>
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "engine.h"

using namespace flmhost;

namespace {
enum class Mode { GEN, CHAT, TEST };
struct Args {
    std::string ckpt, tknr, encode_str, decode_str;
    FileType ft = FileType::UNKNOWN;
    const char* prompt = "";
    int num_threads = -1, max_tokens = 512, qtype = 2, rounds = 0, device = 0, seed = 128391297;
    float topp = 0.9f, temp = 1.0f;
    bool use_numa = false, detail = false, debug = false;
    std::vector<int> devices;
    const char* bad_devices = nullptr;      // a --devices argument that did not parse
    Mode mode = Mode::GEN;
};
const char* Y = "\x1b[33m"; const char* G = "\x1b[32m"; const char* E = "\x1b[0m";

void usage(const char* bin) {
    fprintf(stderr, "Usage:\n   %s [OPTIONS]\nOptions:\n", bin);
    fprintf(stderr, "   --checkpoint,-c   <string>    the file path of the model checkpoint\n");
    fprintf(stderr, "   --tokenizer,-z    <string>    the file path of the tokenizer\n");
    fprintf(stderr, "   --file-type,-f    <string>    flm | gguf | llama2c\n");
    fprintf(stderr, "   --mode            <string>    gen | chat | benchmark(bm)\n");
    fprintf(stderr, "   --prompt,-i       <string>    the input prompt text\n");
    fprintf(stderr, "   --max-tokens,-n   <integer>   the maximum number of generated tokens\n");
    fprintf(stderr, "   --temperature,-t  <float>     the value for temperature sampling, [0, 1]\n");
    fprintf(stderr, "   --topp,-p         <float>     the value for top-p sampling, [0, 1]\n");
    fprintf(stderr, "   --quant,-q        <string>    quantization type, can be INT8, INT16\n");
    fprintf(stderr, "   --threads,-j      <number>    accepted for compatibility (the GPU build has no worker threads)\n");
    fprintf(stderr, "   --device          <number>    HIP device ordinal (this build only)\n");
    fprintf(stderr, "   --devices         <a,b,...>   shard ONE sequence over these HIP devices, 1 to 8 of them (this build only)\n");
    fprintf(stderr, "   --encode,-e       <string>    encode the input string into tokens\n");
    fprintf(stderr, "   --decode,-d       <string>    decode the input tokens to text\n");
    fprintf(stderr, "   --help,-h                     print this message\n");
}

// The flag set is the drop-in surface (the reference's Arguments::parse, main.cpp:171-244, accepts the same spellings); here it is a table: one row per flag with what it
// stores, so that the two builds' flags can be compared row by row and a new flag is one line.
namespace {
bool parse_devices(const char* v, std::vector<int>& out) {      // a comma-separated list of 1 to 8 non-negative ordinals, nothing else
    out.clear();
    bool ok = *v != 0;
    while (ok && *v) {
        char* e; const long d = strtol(v, &e, 10);
        ok = e != v && d >= 0 && d < 1024 && (*e == 0 || (*e == ',' && e[1] != 0)) && out.size() < 8;
        if (ok) { out.push_back((int)d); v = *e ? e + 1 : e; }
    }
    return ok;
}
template <class E> bool pick(const char* v, std::initializer_list<std::pair<const char*, E>> names, E& out) {     // case-insensitive keyword -> enum; unknown words leave `out` alone (as the reference does)
    for (const auto& n : names) if (!strcasecmp(v, n.first)) { out = n.second; return true; }
    return false;
}
struct Flag {
    const char* short_name; const char* long_name; bool takes_value;
    void (*apply)(Args& a, const char* v);
};
const Flag kFlags[] = {
    {"-j", "--threads",        true,  [](Args& a, const char* v) { a.num_threads = atoi(v); }},
    {"-q", "--quant",          true,  [](Args& a, const char* v) { pick<int>(v, {{"int16", 1}, {"int8", 2}, {"int4", 3}}, a.qtype); }},
    {nullptr, "--numa",        false, [](Args& a, const char*) { a.use_numa = true; }},
    {nullptr, "--uma",         false, [](Args& a, const char*) { a.use_numa = false; }},
    {nullptr, "--detail",      false, [](Args& a, const char*) { a.detail = true; }},
    {nullptr, "--debug",       false, [](Args& a, const char*) { a.debug = true; a.detail = true; }},
    {"-c", "--checkpoint",     true,  [](Args& a, const char* v) { a.ckpt = v; }},
    {"-z", "--tokenizer",      true,  [](Args& a, const char* v) { a.tknr = v; }},
    {"-f", "--file-type",      true,  [](Args& a, const char* v) { pick<FileType>(v, {{"flm", FileType::FLM}, {"gguf", FileType::GGUF}, {"llama2c", FileType::LLAMA2C}}, a.ft); }},
    {"-i", "--prompt",         true,  [](Args& a, const char* v) { a.prompt = v; }},
    {"-e", "--encode",         true,  [](Args& a, const char* v) { a.encode_str = v; }},
    {"-d", "--decode",         true,  [](Args& a, const char* v) { a.decode_str = v; }},
    {"-n", "--max-new-tokens", true,  [](Args& a, const char* v) { a.max_tokens = atoi(v); }},
    {"-p", "--topp",           true,  [](Args& a, const char* v) { a.topp = (float)atof(v); }},
    {"-t", "--temperature",    true,  [](Args& a, const char* v) { a.temp = (float)atof(v); }},
    {nullptr, "--seed",        true,  [](Args& a, const char* v) { a.seed = atoi(v); }},
    {nullptr, "--rounds",      true,  [](Args& a, const char* v) { a.rounds = atoi(v); }},
    {"-m", "--mode",           true,  [](Args& a, const char* v) { pick<Mode>(v, {{"gen", Mode::GEN}, {"generate", Mode::GEN}, {"chat", Mode::CHAT}, {"benchmark", Mode::TEST}, {"bm", Mode::TEST}}, a.mode); }},
    {nullptr, "--device",      true,  [](Args& a, const char* v) { a.device = atoi(v); }},                                  // (this build only)
    {nullptr, "--devices",     true,  [](Args& a, const char* v) { if (!parse_devices(v, a.devices)) a.bad_devices = v; }},   // (this build only)
};
}
void parse(Args& a, int argc, const char** argv) {
    for (int i = 1; i < argc;) {
        const std::string arg = argv[i++];
        if (arg == "-h" || arg == "--help") { usage(argv[0]); exit(0); }
        const Flag* hit = nullptr;
        for (const Flag& f : kFlags) if ((f.short_name && arg == f.short_name) || arg == f.long_name) { hit = &f; break; }
        if (!hit) { fprintf(stderr, "Unknown argument:\x1b[31m%s\x1b[0m\n", arg.c_str()); usage(argv[0]); exit(-1); }
        const char* v = nullptr;
        if (hit->takes_value) { if (i >= argc) { usage(argv[0]); exit(-1); } v = argv[i++]; }
        hit->apply(a, v);
        if (a.bad_devices) { fprintf(stderr, "Invalid --devices list:\x1b[31m%s\x1b[0m (expected 1 to 8 HIP device ordinals, e.g. 0,1,2,3)\n", a.bad_devices); usage(argv[0]); exit(-1); }
    }
    if (a.rounds < 1) a.rounds = a.mode == Mode::TEST ? 16 : 1;
}

int64_t now_us() { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now().time_since_epoch()).count(); }

int encode_decode(const Args& a) {                       // main.cpp:246-286
    ModelFile mf; std::string err;
    if (!load_model_file(a.ckpt, a.tknr, FileType::UNKNOWN, true, false, mf, err)) { fprintf(stderr, "Failed to load model\n%s\n", err.c_str()); return -1; }
    Tokenizer tk; tk.set_vocab(std::move(mf.vocab));
    auto show = [](const char* title, const std::vector<int>& v) { print_vector(title, v); };
    if (!a.decode_str.empty()) {
        std::vector<int> toks; const char* s = a.decode_str.c_str();
        while (*s) { if (*s == '-' || (*s >= '0' && *s <= '9')) { char* e; toks.push_back((int)strtol(s, &e, 10)); s = e; } else ++s; }
        show("tokens: ", toks);
        std::cout << "  text: " << tk.decode(toks) << std::endl;
    } else {
        std::cout << "  text: " << a.encode_str << std::endl;
        show("tokens: ", tk.encode(a.encode_str, true));
    }
    return 0;
}
} // namespace

int main(int argc, const char** argv) {
    Args args; parse(args, argc, argv);
    if (!args.encode_str.empty() || !args.decode_str.empty()) return encode_decode(args);
    if (!args.prompt || !args.prompt[0])
        args.prompt = "That was a long long story happened in the ancient Europe. It was about a brave boy name Oliver. Oliver lived in a small village among many big moutains. It was a beautiful village.";
    if (args.detail) {
        fprintf(stderr, "num_threads:%s%d%s\n   use_numa:%s%d%s\n  ckpt_path:%s%s%s\n  tknr_path:%s%s%s\n      top_p:%s%g%s\ntemperature:%s%g%s\n\n",
                Y, args.num_threads, E, Y, (int)args.use_numa, E, Y, args.ckpt.c_str(), E, Y, args.tknr.c_str(), E, Y, args.topp, E, Y, args.temp, E);
    }
    if (args.devices.empty()) args.devices.push_back(args.device);
    // ranks sharing a GPU wait for each other inside their launches: each needs a hardware queue of its own (HIP's default is 4 per process)
    if (args.devices.size() > 1) setenv("GPU_MAX_HW_QUEUES", "16", 0);
    GpuTransformer tf(args.detail || args.debug);
    if (!tf.load(args.ckpt, args.tknr, args.ft, args.qtype, args.devices)) { fprintf(stderr, "Failed to load model\n%s\n", tf.error().c_str()); return 1; }
    args.qtype = tf.get_quant_type();
    if (args.detail) fprintf(stderr, "Model loaded\n\n");

    double sum_prompt_tok = 1e-10, sum_output_tok = 1e-10, sum_prompt_ms = 1e-10, sum_output_ms = 1e-10;
    for (int r = 0; r < args.rounds; ++r) {
        int prompt_tokens = 0, output_tokens = 0; int64_t first_us = 0;
        const int64_t t0 = now_us();
        auto cb = [&](const char* text, int n_in, int n_out, bool ended) -> bool {
            if (first_us == 0) {
                if (args.mode != Mode::TEST) { printf("prompt: %s%s%s\n", Y, args.prompt, E); printf("output: %s", G); }
                first_us = now_us() - t0; prompt_tokens = n_in;
            } else output_tokens = n_out;
            if (args.mode != Mode::TEST && text) { printf("%s", text); fflush(stdout); }
            return !ended;
        };
        if (!tf.generate(args.prompt, cb, args.max_tokens, args.temp, args.topp) && !tf.error().empty()) { fprintf(stderr, "%s\n", tf.error().c_str()); return 1; }
        const int64_t total_us = now_us() - t0;
        if (args.mode != Mode::TEST) printf("%s\n\n", E);
        sum_prompt_tok += prompt_tokens; sum_output_tok += output_tokens;
        sum_prompt_ms += first_us / 1000.; sum_output_ms += (total_us - first_us) / 1000.;
    }
    const double pt = sum_prompt_tok / args.rounds, ot = sum_output_tok / args.rounds, pm = sum_prompt_ms / args.rounds, om = sum_output_ms / args.rounds;
    const double first_lat = pm / pt, later_lat = om / (ot - 1);
    const char* qn = args.qtype == 2 ? "int8" : args.qtype == 1 ? "int16" : args.qtype == 3 ? "int4" : "None";
    // the reference's summary line (main.cpp:136-145); simd_size reports the wavefront width here
    printf("num_threads:%s%2d%s\tquant:%s%s%s\tuse_numa:%s%d%s\tsimd_size:%d\tprompt_size:%3d\toutput_size:%3d\ttotal_latancy:%5.0fms\t"
           "prompt_token_latancy:%s%4.2f%sms\toutput_token_latancy:%s%4.2f%sms\tprompt_speed:%s%5.1f%stps\toutput_speed:%s%5.1f%stps\n",
           Y, args.num_threads, E, G, qn, E, G, (int)args.use_numa, E, 64, (int)pt, (int)ot, pm + om,
           Y, first_lat, E, Y, later_lat, E, G, 1000. / first_lat, E, G, 1000. / later_lat, E);
    return 0;
}
