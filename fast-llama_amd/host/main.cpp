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
#include <iostream>
#include <string>
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

void parse(Args& a, int argc, const char** argv) {      // Arguments::parse (main.cpp:171-244)
    for (int i = 1; i < argc;) {
        const std::string arg = argv[i++];
        auto val = [&]() -> const char* { if (i >= argc) { usage(argv[0]); exit(-1); } return argv[i++]; };
        if (arg == "-j" || arg == "--threads") a.num_threads = atoi(val());
        else if (arg == "-q" || arg == "--quant") { const char* s = val(); if (!strcasecmp(s, "int16")) a.qtype = 1; else if (!strcasecmp(s, "int8")) a.qtype = 2; else if (!strcasecmp(s, "int4")) a.qtype = 3; }
        else if (arg == "--numa") a.use_numa = true;
        else if (arg == "--uma") a.use_numa = false;
        else if (arg == "--detail") a.detail = true;
        else if (arg == "-c" || arg == "--checkpoint") a.ckpt = val();
        else if (arg == "-z" || arg == "--tokenizer") a.tknr = val();
        else if (arg == "-f" || arg == "--file-type") { const char* v = val(); if (!strcasecmp(v, "flm")) a.ft = FileType::FLM; else if (!strcasecmp(v, "gguf")) a.ft = FileType::GGUF; else if (!strcasecmp(v, "llama2c")) a.ft = FileType::LLAMA2C; }
        else if (arg == "-i" || arg == "--prompt") a.prompt = val();
        else if (arg == "-e" || arg == "--encode") a.encode_str = val();
        else if (arg == "-d" || arg == "--decode") a.decode_str = val();
        else if (arg == "-n" || arg == "--max-new-tokens") a.max_tokens = atoi(val());
        else if (arg == "-p" || arg == "--topp") a.topp = (float)atof(val());
        else if (arg == "-t" || arg == "--temperature") a.temp = (float)atof(val());
        else if (arg == "--seed") a.seed = atoi(val());
        else if (arg == "--rounds") a.rounds = atoi(val());
        else if (arg == "--device") a.device = atoi(val());
        else if (arg == "--devices") {                       // a comma-separated list of 1 to 8 non-negative ordinals, nothing else
            const char* v = val(); a.devices.clear();
            bool ok = *v != 0;
            while (ok && *v) {
                char* e; const long d = strtol(v, &e, 10);
                ok = e != v && d >= 0 && d < 1024 && (*e == 0 || (*e == ',' && e[1] != 0)) && a.devices.size() < 8;
                if (ok) { a.devices.push_back((int)d); v = *e ? e + 1 : e; }
            }
            if (!ok) { fprintf(stderr, "Invalid --devices list:\x1b[31m%s\x1b[0m (expected 1 to 8 HIP device ordinals, e.g. 0,1,2,3)\n", argv[i - 1]); usage(argv[0]); exit(-1); }
        }
        else if (arg == "-m" || arg == "--mode") { const char* s = val(); if (!strcasecmp(s, "gen") || !strcasecmp(s, "generate")) a.mode = Mode::GEN; else if (!strcasecmp(s, "chat")) a.mode = Mode::CHAT; else if (!strcasecmp(s, "benchmark") || !strcasecmp(s, "bm")) a.mode = Mode::TEST; }
        else if (arg == "--debug") { a.debug = true; a.detail = true; }
        else if (arg == "-h" || arg == "--help") { usage(argv[0]); exit(0); }
        else { fprintf(stderr, "Unknown argument:\x1b[31m%s\x1b[0m\n", arg.c_str()); usage(argv[0]); exit(-1); }
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
