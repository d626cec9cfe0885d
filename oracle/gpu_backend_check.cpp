// oracle/gpu_backend_check.cpp -- TEST INFRASTRUCTURE (build container: needs the reference's headers and objects).
// Proves that include/flm_gpu.h binds inside the reference: oracle/gpu_backend.h (the binding of INTEGRATION.md) is compiled
// against the reference's own TransformerModel / Tensor, linked with the reference's objects and with libflm_gpu.so, and
//   gpu_backend_check model.flm [n_decode]
// loads the file with the REFERENCE loader, runs ParallelTransformer::forward (the reference CPU path) and GpuBackend::forward
// on the same prompt and greedy continuation and compares every logit bit for bit.
// Exit codes: 0 identical; 3 no usable GPU (the binding still compiled, linked and loaded -- what the build container can
// show); 1 mismatch or error.
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "transformer.h"             // reference (compiled with -fno-access-control: forward is private)
#include "gpu_backend.h"

using namespace cpuft;

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s model.flm [n_decode]\n", argv[0]); return 1; }
    const int ndec = argc > 2 ? atoi(argv[2]) : 4;
    TransformerModel tf;
    if (!tf.load(argv[1])) { fprintf(stderr, "reference loader failed on %s\n", argv[1]); return 1; }
    GpuBackend gpu;
    if (!gpu.init(tf, 0)) { printf("binding compiled, linked and loaded; no usable GPU here: %s\n", gpu.error()); return 3; }
    ParallelTransformer ref(false);
    if (!ref.load(argv[1], "", ModelFileType::UNKNOWN, QuantType::INT8, 2, false, 64, 0)) { fprintf(stderr, "reference model failed to load\n"); return 1; }
    const int V = tf.conf.vocab_size;
    std::vector<int> prompt = {1};
    for (int i = 1; i < 8; ++i) prompt.push_back(int((long)i * 7919 % V));
    std::vector<float> lg(V);
    std::vector<int> cur = prompt;
    int pos = 0, bad = 0;
    for (int step = 0; step <= ndec; ++step) {
        Tensor lr;
        ref.forward(std::span<const int>(cur.data(), cur.size()), pos, lr);
        if (!gpu.forward(std::span<const int>(cur.data(), cur.size()), pos, lg.data())) { fprintf(stderr, "gpu forward failed: %s\n", gpu.error()); return 1; }
        if (memcmp(lr.float_data(), lg.data(), sizeof(float) * size_t(V)) != 0) { ++bad; printf("step %d: logits differ\n", step); }
        int best = 0; for (int i = 1; i < V; ++i) if (lg[i] > lg[best]) best = i;
        pos += int(cur.size()); cur.assign(1, best);
    }
    printf("%s: %d forwards through the reference binding, logits %s the reference CPU path's\n", argv[1], ndec + 1, bad ? "DIFFER from" : "bit-identical to");
    return bad ? 1 : 0;
}
