#include "sampler.h"

#include <math.h>
#include <stdlib.h>

namespace flmhost {

float Sampler::coin() {
    _state ^= _state >> 12; _state ^= _state << 25; _state ^= _state >> 27;
    const uint32_t r = (uint32_t)((_state * 0x2545F4914F6CDD1Dull) >> 32);
    return (r >> 8) / 16777216.0f;
}

static int argmax(const float* p, int n) {
    int bi = 0; float bv = p[0];
    for (int i = 1; i < n; ++i) if (p[i] > bv) { bv = p[i]; bi = i; }
    return bi;
}

// cpuft::softmax (tf_operators.cpp:188-209): exp(x - max) with x - max < -15 clamped to 0, then scaled by 1/sum
static void clipped_softmax(float* x, int n) {
    float mx = x[0];
    for (int i = 1; i < n; ++i) if (x[i] > mx) mx = x[i];
    float sum = 0.f;
    for (int i = 0; i < n; ++i) {
        const float d = x[i] - mx;
        if (d < -15) x[i] = 0.f; else { x[i] = expf(d); sum += x[i]; }      // d <= 0 always, so the d >= 6 table branch never fires
    }
    const float inv = (float)(1. / sum);
    for (int i = 0; i < n; ++i) x[i] *= inv;
}

int Sampler::sample(float* logits, float temperature, float topp) {
    if (temperature == 0.0f) return argmax(logits, _n);
    for (int i = 0; i < _n; ++i) logits[i] /= temperature;
    clipped_softmax(logits, _n);
    const float c = coin();
    if (topp <= 0 || topp >= 1) {
        float cdf = 0.f;
        for (int i = 0; i < _n; ++i) { cdf += logits[i]; if (c < cdf) return i; }
        return _n - 1;
    }
    int n0 = 0;
    const float cutoff = (1.0f - topp) / (_n - 1);
    for (int i = 0; i < _n; ++i) if (logits[i] >= cutoff) { _idx[n0].index = i; _idx[n0].prob = logits[i]; ++n0; }
    qsort(_idx.data(), n0, sizeof(PI), [](const void* a, const void* b) {
        const float pa = ((const PI*)a)->prob, pb = ((const PI*)b)->prob; return pa > pb ? -1 : pa < pb ? 1 : 0; });
    float cum = 0.f; int last = n0 - 1;
    for (int i = 0; i < n0; ++i) { cum += _idx[i].prob; if (cum > topp) { last = i; break; } }
    const float r = c * cum;
    float cdf = 0.f;
    for (int i = 0; i <= last; ++i) { cdf += _idx[i].prob; if (r < cdf) return _idx[i].index; }
    return _idx[last].index;
}

} // namespace flmhost
