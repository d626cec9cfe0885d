// sampler.h -- llama2.c style sampler with the reference's exact arithmetic (src/transformer/sampler.cpp):
// argmax at temperature 0, otherwise temperature scaling, the reference's clipped softmax
// (src/blas/tf_operators.cpp:188-209), xorshift* coin and multinomial / top-p selection.
#pragma once
#include <stdint.h>

#include <vector>

namespace flmhost {

class Sampler {
public:
    void build(int vocab_size, uint64_t seed) { _n = vocab_size; _state = seed; _idx.resize(vocab_size); }
    int sample(float* logits, float temperature, float topp);     // modifies logits in place, like the reference
private:
    struct PI { float prob; int index; };
    float coin();
    int _n = 0;
    uint64_t _state = 0;      // seed 0 (the CLI default, transformer.cpp:40) keeps the state at 0: coin == 0 forever
    std::vector<PI> _idx;
};

} // namespace flmhost
