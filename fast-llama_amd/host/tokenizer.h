// tokenizer.h -- SentencePiece-BPE style greedy-merge tokenizer with byte fallback, as the reference's
// Tokenizer (src/transformer/tokenizer.cpp:247-401) behaves; written from scratch on std:: containers.
#pragma once
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "model_file.h"

namespace flmhost {

class Tokenizer {
public:
    void set_vocab(Vocab v);
    int vocab_size() const { return (int)_v.tokens.size(); }
    // encode(): optional BOS, one token per UTF-8 character (byte tokens id = byte + 3 when a character is
    // unknown), then repeatedly merge the adjacent pair whose concatenation has the best score.
    std::vector<int> encode(std::string_view text, bool add_bos = true, bool add_eos = false) const;
    // decode one token after `prev` (leading space stripped after BOS, <0xXX> byte tokens, unsafe bytes dropped)
    std::string decode(int token, int prev_token = -1) const;
    std::string decode(const std::vector<int>& tokens) const;
private:
    int lookup(std::string_view s) const;
    Vocab _v;
    std::unordered_map<std::string, int> _ids;
    int _underline = -1;
};

} // namespace flmhost
