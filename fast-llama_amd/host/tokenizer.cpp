#include "tokenizer.h"

#include <ctype.h>
#include <stdio.h>

namespace flmhost {

void Tokenizer::set_vocab(Vocab v) {
    _v = std::move(v);
    _ids.clear();
    _ids.reserve(_v.tokens.size() * 2);
    for (int i = 0; i < (int)_v.tokens.size(); ++i) _ids[_v.tokens[i].index_text] = i;   // later ids win, as build_text2id_map does
    auto it = _ids.find(_v.conn_tag);
    _underline = it == _ids.end() ? -1 : it->second;
}

int Tokenizer::lookup(std::string_view s) const {
    if (s == " ") return _underline;                     // search_text(" ") (tokenizer.cpp:236-238)
    auto it = _ids.find(std::string(s));
    return it == _ids.end() ? -1 : it->second;
}

std::vector<int> Tokenizer::encode(std::string_view text, bool add_bos, bool add_eos) const {
    std::vector<int> out;
    if (text.empty() || _v.tokens.empty()) return out;
    if (add_bos) out.push_back(_v.bos);
    // split into UTF-8 characters (at most 4 bytes, as the reference's scanner does)
    std::string cur;
    for (size_t i = 0; i < text.size(); ++i) {
        const unsigned char c = (unsigned char)text[i];
        if ((c & 0xC0) != 0x80) cur.clear();
        cur.push_back((char)c);
        const unsigned char nx = i + 1 < text.size() ? (unsigned char)text[i + 1] : 0;
        if ((nx & 0xC0) == 0x80 && cur.size() < 4) continue;
        const int id = lookup(cur);
        if (id >= 0) out.push_back(id);
        else for (unsigned char b : cur) out.push_back((int)b + 3);                  // byte fallback (+3: <unk>,<s>,</s>)
        cur.clear();
    }
    // greedy merges by score
    for (;;) {
        float best_score = -1e10f; int best_id = -1, best_idx = -1;
        for (int i = 0; i + 1 < (int)out.size(); ++i) {
            if (out[i] < 0 || out[i] >= vocab_size() || out[i + 1] < 0 || out[i + 1] >= vocab_size()) continue;
            const std::string merged = _v.tokens[out[i]].index_text + _v.tokens[out[i + 1]].index_text;
            const int id = lookup(merged);
            if (id != -1 && _v.tokens[id].score > best_score) { best_score = _v.tokens[id].score; best_id = id; best_idx = i; }
        }
        if (best_idx < 0) break;
        out[best_idx] = best_id;
        out.erase(out.begin() + best_idx + 1);
    }
    if (add_eos) out.push_back(_v.eos);
    return out;
}

std::string Tokenizer::decode(int token, int prev_token) const {
    if (token < 0 || token >= vocab_size()) return "";
    const std::string& show = _v.tokens[token].show_text;
    const char* piece = show.c_str();
    if (prev_token == 1 && piece[0] == ' ') ++piece;      // strip the space that follows BOS (tokenizer.cpp:349-351)
    unsigned char byte;
    char one[2] = {0, 0};
    if (sscanf(piece, "<0x%02hhX>", &byte) == 1) { one[0] = (char)byte; piece = one; }
    if (piece[0] == '\0') return "";
    if (piece[1] == '\0') { const unsigned char b = (unsigned char)piece[0]; if (!(isprint(b) || isspace(b))) return ""; }
    return piece;
}

std::string Tokenizer::decode(const std::vector<int>& tokens) const {
    std::string res; int prev = -1;
    for (int t : tokens) { res += decode(t, prev); prev = t; }
    return res;
}

} // namespace flmhost
