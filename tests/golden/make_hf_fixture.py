"""Builds the tiny HF-layout LLaMA checkpoint under tests/golden/hf_tiny/ and converts it with the REFERENCE tool
(/root/reference/tools/convert_flm.py, run in the build container) into tests/golden/hf_tiny_{int8,int16,f32}.flm.
tests/test_convert.py checks that this repo's converter reproduces those files byte for byte.
Run:  python tests/golden/make_hf_fixture.py        (needs /root/reference, sentencepiece, torch)"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HF = os.path.join(HERE, "hf_tiny")


def build_hf_dir():
    import sentencepiece as spm
    import torch
    os.makedirs(HF, exist_ok=True)
    rng = np.random.default_rng(11)
    words = ["the", "shape", "of", "it", "tea", "time", "long", "story", "village", "brave", "boy", "mountain", "small", "beautiful",
             "was", "lived", "in", "a", "an", "and", "that", "about", "Oliver", "Europe", "ancient", "happened", "résumé", "naïve", "OK"]
    corpus = os.path.join(HF, "_corpus.txt")
    with open(corpus, "w", encoding="utf-8") as f:
        for _ in range(400):
            f.write(" ".join(rng.choice(words, size=int(rng.integers(4, 12)))) + ".\n")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(HF, "tokenizer"), vocab_size=320, model_type="bpe",
                                   byte_fallback=True, character_coverage=1.0, bos_id=1, eos_id=2, unk_id=0, pad_id=-1,
                                   minloglevel=2)
    os.remove(corpus); os.remove(os.path.join(HF, "tokenizer.vocab"))
    dim, hidden, heads, layers, vocab = 64, 128, 2, 2, 320
    conf = {"_name_or_path": "hf-tiny", "architectures": ["LlamaForCausalLM"], "bos_token_id": 1, "eos_token_id": 2, "hidden_act": "silu",
            "hidden_size": dim, "initializer_range": 0.02, "intermediate_size": hidden, "max_position_embeddings": 256,
            "model_type": "llama", "num_attention_heads": heads, "num_hidden_layers": layers, "num_key_value_heads": heads,
            "pad_token_id": 0, "rms_norm_eps": 1e-05, "tie_word_embeddings": False, "torch_dtype": "float32", "use_cache": True,
            "vocab_size": vocab}
    json.dump(conf, open(os.path.join(HF, "config.json"), "w"), indent=2)
    sd = {}
    def t(*shape, scale=0.05):
        return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))
    sd["model.embed_tokens.weight"] = t(vocab, dim)
    for l in range(layers):
        p = f"model.layers.{l}."
        sd[p + "self_attn.q_proj.weight"] = t(dim, dim); sd[p + "self_attn.k_proj.weight"] = t(dim, dim)
        sd[p + "self_attn.v_proj.weight"] = t(dim, dim); sd[p + "self_attn.o_proj.weight"] = t(dim, dim)
        sd[p + "mlp.gate_proj.weight"] = t(hidden, dim); sd[p + "mlp.down_proj.weight"] = t(dim, hidden); sd[p + "mlp.up_proj.weight"] = t(hidden, dim)
        sd[p + "input_layernorm.weight"] = 1 + t(dim, scale=0.1); sd[p + "post_attention_layernorm.weight"] = 1 + t(dim, scale=0.1)
    sd["model.norm.weight"] = 1 + t(dim, scale=0.1)
    sd["lm_head.weight"] = t(vocab, dim)
    torch.save(sd, os.path.join(HF, "pytorch_model.bin"))


def run_reference():
    for ty in ("int8", "int16", "f32"):
        out = os.path.join(HERE, f"hf_tiny_{ty}.flm")
        subprocess.run([sys.executable, "/root/reference/tools/convert_flm.py", "-m", HF, "-b", "spm", "-t", ty, "-o", out], check=True,
                       stdout=subprocess.DEVNULL)
        print("reference wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    build_hf_dir()
    run_reference()
