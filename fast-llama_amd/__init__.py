"""fast-llama_amd: MI355X-native implementation of fast-llama's per-token transformer hot path.

The product is the C-ABI shared library built from `csrc/` (hand-written HIP for gfx950) plus the
C++ host (`host/`: `.flm`/llama2.c loaders, tokenizer, sampler, `main` CLI).  The Python modules here
are plumbing for tests and `bench.py`: a ctypes binding (`capi`), the `.flm` file format (`flmfile`)
and seeded synthetic checkpoints (`synth`).

The directory name contains a hyphen (it mirrors the reference's name); import it through
`__graft_entry__.load_package()` which registers it as `fast_llama_amd`.
"""
__all__ = ["flmfile", "synth", "capi"]
