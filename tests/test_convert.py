"""fast-llama_amd/convert.py (HF LLaMA directory -> .flm) against the REFERENCE converter: tests/golden/hf_tiny/ is a tiny
HF-layout checkpoint (config.json, SentencePiece tokenizer.model, pytorch_model.bin) and tests/golden/hf_tiny_*.flm are what
/root/reference/tools/convert_flm.py made of it in the build container (tests/golden/make_hf_fixture.py).  The outputs must be
the same bytes; the result must load in this repo's readers and run."""
import ctypes as C
import os

import numpy as np
import pytest

import __graft_entry__ as graft
from fast_llama_amd import convert, flmfile as ff

GOLD = os.path.join(os.path.dirname(__file__), "golden")
HF = os.path.join(GOLD, "hf_tiny")


@pytest.mark.parametrize("out_type", ["int8", "int16"])
def test_converter_is_byte_identical_to_the_reference_tool(tmp_path, out_type):
    out = str(tmp_path / f"{out_type}.flm")
    convert.convert(HF, out, out_type, log=lambda *a: None)
    mine = open(out, "rb").read(); ref = open(os.path.join(GOLD, f"hf_tiny_{out_type}.flm"), "rb").read()
    assert len(mine) == len(ref)
    assert mine == ref


def test_f32_output_differs_from_the_reference_only_where_the_reference_forgets_the_permutation(tmp_path):
    """with -t f32 the reference tool reloads every tensor after permuting it (convert_flm.py:1160-1164) and so writes q_proj /
    k_proj in the HF layout its engine does not rotate correctly; this converter writes the permuted rows in every output type"""
    out = str(tmp_path / "f32.flm")
    convert.convert(HF, out, "f32", log=lambda *a: None)
    mine = open(out, "rb").read(); ref = open(os.path.join(GOLD, "hf_tiny_f32.flm"), "rb").read()
    assert len(mine) == len(ref)
    cm, tm, xm = ff.read_flm(out); cr, tr, xr = ff.read_flm(os.path.join(GOLD, "hf_tiny_f32.flm"))
    assert tm.texts == tr.texts and set(xm) == set(xr)
    for key, v in xr.items():
        if key[0] in (ff.T_ATTN_Q, ff.T_ATTN_K):
            assert np.array_equal(xm[key], convert.permute_qk(np.asarray(v), cr.n_heads)) and not np.array_equal(xm[key], v)
        else:
            assert np.array_equal(xm[key], v)


def test_q_k_rows_are_permuted_to_interleaved_pairs():
    # HF layout of one head: [first halves | second halves]; the engine rotates adjacent pairs (2j, 2j+1)
    w = np.arange(2 * 8 * 3, dtype=np.float32).reshape(16, 3)          # 2 heads x 8 rows
    p = convert.permute_qk(w, 2)
    for h in range(2):
        for j in range(4):
            assert np.array_equal(p[h * 8 + 2 * j], w[h * 8 + j]) and np.array_equal(p[h * 8 + 2 * j + 1], w[h * 8 + 4 + j])


def test_converted_file_loads_in_both_readers(tmp_path):
    out = str(tmp_path / "m.flm")
    convert.convert(HF, out, "int8", log=lambda *a: None)
    cfg, tok, tensors = ff.read_flm(out)
    assert (cfg.dim, cfg.hidden_dim, cfg.n_layers, cfg.n_heads, cfg.vocab_size, cfg.quant_type) == (64, 128, 2, 2, 320, ff.QT_INT8)
    assert len(tok.texts) == 320 and (tok.bos, tok.eos) == (1, 2)
    assert isinstance(tensors[(ff.T_ATTN_Q, 1)], tuple) and not isinstance(tensors[(ff.T_TOKEN_EMBD, 0)], tuple)   # embedding stays fp32
    # the product's C++ reader
    lib = os.path.join(graft.PKG_DIR, "lib", "libflm_host.so")
    if not os.path.exists(lib):
        graft.build()
    h = C.CDLL(lib)
    h.fh_open.restype = C.c_void_p; h.fh_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    h.fh_config.argtypes = [C.c_void_p, C.POINTER(C.c_int)]; h.fh_n_tensors.argtypes = [C.c_void_p]; h.fh_vocab_size.argtypes = [C.c_void_p]
    h.fh_close.argtypes = [C.c_void_p]; h.fh_last_error.restype = C.c_char_p
    hd = h.fh_open(out.encode(), b"", 0, 0)
    assert hd, h.fh_last_error()
    c = (C.c_int * 9)(); h.fh_config(hd, c)
    assert list(c)[:6] == [64, 128, 2, 2, 2, 320] and c[7] == ff.QT_INT8
    assert h.fh_n_tensors(hd) == 3 + 9 * 2 and h.fh_vocab_size(hd) == 320
    h.fh_close(hd)


@pytest.mark.gpu
def test_converted_model_runs_and_matches_the_oracle(gpu, tmp_path):
    import oracle_py as O
    out = str(tmp_path / "m.flm")
    convert.convert(HF, out, "int8", log=lambda *a: None)
    cfg, tok, tensors = ff.read_flm(out)
    cfg.max_length = 256
    ctx = gpu.Ctx(gpu.desc_from_config(cfg)); ctx.upload_all(tensors)
    om = O.OracleModel(cfg, tensors)
    prompt = np.array([1, 20, 33, 47], np.int32)
    a = ctx.forward(prompt, 0); b = om.forward(prompt, 0)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    ctx.close()
