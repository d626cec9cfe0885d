"""`.flm` format: this repo's writer is byte-identical to the reference's own writer
(tests/golden/ref_writer.flm was produced by /root/reference/tools/convert_flm.py's FLFWriter in the
build container), and the reader round-trips."""
import os

import numpy as np

from fast_llama_amd import flmfile as ff, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _tiny():
    cfg = synth.make_config((64, 128, 1, 1, 264), ff.QT_INT8)
    return cfg, synth.make_tokenizer(cfg.vocab_size), synth.make_tensors(cfg, seed=3)


def test_writer_is_byte_identical_to_reference_writer(tmp_path):
    cfg, tok, tensors = _tiny()
    p = tmp_path / "mine.flm"
    ff.write_flm(str(p), cfg, tok, tensors)
    mine = open(p, "rb").read(); ref = open(os.path.join(GOLD, "ref_writer.flm"), "rb").read()
    assert len(mine) == len(ref)
    assert mine == ref


def test_reader_roundtrip(tmp_path):
    cfg, tok, tensors = _tiny()
    p = tmp_path / "rt.flm"
    ff.write_flm(str(p), cfg, tok, tensors)
    c2, t2, x2 = ff.read_flm(str(p))
    for k in ("dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "quant_type", "quant_group_size", "max_length", "name"):
        assert getattr(c2, k) == getattr(cfg, k), k
    assert t2.texts == tok.texts and t2.types == tok.types and np.allclose(t2.scores, tok.scores)
    assert (t2.bos, t2.eos, t2.pad) == (1, 2, 0)
    assert set(x2) == set(tensors)
    for k, v in tensors.items():
        if isinstance(v, tuple):
            assert np.array_equal(x2[k][0], v[0]) and np.array_equal(x2[k][1], v[1])
        else:
            assert np.array_equal(x2[k], v)


def test_reference_written_file_reads_back():
    c, t, x = ff.read_flm(os.path.join(GOLD, "ref_writer.flm"))
    assert (c.dim, c.hidden_dim, c.n_layers, c.vocab_size, c.quant_type) == (64, 128, 1, 264, ff.QT_INT8)
    assert len(t.texts) == 264 and t.texts[3] == "<0x00>"
    assert x[(ff.T_ATTN_Q, 0)][0].dtype == np.int8 and x[(ff.T_ATTN_Q, 0)][1].shape == (64, 1)
