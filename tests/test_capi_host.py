"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every symbol include/flm_gpu.h
declares, and the pure-host parts (shard plan, argument validation) behave."""
import ctypes as C
import os
import re

import pytest

from fast_llama_amd import capi, flmfile as ff, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "flm_gpu.h")).read()
    declared = set(re.findall(r"\b(flm_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    lib = capi.lib()
    for s in declared:
        assert hasattr(lib, s), s


def test_plan_shards_7b():
    d = capi.desc_from_config(synth.make_config("7B", ff.QT_INT8))
    for world in (1, 2, 4, 8):
        plans = [capi.plan_shards(d, r, world) for r in range(world)]
        assert sum(p.head_count for p in plans) == 32 and sum(p.hidden_count for p in plans) == 11008
        assert sum(p.dim_count for p in plans) == 4096 and sum(p.vocab_count for p in plans) == 32000
        for a, b in zip(plans, plans[1:]):        # contiguous, equal slices (all-gather layout == tensor layout)
            assert a.head_begin + a.head_count == b.head_begin and a.hidden_begin + a.hidden_count == b.hidden_begin
            assert a.dim_begin + a.dim_count == b.dim_begin and a.vocab_begin + a.vocab_count == b.vocab_begin
            assert (a.head_count, a.hidden_count, a.dim_count) == (b.head_count, b.hidden_count, b.dim_count)


def test_plan_shards_errors():
    d = capi.desc_from_config(synth.make_config("tiny", ff.QT_INT8))
    with pytest.raises(capi.FlmError):
        capi.plan_shards(d, 0, 8)        # 4 heads cannot feed 8 ranks
    with pytest.raises(capi.FlmError):
        capi.plan_shards(d, 2, 2)


def test_null_arguments_are_rejected_without_a_gpu():
    lib = capi.lib()
    assert lib.flm_forward(None, None, 1, 0, None) != 0
    assert lib.flm_op_quantize(2, None, None, None, C.c_size_t(64), 64) != 0
    assert lib.flm_ctx_create(None, 0, 0, 1, None, None) != 0
