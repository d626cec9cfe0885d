"""The drop-in CLI (fast-llama_amd/bin/main) against transcripts of the reference CLI (tests/golden/cli_transcripts.npz,
produced by tests/golden/make_golden.py:g_cli from oracle/_ref/main): same prompt echo, same token list, the same
generated text, the same summary-line fields (timings aside)."""
import os
import re
import subprocess

import numpy as np
import pytest

import __graft_entry__ as graft
from fast_llama_amd import flmfile as ff, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MAIN = os.path.join(graft.PKG_DIR, "bin", "main")


def _cases():
    src = open(os.path.join(GOLD, "make_golden.py")).read()
    # only the CLI_CASES literal is needed; importing the generator would pull in the reference bindings
    m = re.search(r"CLI_CASES = \[.*?\n\]\n", src, re.S)
    ns = {"ff": ff}
    exec(m.group(0), ns)
    exec(re.search(r"GGUF_CASES = \[.*?\n\]\n", src, re.S).group(0), ns)
    return ns["CLI_CASES"], ns["GGUF_CASES"]


def _strip_timing(b: bytes) -> bytes:
    b = re.sub(rb"total_latancy:.*", b"total_latancy:<t>", b)
    return re.sub(rb"num_threads:\x1b\[33m *-?\d+\x1b\[0m", b"num_threads:<n>", b)


def _run(name, tmp_path, extra_args=()):
    case = next(c for c in _cases()[0] if c[0] == name)
    _, shape, qt, seed, extra = case
    cfg = synth.make_config(shape, qt)
    path = str(tmp_path / f"{name}.flm")
    synth.write_synthetic_flm(path, cfg, seed=seed)
    if not os.path.exists(MAIN):
        graft.build()
    r = subprocess.run([MAIN, "-c", path, "-j", "1", *extra, *extra_args], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")
    return r.stdout, bytes(np.load(os.path.join(GOLD, "cli_transcripts.npz"))[name])


@pytest.mark.parametrize("name", ["encode", "decode"])
def test_cli_encode_decode_modes_match_reference(name, tmp_path):
    got, want = _run(name, tmp_path)
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["greedy_int8", "sample_int16", "tiny128_int8"])
def test_cli_generation_matches_reference_transcript(gpu, name, tmp_path):
    got, want = _run(name, tmp_path)
    assert _strip_timing(got) == _strip_timing(want)


@pytest.mark.gpu
def test_cli_benchmark_mode_prints_only_summary(gpu, tmp_path):
    got, want = _run("greedy_int8", tmp_path, ["--mode", "bm", "--rounds", "2"])
    lines = got.split(b"\n")
    assert not any(l.startswith(b"output:") for l in lines)
    assert _strip_timing(lines[-2]) == _strip_timing(want.split(b"\n")[-2])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gguf_f32_int8", "gguf_f32_int16"])
def test_cli_on_gguf_matches_reference_transcript(gpu, name, tmp_path):
    """F32 gguf (gguf_loader.cpp:205-489): fp32 masters quantized with -q on the device, no q/k permutation; transcripts of the reference
    binary on the same file.  The same tensors in an fp32-master .flm give the same output again."""
    from fast_llama_amd import gguffile
    _, shape, seed, f16, extra = next(c for c in _cases()[1] if c[0] == name)
    cfg = synth.make_config(shape, ff.QT_NONE)
    tensors = synth.make_tensors(cfg, seed=seed, fp32_master=True)
    path = str(tmp_path / f"{name}.gguf")
    gguffile.write_gguf(path, cfg, synth.make_tokenizer(cfg.vocab_size), tensors, f16=f16)
    r = subprocess.run([MAIN, "-c", path, "-j", "1", *extra], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")
    want = bytes(np.load(os.path.join(GOLD, "cli_transcripts.npz"))[name])
    assert _strip_timing(r.stdout) == _strip_timing(want)
    flm = str(tmp_path / f"{name}.flm")
    synth.write_synthetic_flm(flm, cfg, tensors=tensors, fp32_master=True)
    r2 = subprocess.run([MAIN, "-c", flm, "-j", "1", *[a for a in extra if a not in ("-f", "gguf")]], capture_output=True, timeout=600)
    assert r2.returncode == 0, r2.stderr.decode(errors="replace")
    strip_name = lambda b: re.sub(rb"model:[^\t]*", b"model:<m>", _strip_timing(b))
    assert strip_name(r2.stdout) == strip_name(r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("name,devices", [("greedy_int8", "0,0"), ("sample_int16", "0,0,0,0")])
def test_cli_devices_shards_one_sequence_over_ranks(gpu, name, devices, tmp_path):
    """--devices a,b,...: the reference's parallel width (`-j`, main.cpp:30,78; split_rows transformer.cpp:264-287) as tensor-parallel ranks, one host thread
    per device, peers mapped inside the process.  On the 1-GPU box the ranks share device 0; the transcript must be the reference's (the sharded run is
    bit-identical to the single-GPU one), greedy and sampled."""
    got, want = _run(name, tmp_path, ["--devices", devices])
    assert _strip_timing(got) == _strip_timing(want)
