"""The boundary binds: oracle/gpu_backend.h -- the reference-side binding of include/flm_gpu.h that INTEGRATION.md describes --
is compiled against the REFERENCE's own headers and objects and linked with libflm_gpu.so (oracle/Makefile, target `ref`, build
container).  Here the resulting program is run: without a GPU it must load and report that (exit 3); on the GPU box it loads a
.flm with the reference's loader, runs the reference's forward and the binding's forward and compares the logits bit for bit."""
import os
import subprocess

import pytest

import __graft_entry__ as graft
from fast_llama_amd import flmfile as ff, synth

CHECK = os.path.join(graft.ROOT, "oracle", "_ref", "gpu_backend_check")


def _run(tmp_path, shape, qt):
    cfg = synth.make_config(shape, qt)
    path = str(tmp_path / "m.flm")
    synth.write_synthetic_flm(path, cfg, seed=5)
    r = subprocess.run([CHECK, path, "4"], capture_output=True, text=True, timeout=600)
    out = "\n".join(l for l in r.stdout.splitlines() if not l.startswith("DEBUG"))
    return r.returncode, out, r.stderr


@pytest.mark.skipif(not os.path.exists(CHECK), reason="oracle/_ref is built only where /root/reference exists")
def test_binding_compiles_links_and_loads(tmp_path):
    rc, out, err = _run(tmp_path, "tiny", ff.QT_INT8)
    assert rc in (0, 3), (rc, out, err)
    assert ("bit-identical" in out) if rc == 0 else ("binding compiled, linked and loaded" in out)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,qt", [("tiny", ff.QT_INT8), ("small", ff.QT_INT16)])
def test_binding_forward_equals_reference_forward(gpu, tmp_path, shape, qt):
    if not os.path.exists(CHECK):
        pytest.fail("oracle/_ref/gpu_backend_check is missing: build() makes it in the build container and it travels to the GPU box")
    rc, out, err = _run(tmp_path, shape, qt)
    assert rc == 0 and "bit-identical" in out, (rc, out, err)
