"""The boundary exercised from plain C (examples/kat_c.c): gcc, include/fmx.h, libfmx.so -- the call sequence FM() and
FMPredict() make around the learner seam -- reproducing the reference's own known answers."""
import os
import subprocess

import numpy as np
import pytest

from tests import kat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "kat_c")
    lib_dir = os.path.join(ROOT, "fmwr_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "kat_c.c"), "-L" + lib_dir, "-lfmx", "-Wl,-rpath," + lib_dir, "-lm", "-o", exe])
    return exe


def test_c_caller_compiles_against_the_header(tmp_path):
    """No GPU needed: the header is C99 and the library links from C; without a device the program fails loudly."""
    from fmwr_amd import build
    build.build()
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_caller_reproduces_the_reference_known_answers(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = dict(line.split() for line in r.stdout.strip().splitlines())
    assert int(out["examples"]) == 50
    assert abs(float(out["w0"]) - kat.SGD_W0) < 1e-13
    np.testing.assert_allclose([float(out["lin%d" % j]) for j in range(5)], kat.SGD_W, rtol=0, atol=1e-13)
    np.testing.assert_allclose([float(out["v0_%d" % j]) for j in range(5)], kat.SGD_V0, rtol=0, atol=1e-13)
    assert abs(float(out["ll"]) - kat.SGD_LL[-1]) < 5e-10
    prob = np.array([float(out["p%d" % i]) for i in range(6)])
    assert np.all((prob > 0) & (prob < 1))
    # the multi-GPU entry from C: n_gpus = 2 (both replicas on this device) equals one engine on the interleaved global batches
    assert int(out["g2_examples"]) == 18 and int(out["g1_examples"]) == 18
    keys = ["w0"] + ["lin%d" % j for j in range(5)] + ["v%d" % j for j in range(15)]
    a = np.array([float(out["g2_" + key]) for key in keys]); b = np.array([float(out["g1_" + key]) for key in keys])
    assert np.max(np.abs(a - b)) < 1e-12
    assert np.any(a[1:6] != 0.0)
