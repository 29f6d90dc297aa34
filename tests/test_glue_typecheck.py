"""examples/FM_glue.cpp -- the Rcpp bodies of FM / FMPredict / FMTrack (/root/reference/src/FM.cpp:7,177,218; registered as in
src/RcppExports.cpp:10-52) written against include/fmx.h -- has no compiler in this image: R and Rcpp are absent.  This test type-checks it
(`g++ -fsyntax-only`) against tests/rcpp_shim/Rcpp.h, a DECLARATION-ONLY stand-in for the few Rcpp names it uses.  The shim defines nothing
and pins nothing; what the test catches is drift between the glue and the C ABI: a renamed entry point, a changed argument list, a dropped
constant.  A mutated copy of the glue must FAIL the same check (so a shim that swallowed everything would be noticed)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "examples", "FM_glue.cpp")


def _check(path):
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Werror=return-type", "-I", os.path.join(ROOT, "tests", "rcpp_shim"), "-I", os.path.join(ROOT, "include"), path]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_the_rcpp_glue_type_checks_against_fmx_h():
    r = _check(GLUE)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("old,new", [
    ("fmx_train(H.e, H.m, max_iter, nullptr)", "fmx_train(H.e, H.m, max_iter)"),                       # an argument dropped
    ("fmx_predict(H.e, H.m, out.begin(), link)", "fmx_predict(H.e, out.begin(), H.m, link)"),           # arguments swapped
    ("fmx_matrix_normalize(H.m, mean.begin(), sd.begin())", "fmx_matrix_normalise(H.m, mean.begin(), sd.begin())"),   # an entry point renamed
    ("c.solver = FMX_SOLVER_TDAP", "c.solver = FMX_SOLVER_TDAP2"),                                      # a constant that does not exist
    ("c.random_step = (int)solver[\"random_step\"]; }\n  else if (s == \"FTRL\")", "c.random_stride = 1; }\n  else if (s == \"FTRL\")"),  # a config field that does not exist
])
def test_a_drifted_glue_fails_the_type_check(tmp_path, old, new):
    src = open(GLUE).read()
    assert old in src
    bad = tmp_path / "FM_glue_mutated.cpp"
    bad.write_text(src.replace(old, new, 1))
    r = _check(str(bad))
    assert r.returncode != 0
