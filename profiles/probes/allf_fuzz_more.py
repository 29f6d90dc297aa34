"""120 more seeds of tests/test_gpu_coloured.py::test_feature_major_fuzz (third session of r5: seeds 12..131 all pass)."""
import sys
sys.path.insert(0, ".")
import tests.test_gpu_coloured as t
bad = []
for seed in range(12, 132):
    try:
        t.test_feature_major_fuzz(seed)
    except AssertionError as ex:
        bad.append((seed, str(ex)[:200]))
        print("FAIL", seed, str(ex)[:200], flush=True)
print("seeds 12..131:", "all pass" if not bad else bad)
