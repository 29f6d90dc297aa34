"""Regenerates tests/golden/probit_tables.json: a SAMPLE of the reference's probit grids (util/RandomData.h: Phi on 2861
points; util/RandomData_.h: dnorm/(1-pnorm) on 40001 points), read as data from /root/reference -- every 13th / 97th
grid point plus both ends and the constants.  The oracle and the engine regenerate the full grids from their defining
formulas; this sample is what pins that regeneration wherever /root/reference is absent (tests/test_oracle_probit.py).
Run in the build container:  python tests/golden/make_probit_golden.py"""
import json
import os
import re

REF = "/root/reference/src/util"
HERE = os.path.dirname(os.path.abspath(__file__))


def array(text, name):
    i = text.index(name + " ")
    j = text.index("{", i)
    k = text.index("}", j)
    return [float(x) for x in text[j + 1:k].replace("\n", " ").split(",") if x.strip()]


def scalar(text, name):
    return float(re.search(r"static const double %s = ([^;]+);" % re.escape(name), text).group(1))


a = open(os.path.join(REF, "RandomData.h")).read()
b = open(os.path.join(REF, "RandomData_.h")).read()
pn_y, dp_y = array(a, "_Y_"), array(b, "__Y__")
pn_idx = sorted(set(list(range(0, len(pn_y), 13)) + [len(pn_y) - 1]))
dp_idx = sorted(set(list(range(0, len(dp_y), 97)) + [len(dp_y) - 1]))
out = {
    "source": "evanwang1990/FMwR src/util/RandomData.h, RandomData_.h (sampled grid points)",
    "pnorm": {"points": len(pn_y), "max": scalar(a, "_MAX_"), "hinv": scalar(a, "_HINV_"), "index": pn_idx, "y": [pn_y[i] for i in pn_idx]},
    "dpnorm": {"points": len(dp_y), "min": scalar(b, "_MIN_"), "max": scalar(b, "_MAX_"), "index": dp_idx, "y": [dp_y[i] for i in dp_idx]},
}
json.dump(out, open(os.path.join(HERE, "probit_tables.json"), "w"))
print(len(pn_idx), len(dp_idx))
