import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
BASELINE_IMAGES = ["img", "img2", "img3", "img4", "img5", "img6"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


def load_golden(name):
    """-> (desc, coef, qtabs, rgb) of one baseline image fixture."""
    from oracle.pyoracle import make_desc
    z = np.load(os.path.join(GOLD, name + ".npz"))
    d = z["desc"]
    return make_desc(d[0], d[1], d[2], d[3], d[4:7]), z["coef"], z["qtabs"], z["rgb"]


def load_kat():
    """-> {case: (desc, coef, qtabs, rgb)} of the known-answer block vectors."""
    from oracle.pyoracle import make_desc
    z = np.load(os.path.join(GOLD, "kat_blocks.npz"))
    names = sorted({k.split("__")[0] for k in z.files})
    out = {}
    for n in names:
        d = z[n + "__desc"]
        out[n] = (make_desc(d[0], d[1], d[2], d[3], d[4:7]), z[n + "__coef"], z[n + "__qtabs"], z[n + "__rgb"])
    return out
