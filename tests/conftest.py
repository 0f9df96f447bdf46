import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Golden vectors generated from the reference (tests/golden/make_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "golden.npz")
    with np.load(path, allow_pickle=False) as z:
        out = {k: z[k] for k in z.files}
    # G10 (examples/tutorial.ipynb posterior) lives in its own file, tests/golden/make_golden_tutorial.py
    with np.load(os.path.join(ROOT, "tests", "golden", "golden_tutorial.npz"), allow_pickle=False) as z:
        out.update({k: z[k] for k in z.files})
    # G11-G17 (second round): tests/golden/make_golden_r2.py
    with np.load(os.path.join(ROOT, "tests", "golden", "golden_r2.npz"), allow_pickle=False) as z:
        out.update({k: z[k] for k in z.files})
    # round 3: the generating numpy's 10**logLam per row of G8, G9, G11-G14 (tests/golden/make_golden_r3.py)
    with np.load(os.path.join(ROOT, "tests", "golden", "golden_r3.npz"), allow_pickle=False) as z:
        out.update({k: z[k] for k in z.files if k.endswith("_sc2")})
    return out


@pytest.fixture(scope="session")
def golden_meta():
    """Non-array fixtures of make_golden_r2.py (G16: identifier strings of misc.gen_identifier)."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden_r2_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
