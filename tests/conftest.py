import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "spmv-research_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


MANIFEST = load_manifest()
CASES = sorted(MANIFEST["cases"].keys())


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return MANIFEST["cases"][name], {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
