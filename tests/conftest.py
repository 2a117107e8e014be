import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def split_golden(g, dtype=torch.float32):
    """-> (plain arrays as tensors, state_dict, grads)"""
    plain, sd, grads = {}, {}, {}
    for k, v in g.items():
        t = torch.from_numpy(np.asarray(v))
        if t.is_floating_point():
            t = t.to(dtype)
        if k.startswith("sd/"):
            sd[k[3:]] = t
        elif k.startswith("grad/"):
            grads[k[5:]] = t
        elif k.startswith("sd_after/"):
            plain[k] = t
        else:
            plain[k] = t
    return plain, sd, grads


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import asr_oracle
    return asr_oracle


def has_gpu():
    return torch.cuda.is_available()
