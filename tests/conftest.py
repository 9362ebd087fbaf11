import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def rrects(n, rng, extent=1024.0, lo=8.0, hi=256.0, jitter=0.5):
    """Synthetic rotated rectangles: centre U(0,extent)^2, long side
    log-U(lo,hi), aspect U(1,6), angle U(0,pi), N(0,jitter) corner noise."""
    c = rng.uniform(0, extent, (n, 2))
    long_side = np.exp(rng.uniform(np.log(lo), np.log(hi), n))
    w, h = long_side, long_side / rng.uniform(1, 6, n)
    a = rng.uniform(0, np.pi, n)
    ca, sa = np.cos(a), np.sin(a)
    ux = np.stack([w / 2 * ca - h / 2 * sa, -w / 2 * ca - h / 2 * sa,
                   -w / 2 * ca + h / 2 * sa, w / 2 * ca + h / 2 * sa], 1)
    uy = np.stack([w / 2 * sa + h / 2 * ca, -w / 2 * sa + h / 2 * ca,
                   -w / 2 * sa - h / 2 * ca, w / 2 * sa - h / 2 * ca], 1)
    p = np.empty((n, 8))
    p[:, 0::2] = c[:, :1] + ux
    p[:, 1::2] = c[:, 1:] + uy
    return (p + rng.normal(0, jitter, p.shape)).astype(np.float32)
