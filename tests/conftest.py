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


NMS_SET_KINDS = ("uniform", "dense", "skewed")


def nms_candidate_set(kind, m, rng, extent=1024.0):
    """SURVEY 8(d) synthetic candidate sets for the rotated NMS -> (boxes [m,8] f32, scores [m] f32, classes [m] i64).
      uniform  centres U(0, extent)^2, classes uniform over 15 (the DOTA-1.0 head)
      dense    70 % of the boxes packed into one 256 x 256 window (long suppression chains), classes uniform over 15
      skewed   the DOTA-1.5 histogram: 16 classes, 60 % of the boxes in {4, 5, 6} (small-vehicle / large-vehicle /
               ship; nms.py:77-79 then merges 5 into 4), the rest uniform over the other 13
    Quads: rotated rectangles, long side log-U(8,256), aspect U(1,6), angle U(0,pi), N(0,0.5) corner jitter;
    scores U(0.05,1)."""
    b = rrects(m, rng, extent=extent)
    s = rng.uniform(0.05, 1, m).astype(np.float32)
    if kind == "uniform":
        c = rng.integers(0, 15, m)
    elif kind == "dense":
        c = rng.integers(0, 15, m)
        nd = int(0.7 * m)
        win = rrects(nd, rng, extent=256.0)
        win += np.float32(extent / 2 - 128.0)
        idx = rng.permutation(m)[:nd]
        b[idx] = win
    elif kind == "skewed":
        hot = rng.random(m) < 0.6
        others = np.array([k for k in range(16) if k not in (4, 5, 6)])
        c = np.where(hot, rng.choice([4, 5, 6], m), rng.choice(others, m))
    else:
        raise ValueError(kind)
    return b.astype(np.float32), s, c.astype(np.int64)
