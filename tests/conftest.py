import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def test_images():
    d = np.load(os.path.join(GOLDEN, "test_images.npz"))
    return [d["bgr1"], d["bgr2"]]


@pytest.fixture(scope="session")
def utils_points():
    d = np.load(os.path.join(GOLDEN, "utils_points.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


def synth_image(seed, w, h, channels=1):
    """Random rectangles + discs + noise: enough corners at every pyramid level."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 110, np.float32)
    for _ in range(max(40, w * h // 1500)):
        x0, y0 = rng.integers(0, w), rng.integers(0, h)
        ww, hh = rng.integers(3, max(4, w // 6)), rng.integers(3, max(4, h // 6))
        img[y0:y0 + hh, x0:x0 + ww] = rng.integers(0, 256)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(max(10, w * h // 6000)):
        cx, cy, r = rng.integers(0, w), rng.integers(0, h), rng.integers(2, max(3, min(w, h) // 10))
        img[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = rng.integers(0, 256)
    img += rng.normal(0, 3, img.shape)
    g = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if channels == 1:
        return g
    out = np.stack([np.clip(g.astype(np.int32) + rng.integers(-20, 20, g.shape), 0, 255).astype(np.uint8)
                    for _ in range(3)], axis=2)
    return np.ascontiguousarray(out)
