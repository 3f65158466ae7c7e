"""Back-end matchers (BoW, epipolar, fuse search, best descriptor) on 1000-feature key frames: time per call."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from orb_slam_amd import orbx
import tests.test_gpu_parity as T

d = np.load('tests/golden/test_images.npz')
ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480)
rng = np.random.default_rng(1)
k0, d0 = ex.compute(d['bgr1'], ex.detect(d['bgr1']))
n = len(k0)
flips = (rng.uniform(size=(n, 256)) < rng.uniform(0.0, 0.25, (n, 1))).astype(np.uint8)
d1 = d0 ^ np.packbits(flips, axis=1)
k1 = k0.copy()
k1["y"] += rng.normal(0, 1.5, n).astype(np.float32)
_, sf, _, sig, inv_sig = ex.scale_tables()
nodes, ref_nodes = T._feature_vector(d0, rng, 100), T._feature_vector(d1, rng, 100)
has = np.zeros(n, np.uint8)
F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
m = orbx.Matcher()
def timed(label, fn, reps=20):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    dt = (time.perf_counter() - t0) / reps
    print('%s: %.3f ms per call' % (label, dt * 1e3))
    return r
timed('match_bow (1000 x 1000, 100 nodes)', lambda: m.match_bow(k0, d0, k1, d1, 1 - has, nodes, ref_nodes))
timed('match_epipolar', lambda: m.match_epipolar(k0, d0, k1, d1, has, has, F, sig, nodes, ref_nodes))
if hasattr(m, 'match_bow_batch'):
    P = 32
    timed('match_bow_batch, %d pairs' % P, lambda: m.match_bow_batch([(k0, d0, k1, d1, 1 - has, nodes, ref_nodes)] * P), 5)
    timed('match_epipolar_batch, %d pairs' % P, lambda: m.match_epipolar_batch([(k0, d0, k1, d1, has, has, F, nodes, ref_nodes)] * P, sig), 5)
