import sys, time, numpy as np
sys.path.insert(0, '.')
from orb_slam_amd import orbx
d = np.load('tests/golden/test_images.npz')
img = d['bgr1']
ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480)
for cache in (False, True):
    ex.set_image_cache(cache)
    for _ in range(5):
        k = ex.detect(img); k2, dsc = ex.compute(img, k)
    t0 = time.perf_counter()
    N = 300
    for _ in range(N):
        k = ex.detect(img); k2, dsc = ex.compute(img, k)
    dt = (time.perf_counter() - t0) / N
    print('image cache %d: single-frame detect+compute %.3f ms -> %.0f frames/s, kps %d' % (cache, dt * 1e3, 1 / dt, len(k2)))
