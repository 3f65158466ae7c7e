import sys, time, numpy as np
sys.path.insert(0, '.')  # run from the repository root
from orb_slam_amd import orbx
import torch
w, h, B = 3840, 2160, 8
rng = np.random.default_rng(4)
frames = []
for f in range(B):
    img = np.full((h, w), 100, np.float32)
    for _ in range(6000):
        x0, y0 = rng.integers(0, w), rng.integers(0, h)
        ww, hh = rng.integers(3, 400), rng.integers(3, 300)
        img[y0:y0 + hh, x0:x0 + ww] = rng.integers(0, 256)
    img += rng.normal(0, 2.5, img.shape)
    frames.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
dev = torch.from_numpy(np.stack(frames)).cuda()
torch.cuda.synchronize()
ex = orbx.Extractor(n_key_points=8000, level_pyramid=12, max_width=w, max_height=h, max_batch=B)
for sel in (True, False):
    ex.set_device_select(sel)
    for _ in range(2):
        ex.extract_batch(dev.data_ptr(), B, w * h, w, h, w, 1, 9000, to_host=False)
    t0 = time.perf_counter()
    N = 5
    for _ in range(N):
        _, counts, _ = ex.extract_batch(dev.data_ptr(), B, w * h, w, h, w, 1, 9000, to_host=False)
    dt = (time.perf_counter() - t0) / N
    print('device_select', sel, 'batch %d x 4K: %.2f ms -> %.1f frames/s' % (B, dt * 1e3, B / dt), 'kps', counts[:3], ex.last_timings())
