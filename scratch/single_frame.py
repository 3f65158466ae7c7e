import sys, time, numpy as np
sys.path.insert(0, '.')
from orb_slam_amd import orbx
d = np.load('tests/golden/test_images.npz')
img = d['bgr1']
ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480)
for _ in range(5):
    k = ex.detect(img); k2, dsc = ex.compute(img, k)
t0 = time.perf_counter()
N = 200
for _ in range(N):
    k = ex.detect(img); k2, dsc = ex.compute(img, k)
dt = (time.perf_counter() - t0) / N
print('single-frame host-boundary detect+compute: %.3f ms -> %.0f frames/s, kps %d' % (dt * 1e3, 1 / dt, len(k2)))
import torch
B = 256
frames = torch.from_numpy(np.stack([img] * B))
pinned = frames.pin_memory()
dev = torch.empty_like(frames, device='cuda')
exb = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=B)
torch.cuda.synchronize()
for _ in range(2):
    dev.copy_(pinned, non_blocking=True); torch.cuda.synchronize()
    exb.extract_batch(dev.data_ptr(), B, 640 * 480 * 3, 640, 480, 640 * 3, 3, 1280, to_host=False)
t0 = time.perf_counter()
for _ in range(5):
    dev.copy_(pinned, non_blocking=True); torch.cuda.synchronize()
    exb.extract_batch(dev.data_ptr(), B, 640 * 480 * 3, 640, 480, 640 * 3, 3, 1280, to_host=False)
dt = (time.perf_counter() - t0) / 5
print('batch of %d incl. pinned H2D upload (not overlapped): %.2f ms -> %.0f frames/s' % (B, dt * 1e3, B / dt))
