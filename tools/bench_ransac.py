"""RANSAC stage times at config 3 (1024 pairs x 200 iterations x 2000 correspondences)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orb_slam_amd import orbx
W, H = 640, 480
P, n, it = 1024, 2000, 200
rng = np.random.default_rng(3)
ref = np.stack([rng.uniform(0, W, (P, n)), rng.uniform(0, H, (P, n))], 2).astype(np.float32)
Hm = np.array([[1.01, 0.01, 3.0], [-0.01, 0.99, -2.0], [1e-5, -1e-5, 1.0]])
q = ref.astype(np.float64) @ Hm[:2, :2].T + Hm[:2, 2]
w = ref.astype(np.float64) @ Hm[2, :2] + 1.0
pts = (q / w[..., None] + rng.normal(0, 1.0, (P, n, 2))).astype(np.float32)
out = rng.uniform(size=(P, n)) < 0.3
pts[out] = np.stack([rng.uniform(0, W, out.sum()), rng.uniform(0, H, out.sum())], 1).astype(np.float32)
sets = np.stack([np.stack([rng.choice(n, 8, replace=False) for _ in range(it)]) for _ in range(64)])
sets = np.tile(sets, (P // 64 + 1, 1, 1))[:P].astype(np.int32)
dev = "cuda"
d_pts = torch.from_numpy(pts).to(dev); d_ref = torch.from_numpy(ref).to(dev); d_sets = torch.from_numpy(sets).to(dev)
d_F = torch.empty((P, 9), dtype=torch.float32, device=dev); d_H = torch.empty_like(d_F)
d_if = torch.empty((P, n), dtype=torch.uint8, device=dev); d_ih = torch.empty_like(d_if)
d_sc = torch.empty((P, 2), dtype=torch.float32, device=dev); d_b = torch.empty((P, 2), dtype=torch.int32, device=dev)
rs = orbx.Ransac()
if os.environ.get("RANSAC_NO_FORK"):
    rs.set_fork(False)   # every kernel alone on the chip
def step():
    rs.fh_batch(d_pts.data_ptr(), d_ref.data_ptr(), n, d_sets.data_ptr(), it, P, 1.0, d_F.data_ptr(), d_H.data_ptr(),
                d_if.data_ptr(), d_ih.data_ptr(), d_sc.data_ptr(), d_b.data_ptr())
torch.cuda.synchronize(); step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(json.dumps({"ms_per_batch": dt * 1e3, "hyp_per_s": P * it * 2 / dt, "stage_ms": rs.last_timings(),
                  "checksum": [float(d_F.double().nan_to_num().sum()), float(d_H.double().nan_to_num().sum()), int(d_if.sum()), int(d_ih.sum())]}))
