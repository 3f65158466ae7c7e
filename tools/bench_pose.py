"""Pose recovery (findRtWithFundamental + triangulatePoints) for 1024 pairs x 1400 inlier correspondences:
the batched device entry points (host pointers in and out) beside the CPU oracle on one pair."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from orb_slam_amd import orbx
from oracle import pyoracle as O

rng = np.random.default_rng(3)
K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]], np.float32)
Kd = K.astype(np.float64)
P, n = 1024, 1400
a = 0.05
R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
t = np.array([0.3, 0.02, 0.05])
X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2, 6, n)], 1)
x1 = (Kd @ X.T).T
x2 = (Kd @ (R @ X.T + t[:, None])).T
ref = np.ascontiguousarray(x1[:, :2] / x1[:, 2:])
pts = np.ascontiguousarray(x2[:, :2] / x2[:, 2:] + rng.normal(0, 0.5, (n, 2)))
tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
Ki = np.linalg.inv(Kd)
F = Ki.T @ tx @ R @ Ki
F = (F / np.linalg.norm(F)).astype(np.float32)
Fs = np.stack([F] * P)
pts_all = np.ascontiguousarray(np.tile(pts, (P, 1)))
ref_all = np.ascontiguousarray(np.tile(ref, (P, 1)))
start = (np.arange(P + 1) * n).astype(np.int32)
rs = orbx.Ransac()
for _ in range(2):
    Rr, tr, good = rs.recover_pose_batch(Fs, K, pts_all, ref_all, start)
    xyz = rs.triangulate_batch(K, Rr, tr, pts_all, ref_all, start)
t0 = time.perf_counter()
N = 5
for _ in range(N):
    Rr, tr, good = rs.recover_pose_batch(Fs, K, pts_all, ref_all, start)
t1 = time.perf_counter()
for _ in range(N):
    xyz = rs.triangulate_batch(K, Rr, tr, pts_all, ref_all, start)
t2 = time.perf_counter()
print('recover pose: %.2f ms per %d pairs (%.0f pairs/s, %.2e 4x4 SVDs/s); triangulate: %.2f ms (%.2e points/s); good %d/%d' %
      ((t1 - t0) / N * 1e3, P, P * N / (t1 - t0), 4.0 * P * n * N / (t1 - t0), (t2 - t1) / N * 1e3,
       P * n * N / (t2 - t1), good[0], n))
t0 = time.perf_counter()
for _ in range(3):
    oR, ot, og = O.recover_pose_fundamental(F, K, pts, ref)
    o3 = O.triangulate_initial(K, oR, ot, pts, ref)
dt = (time.perf_counter() - t0) / 3
print('CPU oracle, one pair: %.2f ms -> %.0f pairs/s; equal to device: %s' %
      (dt * 1e3, 1 / dt, np.array_equal(oR, Rr[0]) and np.array_equal(o3, xyz[:n])))

# the homography branch (opt-in in the facade): decomposition + visibility filter for the same point sets
Hm = np.array([[1.01, 0.01, 3.0], [-0.01, 0.99, -2.0], [1e-5, -1e-5, 1.0]], np.float32)
Hs = np.stack([Hm] * P)
for _ in range(2):
    hR, ht, hidx, hall, hn = rs.recover_pose_homography_batch(Hs, K, pts_all, ref_all, start)
t0 = time.perf_counter()
for _ in range(N):
    hR, ht, hidx, hall, hn = rs.recover_pose_homography_batch(Hs, K, pts_all, ref_all, start)
dt = (time.perf_counter() - t0) / N
oR4, ot4, on4 = O.decompose_homography(Hm, K)
print('homography pose: %.2f ms per %d pairs (%.0f pairs/s); motions equal to the oracle: %s' %
      (dt * 1e3, P, P / dt, np.array_equal(hall[0, :, :9].reshape(4, 3, 3), oR4)))

# the public n-point solvers: 1024 systems of 79 correspondences (the size of the reference's own test)
m = 79
sp = rng.uniform(0, 640, (P, m, 2)).astype(np.float32)
sr = (sp + rng.normal(0, 2, (P, m, 2))).astype(np.float32)
for _ in range(2):
    Fn = rs.fundamental_npt(sp, sr)
    Hn = rs.homography_dlt(sp, sr)
t0 = time.perf_counter()
for _ in range(N):
    Fn = rs.fundamental_npt(sp, sr)
t1 = time.perf_counter()
for _ in range(N):
    Hn = rs.homography_dlt(sp, sr)
t2 = time.perf_counter()
print('n-point solvers, %d systems x %d points: F %.2f ms, H %.2f ms per call (host pointers in and out)' %
      (P, m, (t1 - t0) / N * 1e3, (t2 - t1) / N * 1e3))
