"""orbx_ransac_fh_batch_static (the entry point that allocates nothing) against orbx_ransac_fh_batch: bit for bit,
on two calls in a row, the second with a smaller input, through the C ABI."""
import numpy as np
import pytest

from orb_slam_amd import orbx

pytestmark = pytest.mark.gpu


def _two_view(rng, n, outliers, noise=0.5):
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2, 6, n)], 1)
    K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])
    a = 0.05
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    t = np.array([0.3, 0.02, 0.05])
    x1 = (K @ X.T).T
    x2 = (K @ (R @ X.T + t[:, None])).T
    ref = x1[:, :2] / x1[:, 2:]
    pts = x2[:, :2] / x2[:, 2:] + rng.normal(0, noise, (n, 2))
    idx = rng.choice(n, outliers, replace=False)
    pts[idx] = np.stack([rng.uniform(0, 640, outliers), rng.uniform(0, 480, outliers)], 1)
    return np.ascontiguousarray(pts, np.float32), np.ascontiguousarray(ref, np.float32)


def _run(torch, rs, fn, pts, ref, sets, sigma):
    P, n, it = pts.shape[0], pts.shape[1], sets.shape[1]
    d_pts, d_ref = torch.from_numpy(pts).cuda(), torch.from_numpy(ref).cuda()
    d_sets = torch.from_numpy(sets.astype(np.int64)).to(torch.int32).cuda()
    # outputs start from a pattern, so rows a call must leave untouched can be told from rows it wrote
    d_F = torch.full((P, 9), -7.0, dtype=torch.float32, device="cuda")
    d_H = torch.full((P, 9), -7.0, dtype=torch.float32, device="cuda")
    d_if = torch.full((P, n), 9, dtype=torch.uint8, device="cuda")
    d_ih = torch.full((P, n), 9, dtype=torch.uint8, device="cuda")
    d_sc = torch.full((P, 2), -7.0, dtype=torch.float32, device="cuda")
    d_best = torch.full((P, 2), 77, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    fn(d_pts.data_ptr(), d_ref.data_ptr(), n, d_sets.data_ptr(), it, P, sigma, d_F.data_ptr(), d_H.data_ptr(),
       d_if.data_ptr(), d_ih.data_ptr(), d_sc.data_ptr(), d_best.data_ptr())
    torch.cuda.synchronize()
    free_after = torch.cuda.mem_get_info()[0]
    return [t.cpu().numpy() for t in (d_F, d_H, d_if, d_ih, d_sc, d_best)], free_before - free_after


def _case(rng, P, n, it, bad_pair=None):
    data = [_two_view(rng, n, n // 4) for _ in range(P)]
    pts = np.ascontiguousarray(np.stack([d[0] for d in data]))
    ref = np.ascontiguousarray(np.stack([d[1] for d in data]))
    sets = np.stack([orbx.ransac_make_sets(n, it, 900 + 13 * i) for i in range(P)])
    if bad_pair is not None:
        sets[bad_pair, it // 2, 3] = n + 5  # an index that names no correspondence
    return pts, ref, sets


def test_static_workspace_equals_allocating_entry_point():
    import torch
    rng = np.random.default_rng(77)
    rs = orbx.Ransac()
    # first call: large (31 pairs x 1000 iterations = 31000 hypotheses, 62000 correspondences), with one invalid set;
    # second call: smaller in every dimension, all sets valid - whatever the first call left in the workspace
    # (its hypotheses, its scores, the raised flag of pair 4) must not leak into it; third: few iterations, which
    # takes the pair-by-pair scoring path
    for P, n, it, bad in ((31, 2000, 1000, 4), (7, 300, 200, None), (3, 64, 20, None)):
        pts, ref, sets = _case(rng, P, n, it, bad)
        want, _ = _run(torch, rs, rs.fh_batch, pts, ref, sets, 1.0)
        got, taken = _run(torch, rs, rs.fh_batch_static, pts, ref, sets, 1.0)
        assert taken == 0, f"the call took {taken} bytes of device memory"   # hipMemGetInfo: free bytes unchanged
        for name, a, b in zip(("F", "H", "inliers_f", "inliers_h", "scores", "best"), got, want):
            assert a.tobytes() == b.tobytes(), (P, n, it, name)
        best = got[5]
        if bad is not None:
            assert (best[bad] == -2).all() and (np.delete(best, bad, 0) >= 0).all()
            assert (got[0][bad] == -7.0).all() and not got[2][bad].any()
        else:
            assert (best >= 0).all()
    # beyond the fixed workspace: refused with a clear error, nothing runs
    for P, n, it in ((1025, 8, 1), (33, 100, 1000), (70, 2000, 10)):
        with pytest.raises(orbx.OrbxError) as e:
            rs.fh_batch_static(1, 1, n, 1, it, P, 1.0, 1, 1, 1, 1, 1, 1)
        assert e.value.status == orbx.CAPACITY and "exceed the fixed workspace" in str(e.value)
    # two handles on streams of their own share the GPU's one array: the library makes the second call wait for the
    # first on the device, so both come back right (alone they would overwrite each other's hypotheses)
    rs2 = orbx.Ransac()
    a, b = _case(rng, 24, 1200, 900, None), _case(rng, 20, 900, 700, None)
    want_a, _ = _run(torch, rs, rs.fh_batch, *a, 1.0)
    want_b, _ = _run(torch, rs2, rs2.fh_batch, *b, 1.0)
    for _ in range(3):
        bufs = []
        for r, (pts, ref, sets) in ((rs, a), (rs2, b)):
            P, n, it = pts.shape[0], pts.shape[1], sets.shape[1]
            d = [torch.from_numpy(pts).cuda(), torch.from_numpy(ref).cuda(),
                 torch.from_numpy(sets.astype(np.int64)).to(torch.int32).cuda(),
                 torch.zeros((P, 9), dtype=torch.float32, device="cuda"), torch.zeros((P, 9), dtype=torch.float32, device="cuda"),
                 torch.zeros((P, n), dtype=torch.uint8, device="cuda"), torch.zeros((P, n), dtype=torch.uint8, device="cuda"),
                 torch.zeros((P, 2), dtype=torch.float32, device="cuda"), torch.zeros((P, 2), dtype=torch.int32, device="cuda")]
            bufs.append((r, n, it, P, d))
        torch.cuda.synchronize()
        for r, n, it, P, d in bufs:   # queued back to back, no host wait in between
            r.fh_batch_static(d[0].data_ptr(), d[1].data_ptr(), n, d[2].data_ptr(), it, P, 1.0, *[t.data_ptr() for t in d[3:]])
        torch.cuda.synchronize()
        for (r, n, it, P, d), want in zip(bufs, (want_a, want_b)):
            for name, t, w in zip(("F", "H", "inliers_f", "inliers_h", "scores", "best"), d[3:], want):
                assert t.cpu().numpy().tobytes() == w.tobytes(), name
    rs2.close()
    rs.close()
