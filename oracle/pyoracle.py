"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the shipped package never does.  The shared library is built by `make -C oracle`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
MATCH_DTYPE = np.dtype([("query", "<i4"), ("train", "<i4"), ("img", "<i4"), ("distance", "<f4")])


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("edge_threshold", C.c_int32), ("first_level", C.c_int32), ("wta_k", C.c_int32),
                ("patch_size", C.c_int32), ("fast_threshold", C.c_int32), ("gray_variant", C.c_int32),
                ("resize_variant", C.c_int32), ("retain_variant", C.c_int32), ("blur_variant", C.c_int32),
                ("round_variant", C.c_int32), ("blur_sigma", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("k1", C.c_double), ("k2", C.c_double), ("p1", C.c_double), ("p2", C.c_double),
                ("k3", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


class GridParams(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("grid_cols", C.c_int32), ("grid_rows", C.c_int32), ("grid_size_x", C.c_int32),
                ("grid_size_y", C.c_int32)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_score_fundamental.restype = C.c_double
        _lib.orc_score_homography.restype = C.c_double
        _lib.orc_scale_tables.restype = C.c_float
        _lib.orc_fast_atan2.restype = C.c_float
        _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def orb_params(nfeatures=1000, scale_factor=1.2, nlevels=8, edge_threshold=31, fast_threshold=20,
               gray_variant=0, resize_variant=0, retain_variant=0, blur_variant=0, round_variant=0, blur_sigma=0.0):
    return OrbParams(nfeatures, scale_factor, nlevels, edge_threshold, 0, 2, 31, fast_threshold,
                     gray_variant, resize_variant, retain_variant, blur_variant, round_variant, blur_sigma)


def _img(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    return img, w, h, w * ch, ch


def layout(p, w, h, nlevels=None):
    nlevels = nlevels or p.nlevels
    aw, ah = C.c_int(), C.c_int()
    rects = np.zeros((nlevels, 4), np.int32)
    scales = np.zeros(nlevels, np.float32)
    lib().orc_orb_layout(C.byref(p), w, h, nlevels, C.byref(aw), C.byref(ah), _p(rects), _p(scales))
    return aw.value, ah.value, rects, scales


def gray(p, img):
    img, w, h, stride, ch = _img(img)
    out = np.zeros((h, w), np.uint8)
    lib().orc_gray(C.byref(p), _p(img), w, h, stride, ch, _p(out))
    return out


def pyramid(p, img, nlevels=None):
    img, w, h, stride, ch = _img(img)
    nlevels = nlevels or p.nlevels
    aw, ah, rects, scales = layout(p, w, h, nlevels)
    atlas = np.zeros((ah, aw), np.uint8)
    lib().orc_orb_pyramid(C.byref(p), _p(img), w, h, stride, ch, nlevels, _p(atlas))
    return atlas, rects, scales


def blur(p, atlas, w, h, nlevels=None):
    out = np.ascontiguousarray(atlas).copy()
    lib().orc_orb_blur(C.byref(p), w, h, nlevels or p.nlevels, _p(out))
    return out


def fast_scores(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_fast_scores(_p(img), w, h, w, threshold, _p(out))
    return out


def fast_detect(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = w * h // 4 + 16
    out = np.zeros(cap, KP_DTYPE)
    n = lib().orc_fast_detect(_p(img), w, h, w, threshold, _p(out), cap)
    return out[:n].copy()


def detect(p, img, cap=200000):
    img, w, h, stride, ch = _img(img)
    out = np.zeros(cap, KP_DTYPE)
    n = lib().orc_orb_detect(C.byref(p), _p(img), w, h, stride, ch, _p(out), cap)
    assert n <= cap
    return out[:n].copy()


def detect_distributed(p, img, low_threshold=7, cell_size=30, cap=200000):
    """The opt-in distribution mode (self-oracle): per-cell adaptive FAST threshold + quad-tree selection."""
    img, w, h, stride, ch = _img(img)
    out = np.zeros(cap, KP_DTYPE)
    lib().orc_orb_detect_distributed.restype = C.c_int
    n = lib().orc_orb_detect_distributed(C.byref(p), _p(img), w, h, stride, ch, low_threshold, cell_size, _p(out), cap)
    assert n <= cap
    return out[:n].copy()


def quadtree_select(xs, ys, scores, w, h, lo, quota, high_threshold, cell_size):
    xs = np.ascontiguousarray(xs, np.int32)
    ys = np.ascontiguousarray(ys, np.int32)
    scores = np.ascontiguousarray(scores, np.int32)
    kept = np.zeros(quota + 8, np.int32)
    lib().orc_quadtree_select.restype = C.c_int
    m = lib().orc_quadtree_select(_p(xs), _p(ys), _p(scores), len(xs), w, h, lo, quota, high_threshold, cell_size,
                                  _p(kept))
    return kept[:m].copy()


def detect_stage1(p, img, cap=400000):
    img, w, h, stride, ch = _img(img)
    out = np.zeros(cap, KP_DTYPE)
    counts = np.zeros(p.nlevels, np.int32)
    n = lib().orc_orb_detect_stage1(C.byref(p), _p(img), w, h, stride, ch, _p(out), cap, _p(counts))
    assert n <= cap
    return out[:n].copy(), counts


def compute(p, img, kps):
    img, w, h, stride, ch = _img(img)
    k = np.ascontiguousarray(kps, dtype=KP_DTYPE).copy()
    desc = np.zeros((max(len(k), 1), 32), np.uint8)
    n = lib().orc_orb_compute(C.byref(p), _p(img), w, h, stride, ch, _p(k), len(k), _p(desc))
    return k[:n].copy(), desc[:n].copy()


def scale_tables(scale_factor, nlevels):
    arrs = [np.zeros(nlevels, np.float32) for _ in range(4)]
    lsf = lib().orc_scale_tables(C.c_float(scale_factor), nlevels, *[_p(a) for a in arrs])
    return (lsf, *arrs)


def fast_atan2(y, x):
    return lib().orc_fast_atan2(y, x)


def retain_best(keys, n_points):
    keys = np.ascontiguousarray(keys, np.float32)
    perm = np.zeros(max(len(keys), 1), np.int32)
    n = lib().orc_retain_best(_p(keys), len(keys), n_points, _p(perm))
    return perm[:n].copy()


def camera(fx, fy, cx, cy, dist, width, height):
    f = lambda v: float(np.float32(v))  # the reference keeps intrinsics in cv::Mat_<float>
    return Camera(f(fx), f(fy), f(cx), f(cy), f(dist[0]), f(dist[1]), f(dist[2]), f(dist[3]), f(dist[4]),
                  width, height)


def undistort_points(cam, xy):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    lib().orc_undistort_points(C.byref(cam), _p(xy), len(xy), _p(out))
    return out


def image_bounds(cam):
    out = np.zeros(4, np.float32)
    lib().orc_image_bounds(C.byref(cam), _p(out))
    return out


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def match_hamming(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.zeros(len(q), np.int32)
    dist = np.zeros(len(q), np.int32)
    lib().orc_match_hamming(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
    return idx, dist


def matches_from(idx, dist):
    m = np.zeros(len(idx), MATCH_DTYPE)
    m["query"] = np.arange(len(idx))
    m["train"] = idx
    m["img"] = 0
    m["distance"] = dist.astype(np.float32)
    return m


def filter_matches(m, n_desc_rows=None):
    m = np.ascontiguousarray(m, MATCH_DTYPE).copy()
    n = lib().orc_filter_matches(_p(m), len(m), len(m) if n_desc_rows is None else n_desc_rows)
    return m[:n].copy()


def match_radius(kp, ref_kp, desc, ref_desc, max_pixel_dist):
    kp = np.ascontiguousarray(kp, KP_DTYPE)
    ref_kp = np.ascontiguousarray(ref_kp, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    ref_desc = np.ascontiguousarray(ref_desc, np.uint8)
    out = np.zeros(max(len(kp), 1), MATCH_DTYPE)
    n = lib().orc_match_radius(_p(kp), len(kp), _p(ref_kp), len(ref_kp), _p(desc), _p(ref_desc),
                               C.c_double(max_pixel_dist), _p(out))
    return out[:n].copy()


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    lib().orc_three_maxima(_p(sizes), len(sizes), C.byref(i1), C.byref(i2), C.byref(i3))
    return i1.value, i2.value, i3.value


def grid_params(bounds, cols=64, rows=64):
    min_x, max_x, min_y, max_y = [float(v) for v in bounds]
    uw = np.float32(max_x) - np.float32(min_x)
    uh = np.float32(max_y) - np.float32(min_y)
    # frame.cpp:309-310: int = float / int
    return GridParams(min_x, min_y, max_x, max_y, cols, rows, int(uw / cols), int(uh / rows))


def assign_grid(g, kps):
    k = np.ascontiguousarray(kps, KP_DTYPE).copy()
    ncell = g.grid_cols * g.grid_rows
    start = np.zeros(ncell + 1, np.int32)
    items = np.zeros(max(len(k), 1), np.int32)
    n = lib().orc_assign_grid(C.byref(g), _p(k), len(k), _p(start), _p(items))
    return k[:n].copy(), start, items[:n].copy()


def match_projection_frame(g, start, items, kps, desc, ref_kps, ref_has_mp, proj_xy, proj_z, ref_desc,
                           scale_factors, level_mode, radius, check_orientation):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    ref_kps = np.ascontiguousarray(ref_kps, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    ref_desc = np.ascontiguousarray(ref_desc, np.uint8)
    ref_has_mp = np.ascontiguousarray(ref_has_mp, np.uint8)
    proj_xy = np.ascontiguousarray(proj_xy, np.float32)
    proj_z = np.ascontiguousarray(proj_z, np.float32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.zeros(max(len(kps), 1), MATCH_DTYPE)
    n = lib().orc_match_projection_frame(C.byref(g), _p(start), _p(items), _p(kps), len(kps), _p(desc),
                                         _p(ref_kps), len(ref_kps), _p(ref_has_mp), _p(proj_xy), _p(proj_z),
                                         _p(ref_desc), _p(sf), level_mode, C.c_float(radius),
                                         int(check_orientation), _p(out))
    return out[:n].copy()


def match_projection_points(g, start, items, kps, desc, in_view, proj_xy, pred_level, view_cos, pt_desc,
                            scale_factors, radius, nn_ratio):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    in_view = np.ascontiguousarray(in_view, np.uint8)
    proj_xy = np.ascontiguousarray(proj_xy, np.float32)
    pred_level = np.ascontiguousarray(pred_level, np.int32)
    view_cos = np.ascontiguousarray(view_cos, np.float32)
    pt_desc = np.ascontiguousarray(pt_desc, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.zeros(max(len(kps), 1), MATCH_DTYPE)
    n = lib().orc_match_projection_points(C.byref(g), _p(start), _p(items), _p(kps), len(kps), _p(desc),
                                          len(in_view), _p(in_view), _p(proj_xy), _p(pred_level), _p(view_cos),
                                          _p(pt_desc), _p(sf), C.c_float(radius), C.c_float(nn_ratio), _p(out))
    return out[:n].copy()


def match_bow(kps, desc, ref_kps, ref_desc, ref_has_mp, nodes, ref_nodes, low_threshold, nn_ratio, check_orientation):
    """nodes / ref_nodes = (node_ids uint32 ascending, node_start int32 [n+1], node_items int32)."""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    ref_kps = np.ascontiguousarray(ref_kps, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    ref_desc = np.ascontiguousarray(ref_desc, np.uint8)
    flags = np.ascontiguousarray(ref_has_mp, np.uint8)
    ni, ns, nit = (np.ascontiguousarray(nodes[0], np.uint32), np.ascontiguousarray(nodes[1], np.int32),
                   np.ascontiguousarray(nodes[2], np.int32))
    ri, rs, rit = (np.ascontiguousarray(ref_nodes[0], np.uint32), np.ascontiguousarray(ref_nodes[1], np.int32),
                   np.ascontiguousarray(ref_nodes[2], np.int32))
    out = np.zeros(max(len(kps), 1), MATCH_DTYPE)
    n = lib().orc_match_bow(_p(kps), len(kps), _p(desc), _p(ref_kps), len(ref_kps), _p(ref_desc), _p(flags),
                            _p(ni), _p(ns), _p(nit), len(ni), _p(ri), _p(rs), _p(rit), len(ri), int(low_threshold),
                            C.c_float(nn_ratio), int(bool(check_orientation)), _p(out))
    return out[:n].copy()


def match_epipolar(kps, desc, ref_kps, ref_desc, has_mp, ref_has_mp, F, scale_sigmas, nodes, ref_nodes,
                   low_threshold, check_orientation):
    """Epipolar matcher; nodes as in match_bow.  Returns None when the reference would index out of range."""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    ref_kps = np.ascontiguousarray(ref_kps, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    ref_desc = np.ascontiguousarray(ref_desc, np.uint8)
    has_mp = np.ascontiguousarray(has_mp, np.uint8)
    ref_has_mp = np.ascontiguousarray(ref_has_mp, np.uint8)
    F = np.ascontiguousarray(F, np.float32).reshape(9)
    sig = np.ascontiguousarray(scale_sigmas, np.float32)
    ni, ns, nit = (np.ascontiguousarray(nodes[0], np.uint32), np.ascontiguousarray(nodes[1], np.int32),
                   np.ascontiguousarray(nodes[2], np.int32))
    ri, rs, rit = (np.ascontiguousarray(ref_nodes[0], np.uint32), np.ascontiguousarray(ref_nodes[1], np.int32),
                   np.ascontiguousarray(ref_nodes[2], np.int32))
    out = np.zeros(max(len(kps), 1), MATCH_DTYPE)
    n = lib().orc_match_epipolar(_p(kps), len(kps), _p(desc), _p(ref_kps), len(ref_kps), _p(ref_desc), _p(has_mp),
                                 _p(ref_has_mp), _p(F), _p(sig), _p(ni), _p(ns), _p(nit), len(ni), _p(ri), _p(rs),
                                 _p(rit), len(ri), int(low_threshold), int(bool(check_orientation)), _p(out))
    if n < 0:
        return None
    return out[:n].copy()


def fuse_search(g, start, items, kps, desc, in_view, proj_xy, pred_level, pt_desc, scale_factors, inv_scale_sigmas,
                radius_multiplier):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    in_view = np.ascontiguousarray(in_view, np.uint8)
    proj_xy = np.ascontiguousarray(proj_xy, np.float32)
    pred_level = np.ascontiguousarray(pred_level, np.int32)
    pt_desc = np.ascontiguousarray(pt_desc, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    inv = np.ascontiguousarray(inv_scale_sigmas, np.float32)
    best_idx = np.zeros(max(len(in_view), 1), np.int32)
    best_dist = np.zeros(max(len(in_view), 1), np.int32)
    lib().orc_fuse_search(C.byref(g), _p(start), _p(items), _p(kps), _p(desc), len(in_view), _p(in_view),
                          _p(proj_xy), _p(pred_level), _p(pt_desc), _p(sf), _p(inv), C.c_float(radius_multiplier),
                          _p(best_idx), _p(best_dist))
    return best_idx[:len(in_view)].copy(), best_dist[:len(in_view)].copy()


def best_descriptor(desc, point_start):
    desc = np.ascontiguousarray(desc, np.uint8)
    ps = np.ascontiguousarray(point_start, np.int32)
    best = np.zeros(max(len(ps) - 1, 1), np.int32)
    lib().orc_best_descriptor(_p(desc), _p(ps), len(ps) - 1, _p(best))
    return best[:len(ps) - 1].copy()


def svd_square_f64(A):
    A = np.ascontiguousarray(A, np.float64)
    n = A.shape[0]
    w, u, vt = np.zeros(n), np.zeros((n, n)), np.zeros((n, n))
    lib().orc_svd_square_f64(_p(A), n, _p(w), _p(u), _p(vt))
    return w, u, vt


def triangulate_points(P1, P2, x1, x2):
    P1 = np.ascontiguousarray(P1, np.float64)
    P2 = np.ascontiguousarray(P2, np.float64)
    x1 = np.ascontiguousarray(x1, np.float64).reshape(-1, 2)
    x2 = np.ascontiguousarray(x2, np.float64).reshape(-1, 2)
    out = np.zeros((max(len(x1), 1), 4))
    lib().orc_triangulate_points(_p(P1), _p(P2), _p(x1), _p(x2), len(x1), _p(out))
    return out[:len(x1)]


def recover_pose_fundamental(F, K, pts, ref):
    """Initializer::findRtWithFundamental -> (R, t, recoverPose's count)."""
    F = np.ascontiguousarray(F, np.float32).reshape(9)
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
    ref = np.ascontiguousarray(ref, np.float64).reshape(-1, 2)
    R, t = np.zeros((3, 3)), np.zeros(3)
    lib().orc_recover_pose_fundamental.restype = C.c_int
    good = lib().orc_recover_pose_fundamental(_p(F), _p(K), _p(pts), _p(ref), len(pts), _p(R), _p(t))
    return R, t, good


def decompose_homography(H, K):
    """cv::decomposeHomographyMat -> (R [nsol, 3, 3], t [nsol, 3], normals [nsol, 3])."""
    H = np.ascontiguousarray(H, np.float32).reshape(9)
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    R, t, nrm = np.zeros((4, 3, 3)), np.zeros((4, 3)), np.zeros((4, 3))
    lib().orc_decompose_homography.restype = C.c_int
    nsol = lib().orc_decompose_homography(_p(H), _p(K), _p(R), _p(t), _p(nrm))
    return R[:nsol], t[:nsol], nrm[:nsol]


def filter_homography_visible(R, nrm, before, after):
    """cv::filterHomographyDecompByVisibleRefpoints on CV_32FC2 points -> indices of the solutions kept."""
    R = np.ascontiguousarray(R, np.float64).reshape(-1, 9)
    nrm = np.ascontiguousarray(nrm, np.float64).reshape(-1, 3)
    before = np.ascontiguousarray(before, np.float32).reshape(-1, 2)
    after = np.ascontiguousarray(after, np.float32).reshape(-1, 2)
    sols = np.zeros(4, np.int32)
    lib().orc_filter_homography_visible.restype = C.c_int
    kept = lib().orc_filter_homography_visible(_p(R), _p(nrm), len(R), _p(before), _p(after), len(before), _p(sols))
    return sols[:kept]


def recover_pose_homography(H, K, pts, ref):
    """Initializer::findRtWithHomography with the filter's points as floats -> (R, t, index) or (None, None, -1)."""
    H = np.ascontiguousarray(H, np.float32).reshape(9)
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
    ref = np.ascontiguousarray(ref, np.float64).reshape(-1, 2)
    R, t = np.zeros((3, 3)), np.zeros(3)
    lib().orc_recover_pose_homography.restype = C.c_int
    idx = lib().orc_recover_pose_homography(_p(H), _p(K), _p(pts), _p(ref), len(pts), _p(R), _p(t))
    return (R, t, idx) if idx >= 0 else (None, None, -1)


def triangulate_initial(K, R, t, pts, ref):
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    t = np.ascontiguousarray(t, np.float64).reshape(3)
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
    ref = np.ascontiguousarray(ref, np.float64).reshape(-1, 2)
    out = np.zeros((max(len(pts), 1), 3))
    lib().orc_triangulate_initial(_p(K), _p(R), _p(t), _p(pts), _p(ref), len(pts), _p(out))
    return out[:len(pts)]


def normalize_points(xy):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    T = np.zeros(9, np.float32)
    lib().orc_normalize_points(_p(xy), len(xy), _p(out), _p(T))
    return out, T.reshape(3, 3)


def svd_f32(A):
    A = np.ascontiguousarray(A, np.float32)
    m, n = A.shape
    k = min(m, n)
    w = np.zeros(max(m, n), np.float32)
    u = np.zeros((m, m), np.float32)
    vt = np.zeros((n, n), np.float32)
    lib().orc_svd_f32(_p(A), m, n, _p(w), _p(u), _p(vt))
    return w[:k].copy(), u, vt


def fundamental_8pt(pts_norm, ref_norm):
    a = np.ascontiguousarray(pts_norm, np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(ref_norm, np.float32).reshape(-1, 2)
    F = np.zeros(9, np.float32)
    lib().orc_fundamental_8pt(_p(a), _p(b), len(a), _p(F))
    return F.reshape(3, 3)


def homography_dlt(pts_norm, ref_norm):
    a = np.ascontiguousarray(pts_norm, np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(ref_norm, np.float32).reshape(-1, 2)
    H = np.zeros(9, np.float32)
    lib().orc_homography_dlt(_p(a), _p(b), len(a), _p(H))
    return H.reshape(3, 3)


def ransac_sets(n, iterations, seed0):
    out = np.zeros((iterations, 8), np.uint32)
    lib().orc_ransac_sets(n, iterations, C.c_uint32(seed0), _p(out))
    return out


def ransac_fh(pts, ref, sets8, sigma=1.0):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    ref = np.ascontiguousarray(ref, np.float32).reshape(-1, 2)
    sets8 = np.ascontiguousarray(sets8, np.uint32).reshape(-1, 8)
    n = len(pts)
    F = np.zeros(9, np.float32)
    H = np.zeros(9, np.float32)
    inl_f = np.zeros(n, np.uint8)
    inl_h = np.zeros(n, np.uint8)
    sf, sh = C.c_double(), C.c_double()
    best = np.zeros(2, np.int32)
    lib().orc_ransac_fh(_p(pts), _p(ref), n, _p(sets8), len(sets8), C.c_double(sigma), _p(F), _p(H), _p(inl_f),
                        _p(inl_h), C.byref(sf), C.byref(sh), _p(best))
    return dict(F=F.reshape(3, 3), H=H.reshape(3, 3), inliers_f=inl_f, inliers_h=inl_h, score_f=sf.value,
                score_h=sh.value, best_it=best)


def ransac_fh_all(pts, ref, sets8, sigma=1.0):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    ref = np.ascontiguousarray(ref, np.float32).reshape(-1, 2)
    sets8 = np.ascontiguousarray(sets8, np.uint32).reshape(-1, 8)
    it = len(sets8)
    F = np.zeros((it, 9), np.float32)
    H = np.zeros((it, 9), np.float32)
    Hi = np.zeros((it, 9), np.float32)
    sf = np.zeros(it, np.float32)
    sh = np.zeros(it, np.float32)
    lib().orc_ransac_fh_all(_p(pts), _p(ref), len(pts), _p(sets8), it, C.c_double(sigma), _p(F), _p(H), _p(Hi),
                            _p(sf), _p(sh))
    return F, H, Hi, sf, sh
