"""ctypes binding of liborbx.so (the C ABI declared in include/orbx.h).

This is plumbing for tests and bench.py: every compute call goes through the C ABI into the HIP kernels.
There is no CPU fallback; a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORBX_LIB") or os.path.join(_HERE, "liborbx.so")   # ORBX_LIB: an experimental build

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
MATCH_DTYPE = np.dtype([("query", "<i4"), ("train", "<i4"), ("img", "<i4"), ("distance", "<f4")])

OK, BAD_ARG, CAPACITY, DEVICE, UNSUPPORTED, BUSY = range(6)

# every symbol include/orbx.h declares
SYMBOLS = [
    "orbx_status_string", "orbx_last_error", "orbx_device_count",
    "orbx_extractor_create", "orbx_extractor_destroy", "orbx_extractor_scale_tables", "orbx_detect", "orbx_compute",
    "orbx_extract_batch", "orbx_extractor_device_results", "orbx_extractor_layout", "orbx_extractor_read_plane",
    "orbx_extractor_last_timings", "orbx_matcher_create", "orbx_matcher_destroy", "orbx_match_hamming",
    "orbx_match_hamming_batch", "orbx_match_hamming_cross", "orbx_filter_matches", "orbx_match_radius", "orbx_assign_grid",
    "orbx_match_projection_frame", "orbx_match_projection_points", "orbx_ransac_create", "orbx_ransac_destroy",
    "orbx_ransac_fh", "orbx_ransac_fh_batch", "orbx_ransac_fh_batch_static", "orbx_ransac_fh_all", "orbx_ransac_make_sets",
    "orbx_ransac_last_timings", "orbx_extractor_set_frame_model", "orbx_undistort_points",
    "orbx_extractor_device_grid", "orbx_match_projection_frame_batch", "orbx_selftest_retain_best", "orbx_selftest_retain_best_device", "orbx_match_bow", "orbx_best_descriptor_batch", "orbx_match_epipolar", "orbx_fuse_search", "orbx_match_bow_batch", "orbx_match_epipolar_batch", "orbx_recover_pose_batch", "orbx_recover_pose_homography_batch", "orbx_triangulate_batch",
    "orbx_sequence_open", "orbx_sequence_close", "orbx_sequence_length", "orbx_sequence_entry", "orbx_png_decode",
    "orbx_png_read", "orbx_depth_to_float", "orbx_feeder_create", "orbx_feeder_destroy", "orbx_feeder_start_sequence",
    "orbx_feeder_start_memory", "orbx_feeder_next",
    "orbx_probe_issue_rate", "orbx_match_projection_points_batch", "orbx_extractor_finish", "orbx_extractor_set_image_cache",
    "orbx_extractor_set_device_select", "orbx_extractor_set_blur_fork", "orbx_extractor_set_stream", "orbx_matcher_set_stream", "orbx_ransac_set_stream", "orbx_matcher_set_hamming_kernel",
    "orbx_normalize_points", "orbx_fundamental_npt", "orbx_homography_dlt", "orbx_ransac_set_fork",
    "orbx_selftest_stream_pyramid", "orbx_extractor_set_pyramid_kernel", "orbx_extractor_set_pyramid_tail", "orbx_extractor_set_distribution",
]


class ExtractorParams(C.Structure):
    """orbx_extractor_params; defaults are ORBExtractor's (reference orb_extractor.h:29-37)."""
    _fields_ = [("n_key_points", C.c_int32), ("scale_factor", C.c_float), ("level_pyramid", C.c_int32),
                ("edge_threshold", C.c_int32), ("first_level", C.c_int32), ("wta_k", C.c_int32),
                ("patch_size", C.c_int32), ("score_threshold", C.c_int32)]


class Grid(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("grid_cols", C.c_int32), ("grid_rows", C.c_int32), ("grid_size_x", C.c_int32),
                ("grid_size_y", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("k1", C.c_double), ("k2", C.c_double), ("p1", C.c_double), ("p2", C.c_double), ("k3", C.c_double)]


def make_camera(fx, fy, cx, cy, dist):
    f = lambda v: float(np.float32(v))  # the reference keeps intrinsics and distortion in cv::Mat_<float>
    return Camera(f(fx), f(fy), f(cx), f(cy), *[f(v) for v in dist])


def undistort_points(cam, xy):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    _check(lib().orbx_undistort_points(C.byref(cam), _p(xy), len(xy), _p(out)), "orbx_undistort_points")
    return out


class OrbxError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib().orbx_status_string(status).decode()
        detail = lib().orbx_last_error().decode()
        super().__init__(f"{where}: {msg}" + (f" ({detail})" if detail and status in (DEVICE, CAPACITY, BUSY, UNSUPPORTED) else ""))


_lib = None
_hip = None


def _load_hip_runtime():
    """liborbx.so carries no DT_NEEDED on the HIP runtime: exactly one libamdhip64 may live in a process, and when
    PyTorch is installed it must be the one PyTorch bundles (torch loads that file by RPATH).  Load it globally
    before liborbx so its HIP symbols resolve against it."""
    global _hip
    if _hip is not None:
        return _hip
    candidates = []
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            candidates.append(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so"))
    except Exception:
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    candidates += [os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so"]
    err = None
    for c in candidates:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            _hip = C.CDLL(c, mode=C.RTLD_GLOBAL)
            return _hip
        except OSError as e:
            err = e
    raise ImportError(f"no HIP runtime (libamdhip64.so) could be loaded: {err}")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C orb_slam_amd/csrc` "
                              "(python -c 'import __graft_entry__ as g; g.build()')")
        _load_hip_runtime()
        _lib = C.CDLL(LIB_PATH)
        _lib.orbx_status_string.restype = C.c_char_p
        _lib.orbx_last_error.restype = C.c_char_p
        for name in SYMBOLS:
            getattr(_lib, name)
    return _lib


def _check(status, where):
    if status != OK:
        raise OrbxError(status, where)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    return lib().orbx_device_count()


def make_params(n_key_points=8000, scale_factor=1.2, level_pyramid=4, edge_threshold=31, first_level=0, wta_k=2,
                patch_size=31, score_threshold=20):
    return ExtractorParams(n_key_points, scale_factor, level_pyramid, edge_threshold, first_level, wta_k, patch_size,
                           score_threshold)


def _img(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    return img, w, h, w * ch, ch


class Extractor:
    """Handle over orbx_extractor (ORBExtractor of the reference)."""

    def __init__(self, params=None, max_width=640, max_height=480, max_batch=1, device=0, **kw):
        self.params = params or make_params(**kw)
        self._h = C.c_void_p()
        _check(lib().orbx_extractor_create(C.byref(self.params), max_width, max_height, max_batch, device,
                                           C.byref(self._h)), "orbx_extractor_create")

    def close(self):
        if self._h:
            lib().orbx_extractor_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(lib().orbx_extractor_set_stream(self._h, C.c_void_p(stream_ptr)), "orbx_extractor_set_stream")

    def scale_tables(self):
        n = self.params.level_pyramid
        arrs = [np.zeros(n, np.float32) for _ in range(4)]
        lsf = C.c_float()
        _check(lib().orbx_extractor_scale_tables(self._h, *[_p(a) for a in arrs], C.byref(lsf)), "scale_tables")
        return (lsf.value, *arrs)

    def layout(self, width, height):
        n = self.params.level_pyramid
        aw, ah = C.c_int(), C.c_int()
        rects = np.zeros((n, 4), np.int32)
        scales = np.zeros(n, np.float32)
        _check(lib().orbx_extractor_layout(self._h, width, height, C.byref(aw), C.byref(ah), _p(rects), _p(scales)),
               "orbx_extractor_layout")
        return aw.value, ah.value, rects, scales

    def read_plane(self, what, width, height, frame=0):
        aw, ah, _, _ = self.layout(width, height)
        out = np.zeros((ah, aw), np.uint8)
        _check(lib().orbx_extractor_read_plane(self._h, frame, what, _p(out), out.nbytes), "orbx_extractor_read_plane")
        return out

    def detect(self, image, capacity=100000):
        img, w, h, stride, ch = _img(image)
        out = np.zeros(capacity, KP_DTYPE)
        n = C.c_int()
        _check(lib().orbx_detect(self._h, _p(img), w, h, stride, ch, _p(out), capacity, C.byref(n)), "orbx_detect")
        return out[:n.value].copy()

    def compute(self, image, keypoints):
        img, w, h, stride, ch = _img(image)
        k = np.ascontiguousarray(keypoints, KP_DTYPE).copy()
        n = C.c_int(len(k))
        desc = np.zeros((max(len(k), 1), 32), np.uint8)
        _check(lib().orbx_compute(self._h, _p(img), w, h, stride, ch, _p(k), C.byref(n), _p(desc)), "orbx_compute")
        return k[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, d_images_ptr, n_frames, frame_stride, width, height, stride, channels, capacity,
                      to_host=True, deferred=False):
        """deferred (with to_host=False): return with the work queued; call finish() for the counts and status."""
        kps = np.zeros((n_frames, capacity), KP_DTYPE) if to_host else None
        counts = None if (deferred and not to_host) else np.zeros(n_frames, np.int32)
        desc = np.zeros((n_frames, capacity, 32), np.uint8) if to_host else None
        self._batch_frames = n_frames
        _check(lib().orbx_extract_batch(self._h, C.c_void_p(d_images_ptr), n_frames, C.c_size_t(frame_stride), width,
                                        height, stride, channels, _p(kps), capacity, _p(counts), _p(desc)),
               "orbx_extract_batch")
        return kps, counts, desc

    def finish(self):
        """orbx_extractor_finish -> (counts, redone)."""
        counts = np.zeros(getattr(self, "_batch_frames", 1), np.int32)
        redone = C.c_int(0)
        _check(lib().orbx_extractor_finish(self._h, _p(counts), C.byref(redone)), "orbx_extractor_finish")
        return counts, bool(redone.value)

    def set_blur_fork(self, enabled):
        """Blur of a batch on the handle's side stream (default) or in line (several batches in flight)."""
        _check(lib().orbx_extractor_set_blur_fork(self._h, int(bool(enabled))), "orbx_extractor_set_blur_fork")

    def set_distribution(self, mode, low_threshold=7, cell_size=30):
        """mode 1: per-cell adaptive FAST threshold + quad-tree distribution (opt-in; 0 = cv::ORB's selection)."""
        _check(lib().orbx_extractor_set_distribution(self._h, mode, low_threshold, cell_size),
               "orbx_extractor_set_distribution")

    def set_pyramid_tail(self, base_level, rows_per_step=0):
        """Levels above base_level in one launch (0 = off): see orbx_extractor_set_pyramid_tail."""
        _check(lib().orbx_extractor_set_pyramid_tail(self._h, base_level, rows_per_step),
               "orbx_extractor_set_pyramid_tail")

    def set_pyramid_kernel(self, kind, rows_per_step=0):
        """0 / 1: level by level, 2: the one-pass kernel whenever the geometry allows."""
        _check(lib().orbx_extractor_set_pyramid_kernel(self._h, kind, rows_per_step), "orbx_extractor_set_pyramid_kernel")

    def set_image_cache(self, enabled):
        _check(lib().orbx_extractor_set_image_cache(self._h, int(enabled)), "orbx_extractor_set_image_cache")

    def set_frame_model(self, camera=None, grid=None):
        _check(lib().orbx_extractor_set_frame_model(self._h, C.byref(camera) if camera is not None else None,
                                                    C.byref(grid) if grid is not None else None),
               "orbx_extractor_set_frame_model")

    def set_device_select(self, enabled):
        _check(lib().orbx_extractor_set_device_select(self._h, int(enabled)), "orbx_extractor_set_device_select")

    def device_grid(self):
        a, b = C.c_void_p(), C.c_void_p()
        _check(lib().orbx_extractor_device_grid(self._h, C.byref(a), C.byref(b)), "orbx_extractor_device_grid")
        return a.value, b.value

    def device_results(self):
        kp, desc, cnt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        cap = C.c_int()
        _check(lib().orbx_extractor_device_results(self._h, C.byref(kp), C.byref(desc), C.byref(cnt), C.byref(cap)),
               "orbx_extractor_device_results")
        return kp.value, desc.value, cnt.value, cap.value

    def last_timings(self):
        ms = np.zeros(10, np.float32)
        n = lib().orbx_extractor_last_timings(self._h, _p(ms), 10)
        names = ["gray", "resize", "fast", "nms", "emit", "harris", "blur", "brief", "select", "angle"]
        return dict(zip(names[:n], ms[:n].tolist()))


class BowPair(C.Structure):
    _fields_ = [("keypoints", C.c_void_p), ("n", C.c_int), ("descriptors", C.c_void_p),
                ("ref_keypoints", C.c_void_p), ("n_ref", C.c_int), ("ref_descriptors", C.c_void_p),
                ("ref_has_point", C.c_void_p),
                ("node_ids", C.c_void_p), ("node_start", C.c_void_p), ("node_items", C.c_void_p), ("n_nodes", C.c_int),
                ("ref_node_ids", C.c_void_p), ("ref_node_start", C.c_void_p), ("ref_node_items", C.c_void_p),
                ("n_ref_nodes", C.c_int)]


class EpipolarPair(C.Structure):
    _fields_ = [("keypoints", C.c_void_p), ("n", C.c_int), ("descriptors", C.c_void_p),
                ("ref_keypoints", C.c_void_p), ("n_ref", C.c_int), ("ref_descriptors", C.c_void_p),
                ("has_point", C.c_void_p), ("ref_has_point", C.c_void_p), ("fundamental", C.c_void_p),
                ("node_ids", C.c_void_p), ("node_start", C.c_void_p), ("node_items", C.c_void_p), ("n_nodes", C.c_int),
                ("ref_node_ids", C.c_void_p), ("ref_node_start", C.c_void_p), ("ref_node_items", C.c_void_p),
                ("n_ref_nodes", C.c_int)]


class Matcher:
    """Handle over orbx_matcher (ORBMatcher of the reference)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().orbx_matcher_create(device, C.byref(self._h)), "orbx_matcher_create")

    def close(self):
        if self._h:
            lib().orbx_matcher_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(lib().orbx_matcher_set_stream(self._h, C.c_void_p(stream_ptr)), "orbx_matcher_set_stream")

    def set_hamming_kernel(self, kind):
        _check(lib().orbx_matcher_set_hamming_kernel(self._h, int(kind)), "orbx_matcher_set_hamming_kernel")

    def match_hamming(self, query, train):
        q = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
        idx = np.zeros(len(q), np.int32)
        dist = np.zeros(len(q), np.int32)
        _check(lib().orbx_match_hamming(self._h, _p(q), len(q), _p(t), len(t), _p(idx), _p(dist)),
               "orbx_match_hamming")
        return idx, dist

    def match_projection_points_batch(self, grid, n_pairs, d_frame_index, d_kps, d_desc, d_counts, kp_capacity,
                                      d_cell_start, d_cell_items, points_capacity, d_n_points, d_in_view, d_proj_xy,
                                      d_pred_level, d_view_cos, d_point_desc, scale_factors, radius, nn_ratio,
                                      d_matched, d_distance, d_n_matches):
        sf = np.ascontiguousarray(scale_factors, np.float32)
        v = C.c_void_p
        _check(lib().orbx_match_projection_points_batch(
            self._h, C.byref(grid), n_pairs, v(d_frame_index), v(d_kps), v(d_desc), v(d_counts), kp_capacity,
            v(d_cell_start), v(d_cell_items), points_capacity, v(d_n_points), v(d_in_view), v(d_proj_xy),
            v(d_pred_level), v(d_view_cos), v(d_point_desc), _p(sf), len(sf), C.c_float(radius), C.c_float(nn_ratio),
            v(d_matched), v(d_distance), v(d_n_matches)), "orbx_match_projection_points_batch")

    def match_hamming_batch(self, d_query, d_n_query, query_stride, d_train, d_n_train, train_stride, n_pairs,
                            max_query, max_train, d_idx, d_dist, out_stride):
        _check(lib().orbx_match_hamming_batch(self._h, C.c_void_p(d_query), C.c_void_p(d_n_query),
                                              C.c_size_t(query_stride), C.c_void_p(d_train), C.c_void_p(d_n_train),
                                              C.c_size_t(train_stride), n_pairs, max_query, max_train,
                                              C.c_void_p(d_idx), C.c_void_p(d_dist), C.c_size_t(out_stride)),
               "orbx_match_hamming_batch")

    def match_hamming_cross(self, d_query, d_n_query, query_stride, n_query_sets, d_train, d_n_train, train_stride,
                            n_train_sets, max_query, max_train, d_idx, d_dist, out_stride):
        """Every query set against every train set (device pointers); pair q * n_train_sets + t."""
        _check(lib().orbx_match_hamming_cross(self._h, C.c_void_p(d_query), C.c_void_p(d_n_query),
                                              C.c_size_t(query_stride), n_query_sets, C.c_void_p(d_train),
                                              C.c_void_p(d_n_train), C.c_size_t(train_stride), n_train_sets,
                                              max_query, max_train, C.c_void_p(d_idx), C.c_void_p(d_dist),
                                              C.c_size_t(out_stride)), "orbx_match_hamming_cross")

    def match(self, descriptors, ref_descriptors, filter_matches=True):
        """ORBMatcher::match(descriptors, ref_descriptors, matches, filter) (orb_matcher.cpp:719-733)."""
        idx, dist = self.match_hamming(descriptors, ref_descriptors)
        m = np.zeros(len(idx), MATCH_DTYPE)
        m["query"] = np.arange(len(idx))
        m["train"] = idx
        m["img"] = 0
        m["distance"] = dist.astype(np.float32)
        if filter_matches:
            m = filter_matches_host(m, len(idx))
        return m

    def match_radius(self, kp, ref_kp, desc, ref_desc, max_pixel_dist):
        kp = np.ascontiguousarray(kp, KP_DTYPE)
        ref_kp = np.ascontiguousarray(ref_kp, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        ref_desc = np.ascontiguousarray(ref_desc, np.uint8)
        out = np.zeros(max(len(kp), 1), MATCH_DTYPE)
        n = C.c_int()
        _check(lib().orbx_match_radius(self._h, _p(kp), len(kp), _p(ref_kp), len(ref_kp), _p(desc), _p(ref_desc),
                                       C.c_double(max_pixel_dist), _p(out), C.byref(n)), "orbx_match_radius")
        return out[:n.value].copy()

    def match_projection_frame(self, g, start, items, kps, desc, ref_kps, ref_has_mp, proj_xy, proj_z, ref_desc,
                               scale_factors, level_mode, radius, check_orientation):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        ref_kps = np.ascontiguousarray(ref_kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        ref_desc = np.ascontiguousarray(ref_desc, np.uint8)
        ref_has_mp = np.ascontiguousarray(ref_has_mp, np.uint8)
        proj_xy = np.ascontiguousarray(proj_xy, np.float32)
        proj_z = np.ascontiguousarray(proj_z, np.float32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        start = np.ascontiguousarray(start, np.int32)
        items = np.ascontiguousarray(items, np.int32)
        out = np.zeros(max(len(kps), 1), MATCH_DTYPE)
        n = C.c_int()
        _check(lib().orbx_match_projection_frame(self._h, C.byref(g), _p(start), _p(items), _p(kps), len(kps),
                                                 _p(desc), _p(ref_kps), len(ref_kps), _p(ref_has_mp), _p(proj_xy),
                                                 _p(proj_z), _p(ref_desc), _p(sf), len(sf), level_mode,
                                                 C.c_float(radius), int(check_orientation), _p(out), C.byref(n)),
               "orbx_match_projection_frame")
        return out[:n.value].copy()

    def match_projection_frame_batch(self, g, n_pairs, d_frame_index, d_ref_index, d_kp, d_desc, d_counts, kp_cap,
                                     d_cell_start, d_cell_items, d_has, d_proj_xy, d_proj_z, scale_factors,
                                     level_mode, radius, check_orientation, d_matched, d_dist, d_n_matches):
        sf = np.ascontiguousarray(scale_factors, np.float32)
        _check(lib().orbx_match_projection_frame_batch(
            self._h, C.byref(g), n_pairs, C.c_void_p(d_frame_index), C.c_void_p(d_ref_index), C.c_void_p(d_kp),
            C.c_void_p(d_desc), C.c_void_p(d_counts), kp_cap, C.c_void_p(d_cell_start), C.c_void_p(d_cell_items),
            C.c_void_p(d_has), C.c_void_p(d_proj_xy), C.c_void_p(d_proj_z), _p(sf), len(sf), level_mode,
            C.c_float(radius), int(check_orientation), C.c_void_p(d_matched), C.c_void_p(d_dist),
            C.c_void_p(d_n_matches)), "orbx_match_projection_frame_batch")

    def match_projection_points(self, g, start, items, kps, desc, in_view, proj_xy, pred_level, view_cos, pt_desc,
                                scale_factors, radius, nn_ratio):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        in_view = np.ascontiguousarray(in_view, np.uint8)
        proj_xy = np.ascontiguousarray(proj_xy, np.float32)
        pred_level = np.ascontiguousarray(pred_level, np.int32)
        view_cos = np.ascontiguousarray(view_cos, np.float32)
        pt_desc = np.ascontiguousarray(pt_desc, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        start = np.ascontiguousarray(start, np.int32)
        items = np.ascontiguousarray(items, np.int32)
        out = np.zeros(max(len(kps), 1), MATCH_DTYPE)
        n = C.c_int()
        _check(lib().orbx_match_projection_points(self._h, C.byref(g), _p(start), _p(items), _p(kps), len(kps),
                                                  _p(desc), len(in_view), _p(in_view), _p(proj_xy), _p(pred_level),
                                                  _p(view_cos), _p(pt_desc), _p(sf), len(sf), C.c_float(radius),
                                                  C.c_float(nn_ratio), _p(out), C.byref(n)),
               "orbx_match_projection_points")
        return out[:n.value].copy()

    def best_descriptor_batch(self, desc, point_start):
        desc = np.ascontiguousarray(desc, np.uint8)
        ps = np.ascontiguousarray(point_start, np.int32)
        best = np.zeros(max(len(ps) - 1, 1), np.int32)
        _check(lib().orbx_best_descriptor_batch(self._h, _p(desc), _p(ps), len(ps) - 1, _p(best)),
               "orbx_best_descriptor_batch")
        return best[:len(ps) - 1].copy()

    def match_bow(self, kps, desc, ref_kps, ref_desc, ref_has_point, nodes, ref_nodes, low_threshold=50, nn_ratio=0.6,
                  check_orientation=True):
        """nodes / ref_nodes = (node_ids uint32 ascending, node_start int32 [n+1], node_items int32)."""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        ref_kps = np.ascontiguousarray(ref_kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        ref_desc = np.ascontiguousarray(ref_desc, np.uint8)
        flags = np.ascontiguousarray(ref_has_point, np.uint8)
        ni, ns, nit = (np.ascontiguousarray(nodes[0], np.uint32), np.ascontiguousarray(nodes[1], np.int32),
                       np.ascontiguousarray(nodes[2], np.int32))
        ri, rs, rit = (np.ascontiguousarray(ref_nodes[0], np.uint32), np.ascontiguousarray(ref_nodes[1], np.int32),
                       np.ascontiguousarray(ref_nodes[2], np.int32))
        out = np.zeros(max(len(kps), 1), MATCH_DTYPE)
        n = C.c_int()
        _check(lib().orbx_match_bow(self._h, _p(kps), len(kps), _p(desc), _p(ref_kps), len(ref_kps), _p(ref_desc),
                                    _p(flags), _p(ni), _p(ns), _p(nit), len(ni), _p(ri), _p(rs), _p(rit), len(ri),
                                    int(low_threshold), C.c_float(nn_ratio), int(bool(check_orientation)), _p(out),
                                    C.byref(n)), "orbx_match_bow")
        return out[:n.value].copy()

    def match_epipolar(self, kps, desc, ref_kps, ref_desc, has_point, ref_has_point, F, scale_sigmas, nodes,
                       ref_nodes, low_threshold=50, check_orientation=True):
        """nodes / ref_nodes as in match_bow; F = 3x3 float32 fundamental matrix."""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        ref_kps = np.ascontiguousarray(ref_kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        ref_desc = np.ascontiguousarray(ref_desc, np.uint8)
        has = np.ascontiguousarray(has_point, np.uint8)
        ref_has = np.ascontiguousarray(ref_has_point, np.uint8)
        F = np.ascontiguousarray(F, np.float32).reshape(9)
        sig = np.ascontiguousarray(scale_sigmas, np.float32)
        ni, ns, nit = (np.ascontiguousarray(nodes[0], np.uint32), np.ascontiguousarray(nodes[1], np.int32),
                       np.ascontiguousarray(nodes[2], np.int32))
        ri, rs, rit = (np.ascontiguousarray(ref_nodes[0], np.uint32), np.ascontiguousarray(ref_nodes[1], np.int32),
                       np.ascontiguousarray(ref_nodes[2], np.int32))
        out = np.zeros(max(len(kps), 1), MATCH_DTYPE)
        n = C.c_int()
        _check(lib().orbx_match_epipolar(self._h, _p(kps), len(kps), _p(desc), _p(ref_kps), len(ref_kps),
                                         _p(ref_desc), _p(has), _p(ref_has), _p(F), _p(sig), len(sig), _p(ni), _p(ns),
                                         _p(nit), len(ni), _p(ri), _p(rs), _p(rit), len(ri), int(low_threshold),
                                         int(bool(check_orientation)), _p(out), C.byref(n)), "orbx_match_epipolar")
        return out[:n.value].copy()

    def _nodes(self, nodes, keep):
        arrs = (np.ascontiguousarray(nodes[0], np.uint32), np.ascontiguousarray(nodes[1], np.int32),
                np.ascontiguousarray(nodes[2], np.int32))
        keep.extend(arrs)
        return arrs

    def match_bow_batch(self, pairs, low_threshold=50, nn_ratio=0.6, check_orientation=True):
        """pairs: list of (kps, desc, ref_kps, ref_desc, ref_has_point, nodes, ref_nodes) -> list of match arrays."""
        keep, recs = [], (BowPair * max(len(pairs), 1))()
        for i, (kps, desc, ref_kps, ref_desc, flags, nodes, ref_nodes) in enumerate(pairs):
            kps = np.ascontiguousarray(kps, KP_DTYPE); ref_kps = np.ascontiguousarray(ref_kps, KP_DTYPE)
            desc = np.ascontiguousarray(desc, np.uint8); ref_desc = np.ascontiguousarray(ref_desc, np.uint8)
            flags = np.ascontiguousarray(flags, np.uint8)
            keep.extend([kps, ref_kps, desc, ref_desc, flags])
            ni, ns, nit = self._nodes(nodes, keep)
            ri, rs, rit = self._nodes(ref_nodes, keep)
            recs[i] = BowPair(_p(kps), len(kps), _p(desc), _p(ref_kps), len(ref_kps), _p(ref_desc), _p(flags),
                              _p(ni), _p(ns), _p(nit), len(ni), _p(ri), _p(rs), _p(rit), len(ri))
        total = sum(len(p[0]) for p in pairs)
        out = np.zeros(max(total, 1), MATCH_DTYPE)
        counts = np.zeros(max(len(pairs), 1), np.int32)
        _check(lib().orbx_match_bow_batch(self._h, recs, len(pairs), int(low_threshold), C.c_float(nn_ratio),
                                          int(bool(check_orientation)), _p(out), _p(counts)), "orbx_match_bow_batch")
        res, base = [], 0
        for i, p in enumerate(pairs):
            res.append(out[base:base + counts[i]].copy())
            base += len(p[0])
        return res

    def match_epipolar_batch(self, pairs, scale_sigmas, low_threshold=50, check_orientation=True):
        """pairs: list of (kps, desc, ref_kps, ref_desc, has_point, ref_has_point, F, nodes, ref_nodes)."""
        keep, recs = [], (EpipolarPair * max(len(pairs), 1))()
        sig = np.ascontiguousarray(scale_sigmas, np.float32)
        for i, (kps, desc, ref_kps, ref_desc, has, ref_has, F, nodes, ref_nodes) in enumerate(pairs):
            kps = np.ascontiguousarray(kps, KP_DTYPE); ref_kps = np.ascontiguousarray(ref_kps, KP_DTYPE)
            desc = np.ascontiguousarray(desc, np.uint8); ref_desc = np.ascontiguousarray(ref_desc, np.uint8)
            has = np.ascontiguousarray(has, np.uint8); ref_has = np.ascontiguousarray(ref_has, np.uint8)
            F = np.ascontiguousarray(F, np.float32).reshape(9)
            keep.extend([kps, ref_kps, desc, ref_desc, has, ref_has, F])
            ni, ns, nit = self._nodes(nodes, keep)
            ri, rs, rit = self._nodes(ref_nodes, keep)
            recs[i] = EpipolarPair(_p(kps), len(kps), _p(desc), _p(ref_kps), len(ref_kps), _p(ref_desc), _p(has),
                                   _p(ref_has), _p(F), _p(ni), _p(ns), _p(nit), len(ni), _p(ri), _p(rs), _p(rit),
                                   len(ri))
        total = sum(len(p[0]) for p in pairs)
        out = np.zeros(max(total, 1), MATCH_DTYPE)
        counts = np.zeros(max(len(pairs), 1), np.int32)
        _check(lib().orbx_match_epipolar_batch(self._h, recs, len(pairs), _p(sig), len(sig), int(low_threshold),
                                               int(bool(check_orientation)), _p(out), _p(counts)),
               "orbx_match_epipolar_batch")
        res, base = [], 0
        for i, p in enumerate(pairs):
            res.append(out[base:base + counts[i]].copy())
            base += len(p[0])
        return res

    def fuse_search(self, g, start, items, kps, desc, in_view, proj_xy, pred_level, pt_desc, scale_factors,
                    inv_scale_sigmas, radius_multiplier):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        in_view = np.ascontiguousarray(in_view, np.uint8)
        proj_xy = np.ascontiguousarray(proj_xy, np.float32)
        pred_level = np.ascontiguousarray(pred_level, np.int32)
        pt_desc = np.ascontiguousarray(pt_desc, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        inv = np.ascontiguousarray(inv_scale_sigmas, np.float32)
        start = np.ascontiguousarray(start, np.int32)
        items = np.ascontiguousarray(items, np.int32)
        best_idx = np.zeros(max(len(in_view), 1), np.int32)
        best_dist = np.zeros(max(len(in_view), 1), np.int32)
        _check(lib().orbx_fuse_search(self._h, C.byref(g), _p(start), _p(items), _p(kps), len(kps), _p(desc),
                                      len(in_view), _p(in_view), _p(proj_xy), _p(pred_level), _p(pt_desc), _p(sf),
                                      _p(inv), len(sf), C.c_float(radius_multiplier), _p(best_idx), _p(best_dist)),
               "orbx_fuse_search")
        return best_idx[:len(in_view)].copy(), best_dist[:len(in_view)].copy()


def filter_matches_host(matches, n_descriptor_rows):
    m = np.ascontiguousarray(matches, MATCH_DTYPE).copy()
    n = C.c_int(len(m))
    _check(lib().orbx_filter_matches(_p(m), C.byref(n), n_descriptor_rows), "orbx_filter_matches")
    return m[:n.value].copy()


def make_grid(bounds, cols=64, rows=64):
    """bounds = (min_x, max_x, min_y, max_y); cell sizes as Frame::setupGrid computes them (int = float / int)."""
    min_x, max_x, min_y, max_y = [np.float32(v) for v in bounds]
    return Grid(float(min_x), float(min_y), float(max_x), float(max_y), cols, rows, int((max_x - min_x) / cols),
                int((max_y - min_y) / rows))


def assign_grid(g, kps):
    k = np.ascontiguousarray(kps, KP_DTYPE).copy()
    ncell = g.grid_cols * g.grid_rows
    start = np.zeros(ncell + 1, np.int32)
    items = np.zeros(max(len(k), 1), np.int32)
    n = C.c_int(len(k))
    _check(lib().orbx_assign_grid(C.byref(g), _p(k), C.byref(n), _p(start), _p(items)), "orbx_assign_grid")
    return k[:n.value].copy(), start, items[:n.value].copy()


def selftest_retain_best(keys, n_points):
    keys = np.ascontiguousarray(keys, np.float32)
    perm = np.zeros(max(len(keys), 1), np.int32)
    n = C.c_int()
    _check(lib().orbx_selftest_retain_best(_p(keys), len(keys), n_points, _p(perm), C.byref(n)),
           "orbx_selftest_retain_best")
    return perm[:n.value].copy()


def selftest_retain_best_device(keys, n_points, device=0):
    keys = np.ascontiguousarray(keys, np.float32)
    perm = np.zeros(max(len(keys), 1), np.int32)
    n = C.c_int()
    _check(lib().orbx_selftest_retain_best_device(device, _p(keys), len(keys), n_points, _p(perm), C.byref(n)),
           "orbx_selftest_retain_best_device")
    return perm[:n.value].copy()


def selftest_stream_pyramid(params, image, rows_per_step=8, max_lds_bytes=65536, width=None, height=None,
                            base_level=0, atlas=None):
    """Plan of the one-pass pyramid kernel for an image size and, with an image, the plan executed on the CPU.
    base_level > 0: the tail form; `atlas` must then hold pyramid level base_level already (width, height give the
    image size) and the levels above it are computed into it.
    Returns (info dict, atlas or None); raises OrbxError(UNSUPPORTED) when the geometry does not suit the kernel."""
    info = np.zeros(4, np.int32)
    if base_level > 0 and atlas is not None:
        atlas = np.ascontiguousarray(atlas, np.uint8)
        _check(lib().orbx_selftest_stream_pyramid(C.byref(params), None, width, height, width, 1, rows_per_step,
                                                  C.c_size_t(max_lds_bytes), base_level, _p(atlas),
                                                  C.c_size_t(atlas.size), _p(info)), "orbx_selftest_stream_pyramid")
    elif image is None:
        _check(lib().orbx_selftest_stream_pyramid(C.byref(params), None, width, height, width, 1, rows_per_step,
                                                  C.c_size_t(max_lds_bytes), base_level, None, C.c_size_t(0), _p(info)),
               "orbx_selftest_stream_pyramid")
        atlas = None
    else:
        img, w, h, stride, ch = _img(image)
        _check(lib().orbx_selftest_stream_pyramid(C.byref(params), None, w, h, stride, ch, rows_per_step,
                                                  C.c_size_t(max_lds_bytes), 0, None, C.c_size_t(0), _p(info)),
               "orbx_selftest_stream_pyramid")
        atlas = np.zeros((int(info[3]), int(info[2])), np.uint8)
        _check(lib().orbx_selftest_stream_pyramid(C.byref(params), _p(img), w, h, stride, ch, rows_per_step,
                                                  C.c_size_t(max_lds_bytes), 0, _p(atlas), C.c_size_t(atlas.size),
                                                  _p(info)), "orbx_selftest_stream_pyramid")
    return dict(steps=int(info[0]), lds_bytes=int(info[1]), atlas_w=int(info[2]), atlas_h=int(info[3])), atlas


def probe_issue_rate(kind, device=0):
    """Wave-instructions per second of the chip for one instruction mix (orbx_probe_issue_rate)."""
    v = C.c_double()
    _check(lib().orbx_probe_issue_rate(device, kind, C.byref(v)), "orbx_probe_issue_rate")
    return v.value


def ransac_make_sets(n, iterations, seed0):
    out = np.zeros((iterations, 8), np.uint32)
    _check(lib().orbx_ransac_make_sets(n, iterations, C.c_uint32(seed0), _p(out)), "orbx_ransac_make_sets")
    return out


class Ransac:
    """Handle over orbx_ransac (Initializer::findFundamentalAndHomography of the reference)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().orbx_ransac_create(device, C.byref(self._h)), "orbx_ransac_create")

    def close(self):
        if self._h:
            lib().orbx_ransac_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(lib().orbx_ransac_set_stream(self._h, C.c_void_p(stream_ptr)), "orbx_ransac_set_stream")

    def set_fork(self, enabled):
        _check(lib().orbx_ransac_set_fork(self._h, 1 if enabled else 0), "orbx_ransac_set_fork")

    def recover_pose_batch(self, F, K, pts, ref, pair_start):
        """F: pairs x 3 x 3 float32, K: 3 x 3 float32, pts / ref: total x 2 float64, pair_start: pairs + 1."""
        F = np.ascontiguousarray(F, np.float32).reshape(-1, 9)
        K = np.ascontiguousarray(K, np.float32).reshape(9)
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
        ref = np.ascontiguousarray(ref, np.float64).reshape(-1, 2)
        ps = np.ascontiguousarray(pair_start, np.int32)
        P = len(ps) - 1
        R = np.zeros((max(P, 1), 3, 3), np.float64)
        t = np.zeros((max(P, 1), 3), np.float64)
        good = np.zeros(max(P, 1), np.int32)
        _check(lib().orbx_recover_pose_batch(self._h, P, _p(F), _p(K), _p(pts), _p(ref), _p(ps), _p(R), _p(t),
                                             _p(good)), "orbx_recover_pose_batch")
        return R[:P], t[:P], good[:P]

    def recover_pose_homography_batch(self, H, K, pts, ref, pair_start):
        """Initializer::findRtWithHomography with the visibility filter's points as floats.  Returns (R, t, index of
        the chosen motion or -1, all motions [pairs, 4, 15] as (R 9, t 3, n 3), number of motions 1 | 4)."""
        H = np.ascontiguousarray(H, np.float32).reshape(-1, 9)
        K = np.ascontiguousarray(K, np.float32).reshape(9)
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
        ref = np.ascontiguousarray(ref, np.float64).reshape(-1, 2)
        ps = np.ascontiguousarray(pair_start, np.int32)
        P = len(ps) - 1
        if len(H) != P or len(pts) != len(ref) or (P >= 0 and len(pts) != (ps[-1] if len(ps) else 0)):
            raise ValueError("recover_pose_homography_batch: shapes do not agree")
        R = np.zeros((max(P, 1), 3, 3), np.float64)
        t = np.zeros((max(P, 1), 3), np.float64)
        idx = np.zeros(max(P, 1), np.int32)
        sols = np.zeros((max(P, 1), 4, 15), np.float64)
        nsol = np.zeros(max(P, 1), np.int32)
        _check(lib().orbx_recover_pose_homography_batch(self._h, P, _p(H), _p(K), _p(pts), _p(ref), _p(ps), _p(R),
                                                        _p(t), _p(idx), _p(sols), _p(nsol)),
               "orbx_recover_pose_homography_batch")
        return R[:P], t[:P], idx[:P], sols[:P], nsol[:P]

    def triangulate_batch(self, K, R, t, pts, ref, pair_start):
        K = np.ascontiguousarray(K, np.float32).reshape(9)
        R = np.ascontiguousarray(R, np.float64).reshape(-1, 9)
        t = np.ascontiguousarray(t, np.float64).reshape(-1, 3)
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
        ref = np.ascontiguousarray(ref, np.float64).reshape(-1, 2)
        ps = np.ascontiguousarray(pair_start, np.int32)
        out = np.zeros((max(len(pts), 1), 3), np.float64)
        _check(lib().orbx_triangulate_batch(self._h, len(ps) - 1, _p(K), _p(R), _p(t), _p(pts), _p(ref), _p(ps),
                                            _p(out)), "orbx_triangulate_batch")
        return out[:len(pts)]

    def normalize_points(self, xy):
        """geometry::normalizePoints on the device: (normalised n x 2, T 3 x 3)."""
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        out = np.zeros_like(xy)
        T = np.zeros(9, np.float32)
        _check(lib().orbx_normalize_points(self._h, _p(xy), len(xy), _p(out), _p(T)), "orbx_normalize_points")
        return out, T.reshape(3, 3)

    def _npt(self, fn, name, pts, ref):
        pts = np.ascontiguousarray(pts, np.float32)
        ref = np.ascontiguousarray(ref, np.float32)
        single = pts.ndim == 2
        pts = pts.reshape((1,) + pts.shape) if single else pts
        ref = ref.reshape(pts.shape)
        out = np.zeros((pts.shape[0], 9), np.float32)
        _check(fn(self._h, _p(pts), _p(ref), pts.shape[1], pts.shape[0], _p(out)), name)
        out = out.reshape(-1, 3, 3)
        return out[0] if single else out

    def fundamental_npt(self, pts, ref):
        """geometry::computeFundamentalMat for n x 2 points (or systems x n x 2)."""
        return self._npt(lib().orbx_fundamental_npt, "orbx_fundamental_npt", pts, ref)

    def homography_dlt(self, pts, ref):
        """geometry::computeHomographyMat for n x 2 points (or systems x n x 2)."""
        return self._npt(lib().orbx_homography_dlt, "orbx_homography_dlt", pts, ref)

    def fh(self, pts, ref, sets8, sigma=1.0):
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
        _check(lib().orbx_ransac_fh(self._h, _p(pts), _p(ref), n, _p(sets8), len(sets8), C.c_double(sigma), _p(F),
                                    _p(H), _p(inl_f), _p(inl_h), C.byref(sf), C.byref(sh), _p(best)),
               "orbx_ransac_fh")
        return dict(F=F.reshape(3, 3), H=H.reshape(3, 3), inliers_f=inl_f, inliers_h=inl_h, score_f=sf.value,
                    score_h=sh.value, best_it=best)

    def fh_all(self, pts, ref, sets8, sigma=1.0):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        ref = np.ascontiguousarray(ref, np.float32).reshape(-1, 2)
        sets8 = np.ascontiguousarray(sets8, np.uint32).reshape(-1, 8)
        it = len(sets8)
        F = np.zeros((it, 9), np.float32)
        H = np.zeros((it, 9), np.float32)
        Hi = np.zeros((it, 9), np.float32)
        sf = np.zeros(it, np.float32)
        sh = np.zeros(it, np.float32)
        _check(lib().orbx_ransac_fh_all(self._h, _p(pts), _p(ref), len(pts), _p(sets8), it, C.c_double(sigma), _p(F),
                                        _p(H), _p(Hi), _p(sf), _p(sh)), "orbx_ransac_fh_all")
        return F, H, Hi, sf, sh

    def fh_batch(self, d_pts, d_ref, n, d_sets, iterations, n_pairs, sigma, d_F, d_H, d_inl_f, d_inl_h, d_scores,
                 d_best):
        _check(lib().orbx_ransac_fh_batch(self._h, C.c_void_p(d_pts), C.c_void_p(d_ref), n, C.c_void_p(d_sets),
                                          iterations, n_pairs, C.c_double(sigma), C.c_void_p(d_F), C.c_void_p(d_H),
                                          C.c_void_p(d_inl_f), C.c_void_p(d_inl_h), C.c_void_p(d_scores),
                                          C.c_void_p(d_best)), "orbx_ransac_fh_batch")

    def fh_batch_static(self, d_pts, d_ref, n, d_sets, iterations, n_pairs, sigma, d_F, d_H, d_inl_f, d_inl_h,
                        d_scores, d_best):
        """orbx_ransac_fh_batch_static: the same batch through the library's fixed device workspace (no allocation)."""
        _check(lib().orbx_ransac_fh_batch_static(self._h, C.c_void_p(d_pts), C.c_void_p(d_ref), n,
                                                 C.c_void_p(d_sets), iterations, n_pairs, C.c_double(sigma),
                                                 C.c_void_p(d_F), C.c_void_p(d_H), C.c_void_p(d_inl_f),
                                                 C.c_void_p(d_inl_h), C.c_void_p(d_scores), C.c_void_p(d_best)),
               "orbx_ransac_fh_batch_static")

    def last_timings(self):
        ms = np.zeros(6, np.float32)
        n = lib().orbx_ransac_last_timings(self._h, _p(ms), 6)
        names = ["normalize", "solve_f", "score_f", "score_h", "select", "solve_h"]
        return dict(zip(names[:n], ms[:n].tolist()))


IMREAD_UNCHANGED, IMREAD_COLOR = -1, 1


def png_decode(data, mode=IMREAD_UNCHANGED):
    """cv::imread's result for PNG bytes: (h, w[, c]) uint8 or uint16, channels in BGR(A) order."""
    buf = np.frombuffer(data, np.uint8)
    w, h, ch, bps = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _check(lib().orbx_png_decode(_p(buf), C.c_size_t(len(buf)), mode, None, C.c_size_t(0), C.byref(w), C.byref(h),
                                 C.byref(ch), C.byref(bps)), "orbx_png_decode")
    out = np.zeros(h.value * w.value * ch.value * bps.value, np.uint8)
    _check(lib().orbx_png_decode(_p(buf), C.c_size_t(len(buf)), mode, _p(out), C.c_size_t(len(out)), None, None, None,
                                 None), "orbx_png_decode")
    arr = out.view(np.uint16) if bps.value == 2 else out
    return arr.reshape(h.value, w.value, ch.value) if ch.value > 1 else arr.reshape(h.value, w.value)


def png_read(path, mode=IMREAD_UNCHANGED):
    with open(path, "rb") as f:
        return png_decode(f.read(), mode)


def depth_to_float(bgr8, depth_scale=np.float32(255 * 0.0002)):
    src = np.ascontiguousarray(bgr8, np.uint8)
    out = np.zeros(src.shape, np.float32)
    _check(lib().orbx_depth_to_float(_p(src), C.c_size_t(src.size), C.c_float(depth_scale), _p(out)),
           "orbx_depth_to_float")
    return out


class Sequence:
    """TUM association list (TUMRGBDCamera::setupCameraStream of the reference)."""

    def __init__(self, dataset_dir):
        self._h = C.c_void_p()
        self.dir = dataset_dir
        _check(lib().orbx_sequence_open(dataset_dir.encode(), C.byref(self._h)), "orbx_sequence_open")

    def __len__(self):
        return lib().orbx_sequence_length(self._h)

    def entry(self, i):
        t, rgb, depth = C.c_double(), C.c_char_p(), C.c_char_p()
        _check(lib().orbx_sequence_entry(self._h, i, C.byref(t), C.byref(rgb), C.byref(depth)), "orbx_sequence_entry")
        return t.value, rgb.value.decode(), depth.value.decode()

    def close(self):
        if self._h:
            lib().orbx_sequence_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Feeder:
    """Batches of frames uploaded ahead of the consumer (orbx_feeder)."""

    def __init__(self, width, height, channels, batch, n_slots=3, n_threads=8, device=0):
        self._h = C.c_void_p()
        self.frame_bytes = width * height * channels
        _check(lib().orbx_feeder_create(device, width, height, channels, batch, n_slots, n_threads,
                                        C.byref(self._h)), "orbx_feeder_create")

    def start_sequence(self, seq, first, count):
        self._keep = seq
        _check(lib().orbx_feeder_start_sequence(self._h, seq._h, first, count), "orbx_feeder_start_sequence")

    def start_memory(self, frames, already_pinned=False):
        """frames: C-contiguous uint8 array (n, ...) of decoded frames; must stay alive during the job."""
        self._keep = frames
        n = frames.shape[0]
        stride = frames.strides[0]
        _check(lib().orbx_feeder_start_memory(self._h, C.c_void_p(frames.ctypes.data), C.c_size_t(stride), n,
                                              int(already_pinned)), "orbx_feeder_start_memory")

    def next(self, stream_ptr=None):
        """-> (device pointer, frames); frames == 0 at the end of the job."""
        d, n = C.c_void_p(), C.c_int()
        _check(lib().orbx_feeder_next(self._h, C.c_void_p(stream_ptr) if stream_ptr else None, C.byref(d),
                                      C.byref(n)), "orbx_feeder_next")
        return d.value, n.value

    def close(self):
        if self._h:
            lib().orbx_feeder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
