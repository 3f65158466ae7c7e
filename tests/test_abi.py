"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/orbx.h declares; the
host-only entry points (no GPU work) agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from orb_slam_amd import orbx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "orbx.h")).read()
    declared = set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = orbx.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(orbx.SYMBOLS)


def test_status_strings():
    lib = orbx.lib()
    assert lib.orbx_status_string(0) == b"ok"
    assert lib.orbx_status_string(2) == b"capacity exceeded"


def test_struct_layouts_match_opencv():
    assert orbx.KP_DTYPE.itemsize == 28 and orbx.MATCH_DTYPE.itemsize == 16
    assert C.sizeof(orbx.ExtractorParams) == 32 and C.sizeof(orbx.Grid) == 32


def test_bad_arguments_do_not_touch_the_gpu():
    lib = orbx.lib()
    assert lib.orbx_extractor_create(None, 640, 480, 1, 0, None) == orbx.BAD_ARG
    p = orbx.make_params(wta_k=3)
    h = C.c_void_p()
    assert lib.orbx_extractor_create(C.byref(p), 640, 480, 1, 0, C.byref(h)) == orbx.UNSUPPORTED
    n = C.c_int(-1)
    assert lib.orbx_filter_matches(None, C.byref(n), 0) == orbx.BAD_ARG
    assert lib.orbx_ransac_make_sets(4, 10, 1, None) == orbx.BAD_ARG


def test_filter_matches_matches_oracle(oracle):
    rng = np.random.default_rng(5)
    for n in (1, 2, 50, 1000):
        m = np.zeros(n, orbx.MATCH_DTYPE)
        m["query"] = np.arange(n)
        m["train"] = rng.integers(0, max(2, n // 3), n)
        m["distance"] = rng.integers(0, 90, n).astype(np.float32)
        a = orbx.filter_matches_host(m, n)
        b = oracle.filter_matches(m, n)
        assert np.array_equal(a, b.astype(orbx.MATCH_DTYPE))
    assert len(orbx.filter_matches_host(np.zeros(0, orbx.MATCH_DTYPE), 0)) == 0


def test_assign_grid_matches_oracle(oracle):
    rng = np.random.default_rng(6)
    k = np.zeros(3000, orbx.KP_DTYPE)
    k["x"] = rng.uniform(-5, 645, len(k)).astype(np.float32)
    k["y"] = rng.uniform(-5, 485, len(k)).astype(np.float32)
    k["octave"] = rng.integers(0, 8, len(k))
    for bounds in ((0, 640, 0, 480), (10.803785, 626.0599, 14.68408, 473.3248)):
        g = orbx.make_grid(bounds)
        og = oracle.grid_params(bounds)
        assert (g.grid_size_x, g.grid_size_y) == (og.grid_size_x, og.grid_size_y)
        ka, sa, ia = orbx.assign_grid(g, k)
        kb, sb, ib = oracle.assign_grid(og, k)
        assert np.array_equal(ka, kb) and np.array_equal(sa, sb) and np.array_equal(ia, ib)


def test_ransac_sets_match_oracle(oracle):
    for n, it, seed in ((8, 5, 1), (100, 50, 7), (2000, 200, 3)):
        assert np.array_equal(orbx.ransac_make_sets(n, it, seed), oracle.ransac_sets(n, it, seed))
    s = orbx.ransac_make_sets(50, 20, 2)
    assert all(len(set(r.tolist())) == 8 for r in s)


def retain_best_cases():
    rng = np.random.default_rng(11)
    cases = []
    for n in (0, 1, 2, 3, 4, 5, 17, 100, 1000, 4583):
        for kind in ("ties", "float", "sorted", "reversed", "const"):
            if kind == "ties":
                keys = rng.integers(20, 60, n).astype(np.float32)
            elif kind == "float":
                keys = rng.normal(size=n).astype(np.float32)
            elif kind == "sorted":
                keys = np.arange(n, dtype=np.float32)
            elif kind == "reversed":
                keys = np.arange(n, dtype=np.float32)[::-1].copy()
            else:
                keys = np.full(n, 7, np.float32)
            for npts in sorted({0, 1, 2, n // 3, n // 2, max(n - 1, 0), n, n + 5}):
                cases.append((keys, npts))
    # median-of-3 killer sequence: drives introselect into its heap-select fallback
    n = 4096
    k = np.zeros(n, np.float32)
    half = n // 2
    for i in range(half):
        if i % 2 == 0:
            k[i] = i + 1
        else:
            k[i] = half + i + (half % 2)
    k[half:] = 2 * (np.arange(half) + 1)
    cases.append((-k, 1000))
    cases.append((k, 37))
    return cases


def test_retain_best_replay_matches_libstdcxx(oracle):
    """The library's replay of std::nth_element + std::partition leaves the kept keypoints in exactly the order the
    real libstdc++ calls do, including tie-heavy inputs (FAST scores are small integers) and the heap-select path."""
    for keys, npts in retain_best_cases():
        a = orbx.selftest_retain_best(keys, npts)
        b = oracle.retain_best(keys, npts)
        assert np.array_equal(a, b), (len(keys), npts)


def test_no_fallback_without_library_or_gpu(monkeypatch):
    """The product path has no CPU fallback: a missing liborbx.so raises at the first use, and without a GPU the
    extractor cannot be created."""
    import torch
    monkeypatch.setattr(orbx, "_lib", None)
    monkeypatch.setattr(orbx, "LIB_PATH", orbx.LIB_PATH + ".absent")
    with pytest.raises(ImportError):
        orbx.lib()
    monkeypatch.undo()
    if not torch.cuda.is_available():
        with pytest.raises(orbx.OrbxError):
            orbx.Extractor(n_key_points=100, level_pyramid=2, max_width=64, max_height=64)
        with pytest.raises(orbx.OrbxError):
            orbx.Matcher()


def test_one_pass_pyramid_plan_executes_to_the_oracle_pyramid(oracle):
    """The one-pass pyramid kernel only executes tables the host makes (rows per step and level, ring depths, per-column
    and per-row coefficients).  Here the same tables run on the CPU: the result must be the oracle's pyramid, padded
    rectangles included, for colour and gray images, odd sizes and several step heights; and geometries the kernel does
    not suit must be refused, not mis-planned."""
    from tests.conftest import synth_image
    cases = [(synth_image(21, 640, 480, 3), 8, 8), (synth_image(22, 640, 480), 8, 4), (synth_image(23, 333, 257, 3), 4, 6),
             (synth_image(24, 401, 299), 5, 5), (synth_image(25, 704, 480, 3), 8, 8), (synth_image(26, 880, 600), 5, 6)]
    for img, nlevels, rows in cases:
        prm = orbx.make_params(n_key_points=1000, level_pyramid=nlevels)
        p = oracle.orb_params(nfeatures=1000, nlevels=nlevels)
        info, atlas = orbx.selftest_stream_pyramid(prm, img, rows_per_step=rows, max_lds_bytes=160 * 1024)
        oatlas, rects, _ = oracle.pyramid(p, img)
        assert atlas.shape == oatlas.shape
        for l, (rx, ry, rw, rh) in enumerate(rects):
            b = 32
            assert np.array_equal(atlas[ry - b:ry + rh + b, rx - b:rx + rw + b],
                                  oatlas[ry - b:ry + rh + b, rx - b:rx + rw + b]), (img.shape, nlevels, rows, l)
        assert info["steps"] >= img.shape[0] // rows and 0 < info["lds_bytes"] <= 160 * 1024
        # the tail form: levels up to `base` taken from the atlas as given, the levels above it computed from there
        h, w = img.shape[:2]
        ran = 0
        for base, trows in ((1, 8), (nlevels // 2, 8), (nlevels // 2, 16), (nlevels - 2, 22)):
            a = oatlas.copy()
            for rx, ry, rw, rh in rects[base + 1:]:
                a[ry - 32:ry + rh + 32, rx - 32:rx + rw + 32] = 0x5a
            try:
                tinfo, out = orbx.selftest_stream_pyramid(prm, None, rows_per_step=trows, width=w, height=h,
                                                          base_level=base, atlas=a, max_lds_bytes=160 * 1024)
            except orbx.OrbxError as e:
                assert e.status == orbx.UNSUPPORTED, (img.shape, base, trows)   # too many rows a step for this width
                continue
            ran += 1
            for l, (rx, ry, rw, rh) in enumerate(rects):
                assert np.array_equal(out[ry - 32:ry + rh + 32, rx - 32:rx + rw + 32],
                                      oatlas[ry - 32:ry + rh + 32, rx - 32:rx + rw + 32]), (img.shape, base, trows, l)
        assert ran >= 2, img.shape
    # the VGA / 8-level plan the batch path uses fits several workgroups per CU
    prm = orbx.make_params(n_key_points=1000, level_pyramid=8)
    info, _ = orbx.selftest_stream_pyramid(prm, None, rows_per_step=8, width=640, height=480)
    assert info["lds_bytes"] <= 48 * 1024, info
    # a level lower than twice the border, rings beyond the LDS allowed, more columns or more level-0 items per step
    # than the workgroup's threads own: refused
    for kw in (dict(width=150, height=120, rows_per_step=8), dict(width=640, height=480, rows_per_step=8, max_lds_bytes=4096),
               dict(width=1280, height=720, rows_per_step=8, max_lds_bytes=160 * 1024),
               dict(width=640, height=480, rows_per_step=32, max_lds_bytes=160 * 1024)):
        with pytest.raises(orbx.OrbxError) as e:
            orbx.selftest_stream_pyramid(prm, None, **kw)
        assert e.value.status == orbx.UNSUPPORTED
