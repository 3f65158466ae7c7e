"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on identical inputs.
Integer / byte / index results must be bit-exact; floating-point results here are bit-exact as well because the
kernels evaluate the reference's arithmetic in the reference's order (tolerance 0 unless a test says otherwise)."""
import numpy as np
import pytest

from orb_slam_amd import orbx
from tests.conftest import synth_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ex8000():
    e = orbx.Extractor(n_key_points=8000, level_pyramid=8, max_width=1024, max_height=768, max_batch=8)
    yield e
    e.close()


@pytest.fixture(scope="module")
def ex1000():
    e = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=1024, max_height=768, max_batch=8)
    yield e
    e.close()


@pytest.fixture(scope="module")
def matcher():
    m = orbx.Matcher()
    yield m
    m.close()


@pytest.fixture(scope="module")
def ransac():
    r = orbx.Ransac()
    yield r
    r.close()


def _cases(test_images):
    return [("img1", test_images[0]), ("img2", test_images[1]), ("gray1", test_images[0][:, :, 1].copy()),
            ("synth_odd", synth_image(11, 333, 257)), ("synth_color", synth_image(12, 401, 299, 3)),
            ("small", synth_image(13, 150, 120))]


def test_layout_matches_oracle(oracle, ex8000):
    p = oracle.orb_params(nfeatures=8000)
    for (w, h) in ((640, 480), (333, 257), (1024, 768), (150, 120)):
        aw, ah, rects, scales = ex8000.layout(w, h)
        oaw, oah, orects, oscales = oracle.layout(p, w, h)
        assert (aw, ah) == (oaw, oah) and np.array_equal(rects, orects) and np.array_equal(scales, oscales)


def test_pyramid_fast_blur_planes_bit_exact(oracle, ex8000, test_images):
    p = oracle.orb_params(nfeatures=8000)
    for name, img in _cases(test_images):
        h, w = img.shape[:2]
        k = ex8000.detect(img)
        ok_, desc = ex8000.compute(img, k)
        atlas = ex8000.read_plane(0, w, h)
        score = ex8000.read_plane(1, w, h)
        blur = ex8000.read_plane(2, w, h)
        oatlas, rects, _ = oracle.pyramid(p, img)
        oblur = oracle.blur(p, oatlas, w, h)
        for l, (rx, ry, rw, rh) in enumerate(rects):
            b = 32
            assert np.array_equal(atlas[ry - b:ry + rh + b, rx - b:rx + rw + b],
                                  oatlas[ry - b:ry + rh + b, rx - b:rx + rw + b]), (name, "pyramid", l)
            lvl = np.ascontiguousarray(oatlas[ry:ry + rh, rx:rx + rw])
            assert np.array_equal(score[ry:ry + rh, rx:rx + rw], oracle.fast_scores(lvl, 20)), (name, "fast", l)
            assert np.array_equal(blur[ry:ry + rh, rx:rx + rw], oblur[ry:ry + rh, rx:rx + rw]), (name, "blur", l)


@pytest.mark.parametrize("nkp", [8000, 1000])
def test_detect_and_compute_bit_exact(oracle, ex8000, ex1000, test_images, nkp):
    ex = ex8000 if nkp == 8000 else ex1000
    p = oracle.orb_params(nfeatures=nkp)
    for name, img in _cases(test_images):
        k = ex.detect(img)
        ok_ = oracle.detect(p, img)
        assert len(k) == len(ok_), (name, len(k), len(ok_))
        assert np.array_equal(k, ok_.astype(orbx.KP_DTYPE)), name
        k2, d = ex.compute(img, k)
        ok2, od = oracle.compute(p, img, ok_)
        assert np.array_equal(k2, ok2.astype(orbx.KP_DTYPE)) and np.array_equal(d, od), name
    if nkp == 8000:  # the reference's own known answers, through the HIP path
        assert len(ex.detect(test_images[0])) == 4583 and len(ex.detect(test_images[1])) == 4652


def test_compute_on_moved_unsorted_keypoints(oracle, ex1000, test_images):
    # compute() after undistortion: float coordinates, some pushed over the border, octaves shuffled
    p = oracle.orb_params(nfeatures=1000)
    rng = np.random.default_rng(4)
    img = test_images[0]
    k = oracle.detect(p, img)
    k["x"] += rng.uniform(-6, 6, len(k)).astype(np.float32)
    k["y"] += rng.uniform(-6, 6, len(k)).astype(np.float32)
    k[:5]["x"] = np.float32(30.5)
    k[5:9]["y"] = np.float32(449.5)
    for perm in (np.arange(len(k)), rng.permutation(len(k))):
        kk = k[perm]
        a_k, a_d = ex1000.compute(img, kk)
        o_k, o_d = oracle.compute(p, img, kk)
        assert len(a_k) == len(o_k) < len(kk)
        assert np.array_equal(a_k, o_k.astype(orbx.KP_DTYPE)) and np.array_equal(a_d, o_d)
    e_k, e_d = ex1000.compute(img, k[:0])
    assert len(e_k) == 0


def test_detect_edge_cases(oracle, ex1000):
    p = oracle.orb_params(nfeatures=1000)
    flat = np.full((120, 160), 77, np.uint8)
    assert len(ex1000.detect(flat)) == 0 == len(oracle.detect(p, flat))
    tiny = synth_image(3, 70, 66)  # upper levels are smaller than twice the border: they yield nothing
    assert np.array_equal(ex1000.detect(tiny), oracle.detect(p, tiny).astype(orbx.KP_DTYPE))
    # the size given at creation is a reservation: a larger image makes the buffers grow (cv::ORB takes any size)
    big = synth_image(7, 1300, 900)
    assert np.array_equal(ex1000.detect(big), oracle.detect(p, big).astype(orbx.KP_DTYPE))
    assert np.array_equal(ex1000.detect(tiny), oracle.detect(p, tiny).astype(orbx.KP_DTYPE))
    with pytest.raises(orbx.OrbxError) as e:
        ex1000.detect(np.zeros((8, 16400), np.uint8))
    assert e.value.status == orbx.UNSUPPORTED
    img = synth_image(5, 320, 240)
    with pytest.raises(orbx.OrbxError) as e:
        ex1000.detect(img, capacity=10)
    assert e.value.status == orbx.CAPACITY
    # a lattice of isolated dots with varying brightness: more than eight keypoints in one 64-pixel run of a row
    # (the ninth and later scores of a run take the separate path), and many equal FAST scores
    rng = np.random.default_rng(77)
    dots = np.full((240, 320), 40, np.uint8)
    dots[36:204:6, 36:284:5] = rng.integers(120, 255, dots[36:204:6, 36:284:5].shape).astype(np.uint8)
    ex = orbx.Extractor(n_key_points=3000, level_pyramid=3, max_width=320, max_height=240)
    p3 = oracle.orb_params(nfeatures=3000, nlevels=3)
    got, want = ex.detect(dots), oracle.detect(p3, dots)
    assert len(want) > 1000
    assert np.array_equal(got, want.astype(orbx.KP_DTYPE))
    k2, d2 = ex.compute(dots, got)
    ok2, od2 = oracle.compute(p3, dots, want)
    assert np.array_equal(k2, ok2.astype(orbx.KP_DTYPE)) and np.array_equal(d2, od2)
    ex.close()


def test_one_handle_serves_one_call_at_a_time(oracle, test_images):
    """A handle entered from a second thread while a call is running answers ORBX_BUSY and touches nothing: every call
    that does go through returns the right keypoints, whichever thread made it."""
    import threading
    p = oracle.orb_params(nfeatures=1000)
    want = oracle.detect(p, test_images[0]).astype(orbx.KP_DTYPE)
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480)
    ex.detect(test_images[0])
    stop = threading.Event()
    seen = {"busy": 0, "ok": 0, "wrong": 0, "other": 0}

    def second_caller():
        while not stop.is_set():
            try:
                k = ex.detect(test_images[0])
                seen["ok" if np.array_equal(k, want) else "wrong"] += 1
            except orbx.OrbxError as e:
                seen["busy" if e.status == orbx.BUSY else "other"] += 1

    t = threading.Thread(target=second_caller)
    t.start()
    mine = {"busy": 0, "ok": 0, "wrong": 0, "other": 0}
    for _ in range(300):
        try:
            k = ex.detect(test_images[0])
            mine["ok" if np.array_equal(k, want) else "wrong"] += 1
        except orbx.OrbxError as e:
            mine["busy" if e.status == orbx.BUSY else "other"] += 1
    stop.set()
    t.join()
    assert seen["wrong"] == 0 and mine["wrong"] == 0 and seen["other"] == 0 and mine["other"] == 0, (seen, mine)
    assert seen["busy"] + mine["busy"] > 0 and seen["ok"] + mine["ok"] > 0, (seen, mine)
    ex.close()


def test_compute_reuses_the_pyramid_of_detect(oracle, test_images):
    """detect() then compute() on the same image (MonoFrame::extractFeatures): the second call reuses the uploaded
    image and its pyramid.  Same results as with the reuse switched off; a different image behind the same buffer
    address, or one edited in place, is noticed; and the stage timings show that the reusing compute built no pyramid."""
    p = oracle.orb_params(nfeatures=1000)
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480)
    buf = np.ascontiguousarray(test_images[0]).copy()
    want = {}
    for i, img in enumerate(test_images):
        ok = oracle.detect(p, img)
        want[i] = (ok,) + oracle.compute(p, img, ok)
    for cache in (False, True, True):
        ex.set_image_cache(cache)
        buf[...] = test_images[0]
        k = ex.detect(buf)
        k2, d = ex.compute(buf, k)
        t = ex.last_timings()
        assert (t["gray"] == 0 and t["resize"] == 0) == cache       # no pyramid kernels ran in the reusing compute
        assert np.array_equal(k, want[0][0].astype(orbx.KP_DTYPE)) and np.array_equal(d, want[0][2])
        k3, d3 = ex.compute(buf, k)                                 # compute twice in a row
        assert np.array_equal(d3, want[0][2])
        # the other image written into the SAME buffer between detect and compute: compute must describe what it is given
        buf[...] = test_images[1]
        k4, d4 = ex.compute(buf, want[1][0].astype(orbx.KP_DTYPE))
        assert np.array_equal(k4, want[1][1].astype(orbx.KP_DTYPE)) and np.array_equal(d4, want[1][2])
        # a block edited in place
        k = ex.detect(buf)
        buf[0:40, 100:300] = 255 - buf[0:40, 100:300]
        ok = oracle.detect(p, buf)
        wk, wd = oracle.compute(p, buf, ok)
        k5, d5 = ex.compute(buf, ok.astype(orbx.KP_DTYPE))
        assert np.array_equal(d5, wd)
        # an edit of a few pixels anywhere - single rows in the middle of the image, one pixel at a time: every pixel
        # is part of the fingerprint, so the compute after it never describes the old pixels
        stale_would_differ = 0
        for row in (1, 7, 203, 331, 478):
            k = ex.detect(buf)
            kps = oracle.detect(p, buf)
            _, old_d = oracle.compute(p, buf, kps)
            near = kps[(np.abs(kps["y"] - row) < 6) & (kps["octave"] == 0)]
            x = int(near["x"][0]) if len(near) else 320
            buf[row, max(x - 3, 0):x + 4] = 255 - buf[row, max(x - 3, 0):x + 4]
            wk, wd = oracle.compute(p, buf, kps)
            stale_would_differ += not np.array_equal(wd, old_d)
            k6, d6 = ex.compute(buf, kps.astype(orbx.KP_DTYPE))
            assert np.array_equal(d6, wd), row
        assert stale_would_differ >= 2   # the edits do change descriptors: a stale pyramid would have been caught
    ex.close()


def test_blur_in_line_equals_blur_on_the_side_stream(test_images):
    """orbx_extractor_set_blur_fork only changes when the blur runs: keypoints, descriptors and grid of a batch are the
    same bits with it on the side stream (default) and in line."""
    import torch
    frames = [test_images[0], test_images[1], np.ascontiguousarray(test_images[0][::-1])]
    dev = torch.from_numpy(np.stack(frames)).cuda()
    torch.cuda.synchronize()
    out = []
    for fork in (True, False, True):
        ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=len(frames))
        ex.set_blur_fork(fork)
        for _ in range(2):   # twice on the same handle: the second call starts while nothing of the first is pending
            kps, counts, desc = ex.extract_batch(dev.data_ptr(), len(frames), 640 * 480 * 3, 640, 480, 640 * 3, 3, 1280)
        out.append((kps.copy(), counts.copy(), desc.copy()))
        ex.close()
    for kps, counts, desc in out[1:]:
        assert np.array_equal(counts, out[0][1])
        for f in range(len(frames)):
            n = counts[f]
            assert np.array_equal(kps[f, :n], out[0][0][f, :n]) and np.array_equal(desc[f, :n], out[0][2][f, :n])


def test_deferred_batch_and_finish(oracle, test_images):
    """orbx_extract_batch without host outputs returns with the work queued; orbx_extractor_finish reports the counts
    and the status.  Same device results as the synchronous call; a frame with more candidates than the on-device
    selection holds is finished by the host path inside finish() and flagged; too small a capacity is reported there."""
    import torch
    w, h = 640, 480
    rng = np.random.default_rng(17)
    dense = rng.integers(0, 16, (h, w)).astype(np.uint8)
    for b in range(h // 3):                           # ~20000 corners on one level: beyond the selection kernel's LDS;
        n = len(dense[3 * b, (2 * b) % 4::4])         # random heights, so that scores and responses rarely tie
        dense[3 * b, (2 * b) % 4::4] = rng.integers(120, 256, n)
    gray = [np.ascontiguousarray(test_images[0][:, :, 1]), np.ascontiguousarray(test_images[1][:, :, 1])]
    p = oracle.orb_params(nfeatures=1000, nlevels=1)
    for frames, expect_redo in ((gray, False), ([gray[0], dense], True)):
        dev = torch.from_numpy(np.stack(frames)).cuda()
        torch.cuda.synchronize()
        ex = orbx.Extractor(n_key_points=1000, level_pyramid=1, max_width=w, max_height=h, max_batch=2)
        cap = 1280
        ex.extract_batch(dev.data_ptr(), 2, w * h, w, h, w, 1, cap, to_host=False, deferred=True)
        counts, redone = ex.finish()
        assert redone == expect_redo
        kp_ptr, desc_ptr, cnt_ptr, kcap = ex.device_results()
        kp = torch.as_tensor(_DevView(kp_ptr, (2, kcap, 7), "<f4"), device="cuda").cpu().numpy().view(np.uint8)
        kp = np.ascontiguousarray(kp).view(orbx.KP_DTYPE).reshape(2, kcap)
        desc = torch.as_tensor(_DevView(desc_ptr, (2, kcap, 32), "|u1"), device="cuda").cpu().numpy()
        dcnt = torch.as_tensor(_DevView(cnt_ptr, (2,), "<i4"), device="cuda").cpu().numpy()
        assert np.array_equal(dcnt, counts)
        for f, img in enumerate(frames):
            ok_, od = oracle.compute(p, img, oracle.detect(p, img))
            assert counts[f] == len(ok_)
            assert np.array_equal(kp[f, :counts[f]], ok_.astype(orbx.KP_DTYPE)) and np.array_equal(desc[f, :counts[f]], od)
        # finish() without an outstanding batch hands the same counts over again
        again, redone2 = ex.finish()
        assert np.array_equal(again, counts) and not redone2
        # capacity too small: reported by finish(), not by the call that queued the work
        ex.extract_batch(dev.data_ptr(), 2, w * h, w, h, w, 1, 300, to_host=False, deferred=True)
        with pytest.raises(orbx.OrbxError) as e:
            ex.finish()
        assert e.value.status == orbx.CAPACITY
        # a deferred batch that is never finished is dropped by the next call
        ex.extract_batch(dev.data_ptr(), 2, w * h, w, h, w, 1, cap, to_host=False, deferred=True)
        kps, c2, d2 = ex.extract_batch(dev.data_ptr(), 2, w * h, w, h, w, 1, cap)
        assert np.array_equal(c2, counts)
        ex.close()


def test_more_corners_than_the_candidate_list_holds(oracle):
    """cv::ORB has no limit on the number of FAST corners; the candidate list here starts at one per 16 pyramid pixels
    and must grow when a frame has more.  Bright dots on the lattice spanned by (4, 0) and (2, 3) put no lattice point
    on any dot's radius-3 ring, so every dot is a corner and a strict 3x3 maximum: one corner per 12 pixels."""
    import torch
    w, h = 640, 480
    img = np.zeros((h, w), np.uint8)
    for b in range(h // 3):
        img[3 * b, (2 * b) % 4::4] = 255
    p = oracle.orb_params(nfeatures=50000, nlevels=1)
    want = oracle.detect(p, img)
    assert len(want) > w * h // 16                       # more than the list's initial capacity
    wk, wd = oracle.compute(p, img, want)
    ex = orbx.Extractor(n_key_points=50000, level_pyramid=1, max_width=w, max_height=h, max_batch=2)
    got = ex.detect(img)
    assert np.array_equal(got, want.astype(orbx.KP_DTYPE))
    k, d = ex.compute(img, got)
    assert np.array_equal(k, wk.astype(orbx.KP_DTYPE)) and np.array_equal(d, wd)
    ex.close()
    # the batch path: a sparse frame beside the dense one, on a fresh handle (its list starts small again)
    ex = orbx.Extractor(n_key_points=50000, level_pyramid=1, max_width=w, max_height=h, max_batch=2)
    sparse = synth_image(83, w, h, 1)
    sk, sd = oracle.compute(p, sparse, oracle.detect(p, sparse))
    dev = torch.from_numpy(np.stack([sparse, img])).cuda()
    torch.cuda.synchronize()
    kps, counts, desc = ex.extract_batch(dev.data_ptr(), 2, w * h, w, h, w, 1, 24000)
    assert counts[0] == len(sk) and counts[1] == len(wk)
    assert np.array_equal(kps[0, :counts[0]], sk.astype(orbx.KP_DTYPE)) and np.array_equal(desc[0, :counts[0]], sd)
    assert np.array_equal(kps[1, :counts[1]], wk.astype(orbx.KP_DTYPE)) and np.array_equal(desc[1, :counts[1]], wd)
    ex.close()


def test_extract_batch_equals_per_frame(oracle, test_images):
    import torch
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=6)
    p = oracle.orb_params(nfeatures=1000)
    frames = [test_images[0], test_images[1]] + [synth_image(20 + i, 640, 480, 3) for i in range(4)]
    dev = torch.from_numpy(np.stack(frames)).cuda()
    torch.cuda.synchronize()
    cap = 1500
    ora = []
    for img in frames:
        ok_ = oracle.detect(p, img)
        ora.append(oracle.compute(p, img, ok_))
    for device_select in (True, False):  # both forms of the selection step must give the reference order
        ex.set_device_select(device_select)
        kps, counts, desc = ex.extract_batch(dev.data_ptr(), len(frames), 640 * 480 * 3, 640, 480, 640 * 3, 3, cap)
        for f, (ok2, od) in enumerate(ora):
            assert counts[f] == len(ok2), (device_select, f)
            assert np.array_equal(kps[f, :counts[f]], ok2.astype(orbx.KP_DTYPE)), (device_select, f)
            assert np.array_equal(desc[f, :counts[f]], od), (device_select, f)
    with pytest.raises(orbx.OrbxError) as e:
        ex.extract_batch(dev.data_ptr(), len(frames), 640 * 480 * 3, 640, 480, 640 * 3, 3, 500)
    assert e.value.status == orbx.CAPACITY
    ex.close()


def test_distribution_mode_bit_exact(oracle, test_images):
    """orbx_extractor_set_distribution (opt-in, no counterpart in the reference): per-cell adaptive FAST threshold +
    quad-tree distribution.  The rule is stated sequentially in oracle/distribute_oracle.cpp (a self-oracle); the
    device path must give the same keypoints in the same order, and the same descriptors, for colour and gray frames,
    odd sizes, several quotas (far below and far above the number of corners), cell sizes and low thresholds, a
    frame without a corner, a frame model, and must return to cv::ORB's selection when switched off."""
    import torch
    blank = np.full((480, 640, 3), 90, np.uint8)
    cases = [("vga", [test_images[0], test_images[1], synth_image(61, 640, 480, 3), blank], 1000, 8, 7, 30),
             ("few", [test_images[1], synth_image(62, 640, 480, 3)], 120, 4, 12, 20),
             ("many", [test_images[0], synth_image(63, 640, 480, 3)], 8000, 8, 5, 45),
             ("odd_gray", [synth_image(64, 401, 299), synth_image(65, 401, 299)], 300, 5, 7, 30),
             ("wide", [synth_image(66, 1100, 260, 3)], 600, 3, 9, 25)]
    for name, frames, nkp, nlevels, low, cell in cases:
        h, w = frames[0].shape[:2]
        ch = 1 if frames[0].ndim == 2 else 3
        p = oracle.orb_params(nfeatures=nkp, nlevels=nlevels)
        ex = orbx.Extractor(n_key_points=nkp, level_pyramid=nlevels, max_width=w, max_height=h, max_batch=len(frames))
        dev = torch.from_numpy(np.stack(frames)).cuda()
        torch.cuda.synchronize()
        cap = nkp + 4 * nlevels + 64
        ex.set_distribution(1, low, cell)
        kps, counts, desc = ex.extract_batch(dev.data_ptr(), len(frames), w * h * ch, w, h, w * ch, ch, cap)
        spread, first = [], None
        for f, img in enumerate(frames):
            ok_ = oracle.detect_distributed(p, img, low, cell)
            ok2, od = oracle.compute(p, img, ok_)
            first = ok2 if f == 0 else first
            assert counts[f] == len(ok2), (name, f, counts[f], len(ok2))
            assert np.array_equal(kps[f, :counts[f]], ok2.astype(orbx.KP_DTYPE)), (name, f)
            assert np.array_equal(desc[f, :counts[f]], od), (name, f)
            spread.append(len({(int(x) // 40, int(y) // 40) for x, y in zip(ok2["x"], ok2["y"])}))
        # queued without host outputs and finished later: the same device results; the overflow of a later deferred
        # batch is reported by finish()
        ex.extract_batch(dev.data_ptr(), len(frames), w * h * ch, w, h, w * ch, ch, cap, to_host=False, deferred=True)
        dcounts, redone = ex.finish()
        assert not redone and np.array_equal(dcounts, counts), name
        kp_ptr, desc_ptr, cnt_ptr, kcap = ex.device_results()
        dkp = torch.as_tensor(_DevView(kp_ptr, (len(frames), kcap, 7), "<f4"), device="cuda").cpu().numpy().view(np.uint8)
        dkp = np.ascontiguousarray(dkp).view(orbx.KP_DTYPE).reshape(len(frames), kcap)
        ddesc = torch.as_tensor(_DevView(desc_ptr, (len(frames), kcap, 32), "|u1"), device="cuda").cpu().numpy()
        for f in range(len(frames)):
            assert np.array_equal(dkp[f, :counts[f]], kps[f, :counts[f]]) and np.array_equal(ddesc[f, :counts[f]], desc[f, :counts[f]])
        # one host frame through orbx_detect: a batch of one (so the border filter of the describe stage has run)
        k1 = ex.detect(frames[0])
        assert np.array_equal(k1, first.astype(orbx.KP_DTYPE)), name
        # switched off again: cv::ORB's selection
        ex.set_distribution(0)
        kps0, counts0, desc0 = ex.extract_batch(dev.data_ptr(), len(frames), w * h * ch, w, h, w * ch, ch, cap)
        for f, img in enumerate(frames):
            ok2, od = oracle.compute(p, img, oracle.detect(p, img))
            assert counts0[f] == len(ok2) and np.array_equal(kps0[f, :counts0[f]], ok2.astype(orbx.KP_DTYPE)), (name, f)
            assert np.array_equal(desc0[f, :counts0[f]], od), (name, f)
            if name == "vga" and counts0[f] > 500:
                # the point of the mode: the keypoints cover more of the image
                assert spread[f] > len({(int(x) // 40, int(y) // 40) for x, y in zip(ok2["x"], ok2["y"])}), (name, f)
        ex.close()
    # more corners than the candidate list holds (the dot lattice of test_more_corners_than_the_candidate_list_holds:
    # one corner per 12 pixels).  cv::ORB's selection enlarges the list on the host path; the distribution mode has no
    # host path and must refuse the batch, not truncate it
    dots = np.zeros((480, 640), np.uint8)
    for b in range(480 // 3):
        dots[3 * b, (2 * b) % 4::4] = 255
    ex = orbx.Extractor(n_key_points=500, level_pyramid=1, max_width=640, max_height=480, max_batch=2)
    dev = torch.from_numpy(np.stack([synth_image(67, 640, 480), dots])).cuda()
    torch.cuda.synchronize()
    ex.set_distribution(1, 7, 30)
    with pytest.raises(orbx.OrbxError) as e:
        ex.extract_batch(dev.data_ptr(), 2, 640 * 480, 640, 480, 640, 1, 800)
    assert e.value.status == orbx.UNSUPPORTED
    ex.extract_batch(dev.data_ptr(), 2, 640 * 480, 640, 480, 640, 1, 800, to_host=False, deferred=True)
    with pytest.raises(orbx.OrbxError) as e:
        ex.finish()
    assert e.value.status == orbx.UNSUPPORTED
    ex.close()
    # with the frame model of MonoFrame::extractFeatures (undistortion + bucket grid) on a larger batch
    frames = [test_images[i % 2] if i % 3 == 0 else synth_image(700 + i, 640, 480, 3) for i in range(40)]
    cam = orbx.make_camera(517.3, 516.5, 318.6, 255.3, [0.2624, -0.9531, -0.0054, 0.0026, 1.1633])
    grid = orbx.make_grid((10.8, 626.0, 14.7, 473.3))
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=len(frames))
    ex.set_frame_model(cam, grid)
    ex.set_distribution(1, 7, 30)
    dev = torch.from_numpy(np.stack(frames)).cuda()
    torch.cuda.synchronize()
    kps, counts, desc = ex.extract_batch(dev.data_ptr(), len(frames), 640 * 480 * 3, 640, 480, 640 * 3, 3, 1280)
    p = oracle.orb_params(nfeatures=1000, nlevels=8)
    ocam = oracle.camera(517.3, 516.5, 318.6, 255.3, [0.2624, -0.9531, -0.0054, 0.0026, 1.1633], 640, 480)
    og = oracle.grid_params((10.8, 626.0, 14.7, 473.3))
    for f in (0, 1, 7, 39):
        k = oracle.detect_distributed(p, frames[f], 7, 30)
        xy = oracle.undistort_points(ocam, np.stack([k["x"], k["y"]], 1))
        k["x"], k["y"] = xy[:, 0], xy[:, 1]
        k, start, items = oracle.assign_grid(og, k)
        k, d = oracle.compute(p, frames[f], k)
        assert counts[f] == len(k) and np.array_equal(kps[f, :counts[f]], k.astype(orbx.KP_DTYPE)), f
        assert np.array_equal(desc[f, :counts[f]], d), f
    ex.close()


def test_pyramid_tail_kernel_bit_exact(oracle, test_images):
    """orbx_extractor_set_pyramid_tail: the levels above a base level in one launch (a workgroup per frame reads the
    base level from the atlas and walks the rest through LDS rings).  The atlas must equal the oracle's pyramid,
    padded rectangles included, and the batch extraction the level-by-level result, for every base level, colour and
    gray, odd sizes; a geometry the kernel does not suit silently stays level by level."""
    import torch
    cases = [("vga_color", [test_images[0], test_images[1], synth_image(41, 640, 480, 3)], 8, (1, 2, 3, 4, 5, 6), 8),
             ("vga_rows16", [test_images[1], synth_image(42, 640, 480)[:, :, None].repeat(3, 2)], 8, (2, 3, 4), 16),
             ("odd_gray", [synth_image(43, 401, 299), synth_image(44, 401, 299)], 5, (1, 2, 3), 16),
             ("wide", [synth_image(45, 880, 600, 3)], 6, (1, 3, 4), 8)]
    for name, frames, nlevels, bases, rows in cases:
        h, w = frames[0].shape[:2]
        ch = 1 if frames[0].ndim == 2 else 3
        p = oracle.orb_params(nfeatures=1000, nlevels=nlevels)
        ex = orbx.Extractor(n_key_points=1000, level_pyramid=nlevels, max_width=w, max_height=h, max_batch=len(frames))
        dev = torch.from_numpy(np.stack(frames)).cuda()
        torch.cuda.synchronize()
        k1, c1, d1 = ex.extract_batch(dev.data_ptr(), len(frames), w * h * ch, w, h, w * ch, ch, 2000)
        oatl = [oracle.pyramid(p, img) for img in frames]
        for base in bases:
            ex.set_pyramid_tail(base, rows)
            k2, c2, d2 = ex.extract_batch(dev.data_ptr(), len(frames), w * h * ch, w, h, w * ch, ch, 2000)
            for f in range(len(frames)):
                atlas = ex.read_plane(0, w, h, frame=f)
                oatlas, rects, _ = oatl[f]
                for l, (rx, ry, rw, rh) in enumerate(rects):
                    b = 32
                    assert np.array_equal(atlas[ry - b:ry + rh + b, rx - b:rx + rw + b],
                                          oatlas[ry - b:ry + rh + b, rx - b:rx + rw + b]), (name, base, f, l)
                assert c1[f] == c2[f], (name, base, f)
                assert np.array_equal(k1[f, :c1[f]], k2[f, :c2[f]]) and np.array_equal(d1[f, :c1[f]], d2[f, :c2[f]]), \
                    (name, base, f)
        ex.close()
    # a large batch (more workgroups than CUs) with the setting the bench uses
    frames = [test_images[i % 2] if i % 5 == 0 else synth_image(500 + i, 640, 480, 3) for i in range(130)]
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=len(frames))
    dev = torch.from_numpy(np.stack(frames)).cuda()
    torch.cuda.synchronize()
    k1, c1, d1 = ex.extract_batch(dev.data_ptr(), len(frames), 640 * 480 * 3, 640, 480, 640 * 3, 3, 1280)
    ex.set_pyramid_tail(3, 16)
    k2, c2, d2 = ex.extract_batch(dev.data_ptr(), len(frames), 640 * 480 * 3, 640, 480, 640 * 3, 3, 1280)
    assert np.array_equal(c1, c2)
    for f in range(len(frames)):
        assert np.array_equal(k1[f, :c1[f]], k2[f, :c2[f]]) and np.array_equal(d1[f, :c1[f]], d2[f, :c2[f]]), f
    ex.close()


def test_one_pass_pyramid_kernel_bit_exact(oracle, test_images):
    """The one-pass pyramid kernel (a workgroup per frame, every level from LDS rings, one launch per batch) against
    the oracle's pyramid, padded rectangles included, and the whole batch extraction against the level-by-level
    kernels: colour and gray, odd sizes, several step heights, and a batch of 130 frames."""
    import torch
    cases = [("vga_color", [test_images[0], test_images[1], synth_image(31, 640, 480, 3)], 8, 8),
             ("vga_gray", [test_images[0][:, :, 1].copy(), synth_image(32, 640, 480)], 8, 4),
             ("odd_color", [synth_image(33, 333, 257, 3), synth_image(34, 333, 257, 3)], 4, 6),
             ("odd_gray", [synth_image(35, 401, 299), synth_image(36, 401, 299)], 5, 5),
             ("wide", [synth_image(37, 704, 480, 3)], 8, 8)]
    for name, frames, nlevels, rows in cases:
        h, w = frames[0].shape[:2]
        ch = 1 if frames[0].ndim == 2 else 3
        p = oracle.orb_params(nfeatures=1000, nlevels=nlevels)
        ex = orbx.Extractor(n_key_points=1000, level_pyramid=nlevels, max_width=w, max_height=h, max_batch=len(frames))
        dev = torch.from_numpy(np.stack(frames)).cuda()
        torch.cuda.synchronize()
        res = {}
        for kind in (1, 2):
            ex.set_pyramid_kernel(kind, rows)
            res[kind] = ex.extract_batch(dev.data_ptr(), len(frames), w * h * ch, w, h, w * ch, ch, 2000)
            t = ex.last_timings()
            assert (t["resize"] == 0) == (kind == 2), (name, kind, t)     # kind 2 really took the one-pass kernel
            for f, img in enumerate(frames):
                atlas = ex.read_plane(0, w, h, frame=f)
                oatlas, rects, _ = oracle.pyramid(p, img)
                for l, (rx, ry, rw, rh) in enumerate(rects):
                    b = 32
                    assert np.array_equal(atlas[ry - b:ry + rh + b, rx - b:rx + rw + b],
                                          oatlas[ry - b:ry + rh + b, rx - b:rx + rw + b]), (name, kind, f, l)
        (k1, c1, d1), (k2, c2, d2) = res[1], res[2]
        assert np.array_equal(c1, c2), name
        for f in range(len(frames)):   # slots beyond a frame's count hold whatever an earlier call left there
            assert np.array_equal(k1[f, :c1[f]], k2[f, :c2[f]]) and np.array_equal(d1[f, :c1[f]], d2[f, :c2[f]]), (name, f)
        ex.close()
    # a large batch (more workgroups than CUs): what the oracle gives
    frames = [test_images[i % 2] if i % 5 == 0 else synth_image(400 + i, 640, 480, 3) for i in range(130)]
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=len(frames))
    ex.set_pyramid_kernel(2, 6)
    dev = torch.from_numpy(np.stack(frames)).cuda()
    torch.cuda.synchronize()
    kps, counts, desc = ex.extract_batch(dev.data_ptr(), len(frames), 640 * 480 * 3, 640, 480, 640 * 3, 3, 1500)
    assert ex.last_timings()["resize"] == 0
    p = oracle.orb_params(nfeatures=1000)
    for f in (0, 1, 64, 127, 129):
        ok_ = oracle.detect(p, frames[f])
        ok2, od = oracle.compute(p, frames[f], ok_)
        assert counts[f] == len(ok2) and np.array_equal(kps[f, :counts[f]], ok2.astype(orbx.KP_DTYPE))
        assert np.array_equal(desc[f, :counts[f]], od)
    ex.close()


def test_compute_after_a_failed_batch(oracle, test_images):
    """A batch forks its blur onto a side stream; a batch that fails between the fork and the describe stage (host
    selection, more keypoints than the caller's capacity) must leave the handle clean: the next compute on the same
    handle blurs its own pyramid and matches the oracle bit for bit."""
    import torch
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=2)
    p = oracle.orb_params(nfeatures=1000)
    dev = torch.from_numpy(np.stack([test_images[0], synth_image(71, 640, 480, 3)])).cuda()
    torch.cuda.synchronize()
    other = test_images[1]
    ok = oracle.detect(p, other)
    want_k, want_d = oracle.compute(p, other, ok)
    for device_select in (False, True):
        ex.set_device_select(device_select)
        with pytest.raises(orbx.OrbxError) as e:
            ex.extract_batch(dev.data_ptr(), 2, 640 * 480 * 3, 640, 480, 640 * 3, 3, 400)
        assert e.value.status == orbx.CAPACITY
        k, d = ex.compute(other, ok.astype(orbx.KP_DTYPE))
        assert np.array_equal(k, want_k.astype(orbx.KP_DTYPE)) and np.array_equal(d, want_d), device_select
    ex.close()


def test_failed_frame_leaves_no_keypoints_and_an_empty_grid(oracle, test_images):
    """A deferred batch returns before anybody has looked at the status words, and a matcher may already be queued
    behind it on the stream.  A frame the device selection could not finish (here: more keypoints than the caller's
    capacity) must therefore leave a count of zero and a grid without entries - never the grid an earlier batch left
    in the buffer - and the neighbouring frame of the batch is unaffected."""
    import torch
    cam = orbx.make_camera(517.3, 516.5, 318.6, 255.3, TUM_DIST)
    ocam = oracle.camera(517.3, 516.5, 318.6, 255.3, TUM_DIST, 640, 480)
    grid = orbx.make_grid(oracle.image_bounds(ocam))
    ncell = grid.grid_cols * grid.grid_rows
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=2)
    ex.set_frame_model(cam, grid)
    frames = [test_images[0], np.full_like(test_images[0], 90)]   # ~1000 keypoints, and none
    dev = torch.from_numpy(np.stack(frames)).cuda()
    torch.cuda.synchronize()
    cap = 1280
    kps, counts, desc = ex.extract_batch(dev.data_ptr(), 2, 640 * 480 * 3, 640, 480, 640 * 3, 3, cap)
    assert counts[0] > 900 and counts[1] == 0
    cs, _ = ex.device_grid()
    full = torch.as_tensor(_DevView(cs, (2, ncell + 1), "<i4"), device="cuda").cpu().numpy().copy()
    assert full[0, -1] > 900 and full[1, -1] == 0
    # the same frames with room for 300 keypoints: frame 0 fails, frame 1 (no corners) is fine
    small = 300
    ex.extract_batch(dev.data_ptr(), 2, 640 * 480 * 3, 640, 480, 640 * 3, 3, small, to_host=False, deferred=True)
    torch.cuda.synchronize()
    cs, _ = ex.device_grid()
    _, _, cnt_ptr, kcap = ex.device_results()
    assert kcap == small
    after = torch.as_tensor(_DevView(cs, (2, ncell + 1), "<i4"), device="cuda").cpu().numpy()
    dcnt = torch.as_tensor(_DevView(cnt_ptr, (2,), "<i4"), device="cuda").cpu().numpy()
    assert (after == 0).all() and (dcnt == 0).all()
    with pytest.raises(orbx.OrbxError) as e:
        ex.finish()
    assert e.value.status == orbx.CAPACITY
    # and the handle is as good as new
    kps2, counts2, desc2 = ex.extract_batch(dev.data_ptr(), 2, 640 * 480 * 3, 640, 480, 640 * 3, 3, cap)
    assert np.array_equal(counts2, counts) and np.array_equal(kps2[0, :counts[0]], kps[0, :counts[0]])
    assert np.array_equal(desc2[0, :counts[0]], desc[0, :counts[0]])
    ex.close()


def test_selection_launch_tiers(oracle, test_images):
    """The selection kernel is launched with a small LDS footprint first and full-size for the frames left over; how
    many candidates the small launch holds depends on the keypoint capacity.  Capacities on both sides of every
    switch, with a frame without corners (always taken by the first launch) beside ~4000-candidate frames."""
    import torch
    frames = [test_images[0], np.full_like(test_images[0], 77), test_images[1], synth_image(61, 640, 480, 3)]
    dev = torch.from_numpy(np.stack(frames)).cuda()
    torch.cuda.synchronize()
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=len(frames))
    p = oracle.orb_params(nfeatures=1000)
    ora = [oracle.compute(p, img, oracle.detect(p, img)) for img in frames]
    # 1100/1280: every frame fits the first launch; 1700/2000: the textured frames need the second; 2300: one launch;
    # 4096: more keypoint slots than the selection kernel can hold, the host selection takes over
    for cap in (1100, 1280, 1700, 2000, 2300, 4096):
        kps, counts, desc = ex.extract_batch(dev.data_ptr(), len(frames), 640 * 480 * 3, 640, 480, 640 * 3, 3, cap)
        for f, (ok_, od) in enumerate(ora):
            assert counts[f] == len(ok_), (cap, f)
            assert np.array_equal(kps[f, :counts[f]], ok_.astype(orbx.KP_DTYPE)), (cap, f)
            assert np.array_equal(desc[f, :counts[f]], od), (cap, f)
        assert counts[1] == 0
    ex.close()


def test_selection_of_a_frame_beyond_the_lds_form(oracle, test_images):
    """A VGA frame of pure noise has several times the 8192 candidates the selection kernel keeps in LDS: inside a
    batch it is selected by the kernel's global-memory form, on the device (no hand-over to the host selection),
    while its neighbours take the LDS form; all equal the oracle."""
    import torch
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, (480, 640, 3)).astype(np.uint8)
    frames = [test_images[0], noise, test_images[1], np.ascontiguousarray(noise[::-1])]
    p = oracle.orb_params(nfeatures=1000)
    ora = [oracle.compute(p, img, oracle.detect(p, img)) for img in frames]
    p_all = oracle.orb_params(nfeatures=200000)
    assert len(oracle.detect(p_all, noise)) > 2 * 8192
    dev = torch.from_numpy(np.stack(frames)).cuda()
    torch.cuda.synchronize()
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=len(frames))
    for cap in (1280, 2300):
        ex.extract_batch(dev.data_ptr(), len(frames), 640 * 480 * 3, 640, 480, 640 * 3, 3, cap, to_host=False,
                         deferred=True)
        counts, redone = ex.finish()
        assert not redone
        kps, c2, desc = ex.extract_batch(dev.data_ptr(), len(frames), 640 * 480 * 3, 640, 480, 640 * 3, 3, cap)
        assert np.array_equal(counts, c2)
        for f, (ok_, od) in enumerate(ora):
            assert counts[f] == len(ok_), (cap, f)
            assert np.array_equal(kps[f, :counts[f]], ok_.astype(orbx.KP_DTYPE)), (cap, f)
            assert np.array_equal(desc[f, :counts[f]], od), (cap, f)
    ex.close()
    # more such frames in one batch than the global form has workspace slices (8): the rest get further passes of the
    # same kernel, still on the device, and finish() reports that consumers of the device results must run again
    order = [1, 3] * 5 + [0]
    many = torch.from_numpy(np.stack([frames[i] for i in order])).cuda()
    torch.cuda.synchronize()
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=len(order))
    ex.extract_batch(many.data_ptr(), len(order), 640 * 480 * 3, 640, 480, 640 * 3, 3, 1280, to_host=False, deferred=True)
    counts, redone = ex.finish()
    assert redone
    kps, c2, desc = ex.extract_batch(many.data_ptr(), len(order), 640 * 480 * 3, 640, 480, 640 * 3, 3, 1280)
    assert np.array_equal(counts, c2)
    for f, src in enumerate(order):
        ok_, od = ora[src]
        assert counts[f] == len(ok_), f
        assert np.array_equal(kps[f, :counts[f]], ok_.astype(orbx.KP_DTYPE)), f
        assert np.array_equal(desc[f, :counts[f]], od), f
    ex.close()


def test_extract_batch_ragged_inputs(oracle):
    """Odd frame sizes, padded row strides, gray and colour input, and frames with no or almost no corners inside one
    batch: every frame equals the oracle's detect + compute of that frame."""
    import torch
    for (w, h, ch) in ((333, 217, 1), (501, 363, 3), (64, 64, 3)):
        frames = [synth_image(40 + i, w, h, ch) for i in range(3)]
        flat = np.full_like(frames[0], 90)
        one = flat.copy()
        one[h // 2 - 4:h // 2 + 4, w // 2 - 4:w // 2 + 4] = 250          # a single bright square: a handful of corners
        frames += [flat, one]
        stride = (w * ch + 37) // 16 * 16 + 16                            # rows padded, not a multiple of the width
        buf = np.zeros((len(frames), h, stride), np.uint8)
        for f, img in enumerate(frames):
            buf[f, :, :w * ch] = img.reshape(h, w * ch)
        dev = torch.from_numpy(buf).cuda()
        torch.cuda.synchronize()
        ex = orbx.Extractor(n_key_points=500, level_pyramid=8, max_width=w, max_height=h, max_batch=len(frames))
        p = oracle.orb_params(nfeatures=500)
        kps, counts, desc = ex.extract_batch(dev.data_ptr(), len(frames), h * stride, w, h, stride, ch, 700)
        for f, img in enumerate(frames):
            ok_, od = oracle.compute(p, img, oracle.detect(p, img))
            assert counts[f] == len(ok_), (w, h, ch, f)
            assert np.array_equal(kps[f, :counts[f]], ok_.astype(orbx.KP_DTYPE)), (w, h, ch, f)
            assert np.array_equal(desc[f, :counts[f]], od), (w, h, ch, f)
        assert counts[3] == 0
        ex.close()


def test_match_hamming_bit_exact(oracle, matcher, ex1000, test_images):
    rng = np.random.default_rng(7)
    cases = []
    for nq, nt in ((1, 1), (5, 300), (1000, 1000), (257, 4099), (2000, 77)):
        cases.append((rng.integers(0, 256, (nq, 32), dtype=np.uint8), rng.integers(0, 256, (nt, 32), dtype=np.uint8)))
    # ties: duplicated train rows and few distinct values -> the lowest train index must win
    t = rng.integers(0, 2, (600, 32), dtype=np.uint8) * 255
    cases.append((t[rng.integers(0, 600, 900)].copy(), np.concatenate([t, t])))
    r = [ex1000.compute(img, ex1000.detect(img)) for img in test_images]
    cases.append((r[0][1], r[1][1]))
    # sizes around the 32-row tiles and the 64- / 256-query blocks of the matrix-core kernel; all-zero and all-one rows
    # (distances 0 and 256, the ends of the dot product's range)
    for nq, nt in ((31, 33), (32, 32), (33, 31), (63, 65), (64, 64), (65, 63), (255, 1), (256, 2), (300, 97)):
        cases.append((rng.integers(0, 256, (nq, 32), dtype=np.uint8), rng.integers(0, 256, (nt, 32), dtype=np.uint8)))
    ends = np.concatenate([np.zeros((3, 32), np.uint8), np.full((3, 32), 255, np.uint8),
                           rng.integers(0, 256, (70, 32), dtype=np.uint8)])
    cases.append((ends, ends[::-1].copy()))
    # more train rows than one chunk's 16-bit row index covers; the best row of some queries sits in the second chunk
    big = rng.integers(0, 256, (70000, 32), dtype=np.uint8)
    cases.append((np.concatenate([big[[69999, 65536, 65535, 12]], rng.integers(0, 256, (36, 32), dtype=np.uint8)]), big))
    for kind in (0, 1):   # matrix cores, vector pipes
        matcher.set_hamming_kernel(kind)
        for q, tr in cases:
            idx, dist = matcher.match_hamming(q, tr)
            oidx, odist = oracle.match_hamming(q, tr)
            assert np.array_equal(idx, oidx) and np.array_equal(dist, odist), (kind, q.shape, tr.shape)
    matcher.set_hamming_kernel(0)
    idx, dist = matcher.match_hamming(cases[2][0], np.zeros((0, 32), np.uint8))
    assert np.all(idx == -1) and np.all(dist == -1)
    # ORBMatcher::match(..., filter=true) end to end
    m = matcher.match(r[0][1], r[1][1], True)
    om = oracle.filter_matches(oracle.matches_from(*oracle.match_hamming(r[0][1], r[1][1])))
    assert np.array_equal(m, om.astype(orbx.MATCH_DTYPE))
    assert len(set(m["train"][1:].tolist())) == len(m) - 1


def test_match_hamming_batch_ragged_pairs(oracle, matcher):
    """orbx_match_hamming_batch on several pairs of ragged sizes in one call (empty query or train sets, sizes around
    the 256-row tiles and the 512-query blocks, many equal rows: ties must go to the lowest train index), against the
    oracle pair by pair; then the key-frame database driver (one call per frame, the query block shared by all pairs)
    against its CPU stand-in."""
    import torch
    from orb_slam_amd import kfdb
    rng = np.random.default_rng(61)
    sizes = [(0, 10), (10, 0), (1, 1), (255, 257), (256, 256), (257, 255), (511, 513), (512, 512), (513, 1000),
             (1000, 999), (300, 2)]
    cap_q, cap_t = 1000, 1000
    P = len(sizes)
    q = np.zeros((P, cap_q, 32), np.uint8)
    t = np.zeros((P, cap_t, 32), np.uint8)
    for p, (nq, nt) in enumerate(sizes):
        t[p, :nt] = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        if nt:
            t[p, :nt:3] = t[p, 0]                                    # every third train row identical: ties
            q[p, :nq] = t[p, rng.integers(0, nt, nq)] ^ (rng.integers(0, 256, (nq, 32), dtype=np.uint8) & 0x05)
        else:
            q[p, :nq] = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    d_q, d_t = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    d_nq = torch.tensor([a for a, _ in sizes], dtype=torch.int32, device="cuda")
    d_nt = torch.tensor([b for _, b in sizes], dtype=torch.int32, device="cuda")
    d_idx = torch.full((P, cap_q), -7, dtype=torch.int32, device="cuda")
    d_dist = torch.full((P, cap_q), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    matcher.match_hamming_batch(d_q.data_ptr(), d_nq.data_ptr(), cap_q * 32, d_t.data_ptr(), d_nt.data_ptr(),
                                cap_t * 32, P, cap_q, cap_t, d_idx.data_ptr(), d_dist.data_ptr(), cap_q)
    torch.cuda.synchronize()
    idx, dist = d_idx.cpu().numpy(), d_dist.cpu().numpy()
    for p, (nq, nt) in enumerate(sizes):
        if nt == 0:
            assert (idx[p, :nq] == -1).all()
        else:
            oi, od = oracle.match_hamming(q[p, :nq], t[p, :nt])
            assert np.array_equal(idx[p, :nq], oi) and np.array_equal(dist[p, :nq], od), (p, nq, nt)
        assert (idx[p, nq:] == -7).all()                              # rows past a pair's count are not written
    # counts ABOVE the capacities the call names (they are device data the call cannot look at): every count is
    # read as min(count, capacity), so no result lands in a neighbouring pair's rows or past the buffers
    over_q = torch.tensor([cap_q + 500 if i % 2 else a for i, (a, _) in enumerate(sizes)], dtype=torch.int32, device="cuda")
    over_t = torch.tensor([cap_t + 77 if i % 3 == 0 else b for i, (_, b) in enumerate(sizes)], dtype=torch.int32, device="cuda")
    d_idx.fill_(-7)
    d_dist.fill_(-7)
    guard = torch.full((4096,), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    matcher.match_hamming_batch(d_q.data_ptr(), over_q.data_ptr(), cap_q * 32, d_t.data_ptr(), over_t.data_ptr(),
                                cap_t * 32, P, cap_q, cap_t, d_idx.data_ptr(), d_dist.data_ptr(), cap_q)
    torch.cuda.synchronize()
    idx2, dist2 = d_idx.cpu().numpy(), d_dist.cpu().numpy()
    for p_, (nq, nt) in enumerate(sizes):
        nq2 = cap_q if p_ % 2 else nq
        nt2 = cap_t if p_ % 3 == 0 else nt
        if nt2 == 0:
            assert (idx2[p_, :nq2] == -1).all()
        else:
            oi, od = oracle.match_hamming(q[p_, :nq2], t[p_, :nt2])
            assert np.array_equal(idx2[p_, :nq2], oi) and np.array_equal(dist2[p_, :nq2], od), (p_, nq2, nt2)
        assert (idx2[p_, nq2:] == -7).all()
    assert (guard.cpu().numpy() == -7).all()
    # every query set against every train set in one call (pair q * n_train_sets + t), sets 3..8 against sets 2..9
    QS, TS = list(range(3, 9)), list(range(2, 10))
    c_idx = torch.full((len(QS) * len(TS), cap_q), -7, dtype=torch.int32, device="cuda")
    c_dist = torch.full_like(c_idx, -7)
    matcher.match_hamming_cross(d_q[3:].data_ptr(), d_nq[3:].data_ptr(), cap_q * 32, len(QS), d_t[2:].data_ptr(),
                                d_nt[2:].data_ptr(), cap_t * 32, len(TS), cap_q, cap_t, c_idx.data_ptr(),
                                c_dist.data_ptr(), cap_q)
    torch.cuda.synchronize()
    ci, cd = c_idx.cpu().numpy(), c_dist.cpu().numpy()
    for a, qs in enumerate(QS):
        for b, tset in enumerate(TS):
            nq, nt = sizes[qs][0], sizes[tset][1]
            oi, od = oracle.match_hamming(q[qs, :nq], t[tset, :nt])
            assert np.array_equal(ci[a * len(TS) + b, :nq], oi) and np.array_equal(cd[a * len(TS) + b, :nq], od), (qs, tset)
            assert (ci[a * len(TS) + b, nq:] == -7).all()
    # the key-frame database driver: 6 frames against 4 key frames
    desc = torch.from_numpy(q[3:9, :600].copy())
    counts = torch.tensor([255, 256, 257, 511, 512, 513], dtype=torch.int32)
    db_desc, db_counts = kfdb.gather_database(desc, counts, [0, 2, 3, 5])
    want_i, want_d = kfdb.match_frames_against_database(kfdb.NumpyBackend(), desc, counts, db_desc, db_counts)
    ts = torch.cuda.Stream()   # torch's default stream has handle 0, which set_stream reads as "the handle's own"
    with torch.cuda.stream(ts):
        matcher.set_stream(ts.cuda_stream)
        got_i, got_d = kfdb.match_frames_against_database(kfdb.HipBackend(matcher), desc.cuda(), counts.cuda(),
                                                          db_desc.cuda(), db_counts.cuda(), block=4)
    torch.cuda.synchronize()
    matcher.set_stream(0)
    for f in range(6):
        n = int(counts[f])
        assert torch.equal(got_i[f, :, :n].cpu(), want_i[f, :, :n]) and torch.equal(got_d[f, :, :n].cpu(), want_d[f, :, :n])


def test_match_radius_bit_exact(oracle, matcher, ex8000, test_images):
    r = [ex8000.compute(img, ex8000.detect(img)) for img in test_images]
    for radius in (100.0, 12.5, 0.0):
        m = matcher.match_radius(r[0][0], r[1][0], r[0][1], r[1][1], radius)
        om = oracle.match_radius(r[0][0], r[1][0], r[0][1], r[1][1], radius)
        assert np.array_equal(m, om.astype(orbx.MATCH_DTYPE)), radius
    m = matcher.match_radius(r[0][0], r[1][0], r[0][1], r[1][1], 100.0)
    assert len(m) >= 4500  # src/tests/geometry/orb_matcher_tests.cpp:39


def _projection_case(oracle, ex, img_a, img_b, rng, bounds):
    p = oracle.orb_params(nfeatures=1000)
    ka, da = ex.compute(img_a, ex.detect(img_a))
    kb, db = ex.compute(img_b, ex.detect(img_b))
    g = orbx.make_grid(bounds)
    ka, start, items = orbx.assign_grid(g, ka)
    da = da[:len(ka)] if len(da) == len(ka) else None
    return g, ka, start, items, kb, db


def test_projection_matchers_bit_exact(oracle, matcher, ex1000, test_images):
    rng = np.random.default_rng(9)
    bounds = (10.803785, 626.0599, 14.68408, 473.3248)
    g = orbx.make_grid(bounds)
    og = oracle.grid_params(bounds)
    det = [ex1000.detect(img) for img in test_images]
    k0, start, items = orbx.assign_grid(g, det[0])
    k0, d0 = ex1000.compute(test_images[0], k0)
    kr, dr = ex1000.compute(test_images[1], det[1])
    _, sf, _, _, _ = ex1000.scale_tables()
    n_ref = len(kr)
    proj = np.stack([kr["x"], kr["y"]], 1) + rng.normal(0, 4, (n_ref, 2)).astype(np.float32)
    proj[:7] = [[-3, 5], [700, 100], [320, -2], [320, 500], [11, 15], [626, 473], [5000, 5000]]
    z = np.ones(n_ref, np.float32)
    z[rng.integers(0, n_ref, 20)] = -1
    has = (rng.uniform(size=n_ref) > 0.1).astype(np.uint8)
    has[0] = 1
    for level_mode in (0, 1, 2):
        for radius, orient in ((15.0, True), (30.0, False), (7.0, True)):
            m = matcher.match_projection_frame(g, start, items, k0, d0, kr, has, proj, z, dr, sf, level_mode,
                                               radius, orient)
            om = oracle.match_projection_frame(og, start, items, k0, d0, kr, has, proj, z, dr, sf, level_mode,
                                               radius, orient)
            assert np.array_equal(m, om.astype(orbx.MATCH_DTYPE)), (level_mode, radius, orient)
            if level_mode == 0 and radius == 15.0:
                assert len(m) > 50
    # map-point variant
    in_view = has.copy()
    pred = np.clip(kr["octave"] + rng.integers(-1, 2, n_ref), 0, 7).astype(np.int32)
    vcos = rng.uniform(0.99, 1.0, n_ref).astype(np.float32)
    for radius, ratio in ((1.0, 0.6), (3.0, 0.8), (5.0, 0.95)):
        m = matcher.match_projection_points(g, start, items, k0, d0, in_view, proj, pred, vcos, dr, sf, radius, ratio)
        om = oracle.match_projection_points(og, start, items, k0, d0, in_view, proj, pred, vcos, dr, sf, radius, ratio)
        assert np.array_equal(m, om.astype(orbx.MATCH_DTYPE)), (radius, ratio)
    assert len(m) > 20


class _DevView:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}


def _two_view(rng, n, outliers, noise=0.5, planar=False):
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n),
                  np.full(n, 4.0) if planar else rng.uniform(2, 6, n)], 1)
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


def test_ransac_per_hypothesis_bit_exact(oracle, ransac, utils_points):
    rng = np.random.default_rng(21)
    cases = [_two_view(rng, 300, 90), _two_view(rng, 2000, 600), _two_view(rng, 150, 30, planar=True),
             (utils_points["points"], utils_points["ref_points"]), _two_view(rng, 8, 0), _two_view(rng, 2500, 500)]
    for ci, (pts, ref) in enumerate(cases):
        it = 200 if len(pts) > 8 else 3
        sets = orbx.ransac_make_sets(len(pts), it, 100 + ci)
        F, H, Hi, sf, sh = ransac.fh_all(pts, ref, sets, 1.0)
        oF, oH, oHi, osf, osh = oracle.ransac_fh_all(pts, ref, sets, 1.0)
        for name, a, b in (("F", F, oF), ("H", H, oH), ("Hinv", Hi, oHi), ("score_f", sf, osf), ("score_h", sh, osh)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (ci, name, np.abs(a - b).max())


def test_ransac_degenerate_geometry_bit_exact(oracle, ransac):
    """Rank-deficient eight-point systems (collinear points, four distinct points repeated, identical views, integer
    lattices, a single point): the vanishing singular values take cv::SVDecomp's completion path, divisions by zero
    produce inf / NaN - all of it bit for bit as the oracle, few iterations (the per-pair scoring path) included."""
    rng = np.random.default_rng(29)
    x = rng.uniform(0, 640, 60).astype(np.float32)
    collinear = np.stack([x, 0.5 * x + 20], 1).astype(np.float32)
    four = np.tile(np.array([[10, 10], [600, 20], [30, 400], [500, 450]], np.float32), (10, 1))
    lattice = np.stack(np.meshgrid(np.arange(8) * 40.0, np.arange(8) * 30.0), -1).reshape(-1, 2).astype(np.float32)
    generic = rng.uniform(0, 480, (50, 2)).astype(np.float32)
    cases = [(collinear, collinear[::-1].copy()), (four, four + 1), (lattice, lattice), (lattice, lattice * 2 + 3),
             (generic, generic), (np.full((12, 2), 7, np.float32), generic[:12]), (collinear, generic[:60] * 1e4)]
    for ci, (pts, ref) in enumerate(cases):
        for it in (5, 100):
            sets = orbx.ransac_make_sets(len(pts), it, 200 + ci)
            F, H, Hi, sf, sh = ransac.fh_all(pts, ref, sets, 1.0)
            oF, oH, oHi, osf, osh = oracle.ransac_fh_all(pts, ref, sets, 1.0)
            for name, a, b in (("F", F, oF), ("H", H, oH), ("Hinv", Hi, oHi), ("score_f", sf, osf), ("score_h", sh, osh)):
                # a NaN is a NaN: x86 and gfx950 give their default NaNs different sign bits
                nan = np.isnan(a)
                assert np.array_equal(nan, np.isnan(b)), (ci, it, name)
                assert np.array_equal(a.view(np.uint32)[~nan], b.view(np.uint32)[~nan]), (ci, it, name)
            r = ransac.fh(pts, ref, sets, 1.0)
            o = oracle.ransac_fh(pts, ref, sets, 1.0)
            assert np.array_equal(r["best_it"], o["best_it"]), (ci, it)
            assert np.array_equal(r["inliers_f"], o["inliers_f"]) and np.array_equal(r["inliers_h"], o["inliers_h"])


def test_ransac_result_identical(oracle, ransac, utils_points):
    rng = np.random.default_rng(22)
    cases = [_two_view(rng, 500, 150), _two_view(rng, 1200, 300, planar=True),
             (utils_points["points"], utils_points["ref_points"])]
    for ci, (pts, ref) in enumerate(cases):
        for sigma in (1.0, 2.0):
            sets = orbx.ransac_make_sets(len(pts), 500, 7 + ci)
            r = ransac.fh(pts, ref, sets, sigma)
            o = oracle.ransac_fh(pts, ref, sets, sigma)
            assert np.array_equal(r["best_it"], o["best_it"])
            assert np.array_equal(r["F"], o["F"]) and np.array_equal(r["H"], o["H"])
            assert np.array_equal(r["inliers_f"], o["inliers_f"]) and np.array_equal(r["inliers_h"], o["inliers_h"])
            assert r["score_f"] == o["score_f"] and r["score_h"] == o["score_h"]  # well inside the 1e-5 budget
            assert r["inliers_f"].sum() > 0.5 * len(pts) or ci == 1
    # degenerate input: every hypothesis scores 0 -> no winner, matrices untouched
    pts = np.zeros((20, 2), np.float32)
    ref = np.zeros((20, 2), np.float32)
    sets = orbx.ransac_make_sets(20, 10, 1)
    r = ransac.fh(pts, ref, sets, 1.0)
    o = oracle.ransac_fh(pts, ref, sets, 1.0)
    assert np.array_equal(r["best_it"], o["best_it"])
    with pytest.raises(orbx.OrbxError):
        ransac.fh(pts[:5], ref[:5], sets, 1.0)
    bad = sets.copy()
    bad[0, 0] = 999
    with pytest.raises(orbx.OrbxError):
        ransac.fh(pts, ref, bad, 1.0)


def _same_bits(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    nan = np.isnan(a)
    return np.array_equal(nan, np.isnan(b)) and np.array_equal(a.view(np.uint32)[~nan], b.view(np.uint32)[~nan])


def test_public_solvers_reference_kat_through_hip(oracle, ransac, utils_points):
    """src/tests/geometry/utils_tests.cpp:71-121 with the three public solvers of geometry/utils.h running on the GPU:
    the reference's own tolerances against its golden matrices (79 pairs, the m >= n path of cv::SVDecomp), and bit
    equality with the oracle."""
    pts, ref = utils_points["points"], utils_points["ref_points"]
    pn, T = ransac.normalize_points(pts)
    rn, rT = ransac.normalize_points(ref)
    opn, oT = oracle.normalize_points(pts)
    orn, orT = oracle.normalize_points(ref)
    assert _same_bits(pn, opn) and _same_bits(T, oT) and _same_bits(rn, orn) and _same_bits(rT, orT)
    Fn = ransac.fundamental_npt(pn, rn)
    assert _same_bits(Fn, oracle.fundamental_8pt(pn, rn))
    F = T.T.astype(np.float64) @ Fn.astype(np.float64) @ rT.astype(np.float64)
    assert np.linalg.norm(F / F[2, 2] - utils_points["F_gold"]) <= 1e-3          # utils_tests.cpp:97
    Hn = ransac.homography_dlt(pn, rn)
    assert _same_bits(Hn, oracle.homography_dlt(pn, rn))
    H = np.linalg.inv(T.astype(np.float64)) @ Hn.astype(np.float64) @ rT.astype(np.float64)
    assert np.linalg.norm(H / H[2, 2] - utils_points["H_gold"]) <= 1.0           # utils_tests.cpp:120


def test_public_solvers_any_point_count_bit_exact(oracle, ransac):
    """computeFundamentalMat / computeHomographyMat / normalizePoints for every shape of system cv::SVDecomp
    distinguishes: fewer equations than unknowns (completion by the seeded Gram-Schmidt), exactly 8 and 9, many (more
    than one LDS chunk of products), rank-deficient inputs, and several systems per call."""
    rng = np.random.default_rng(31)
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 10, 33, 79, 300, 1000):
        pts, ref = _two_view(rng, max(n, 8), 0)
        pts, ref = pts[:n].copy(), ref[:n].copy()
        pn, T = ransac.normalize_points(pts)
        opn, oT = oracle.normalize_points(pts)
        assert _same_bits(pn, opn) and _same_bits(T, oT), n
        rn, _ = oracle.normalize_points(ref)
        if n == 1:  # a single point normalises to NaN (0 / 0): feed the solvers the raw points instead
            opn, rn = pts, ref
        assert _same_bits(ransac.fundamental_npt(opn, rn), oracle.fundamental_8pt(opn, rn)), n
        assert _same_bits(ransac.homography_dlt(opn, rn), oracle.homography_dlt(opn, rn)), n
    x = rng.uniform(0, 640, 40).astype(np.float32)
    collinear = np.stack([x, 0.5 * x + 20], 1).astype(np.float32)
    lattice = np.stack(np.meshgrid(np.arange(6) * 40.0, np.arange(6) * 30.0), -1).reshape(-1, 2).astype(np.float32)
    for pts, ref in ((collinear, collinear[::-1].copy()), (lattice, lattice), (lattice, lattice * 2 + 3),
                     (np.zeros((12, 2), np.float32), np.zeros((12, 2), np.float32))):
        assert _same_bits(ransac.fundamental_npt(pts, ref), oracle.fundamental_8pt(pts, ref))
        assert _same_bits(ransac.homography_dlt(pts, ref), oracle.homography_dlt(pts, ref))
    many = [_two_view(rng, 50, 5) for _ in range(70)]
    P = np.stack([m[0] for m in many])
    R = np.stack([m[1] for m in many])
    F = ransac.fundamental_npt(P, R)
    H = ransac.homography_dlt(P, R)
    for q in range(len(many)):
        assert _same_bits(F[q], oracle.fundamental_8pt(P[q], R[q])), q
        assert _same_bits(H[q], oracle.homography_dlt(P[q], R[q])), q
    with pytest.raises(orbx.OrbxError):
        ransac.fundamental_npt(np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))


def test_ransac_batch_equals_single(oracle, ransac):
    import torch
    rng = np.random.default_rng(23)
    P, n, it = 5, 700, 200
    data = [_two_view(rng, n, 200) for _ in range(P)]
    pts = np.ascontiguousarray(np.stack([d[0] for d in data]))
    ref = np.ascontiguousarray(np.stack([d[1] for d in data]))
    sets = np.stack([orbx.ransac_make_sets(n, it, 50 + i) for i in range(P)])
    d_pts, d_ref = torch.from_numpy(pts).cuda(), torch.from_numpy(ref).cuda()
    d_sets = torch.from_numpy(sets.astype(np.int64)).to(torch.int32).cuda()
    d_F = torch.zeros((P, 9), dtype=torch.float32, device="cuda")
    d_H = torch.zeros_like(d_F)
    d_if = torch.zeros((P, n), dtype=torch.uint8, device="cuda")
    d_ih = torch.zeros_like(d_if)
    d_sc = torch.zeros((P, 2), dtype=torch.float32, device="cuda")
    d_best = torch.zeros((P, 2), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()  # the handle launches on its own stream: inputs must have landed
    ransac.fh_batch(d_pts.data_ptr(), d_ref.data_ptr(), n, d_sets.data_ptr(), it, P, 1.0, d_F.data_ptr(),
                    d_H.data_ptr(), d_if.data_ptr(), d_ih.data_ptr(), d_sc.data_ptr(), d_best.data_ptr())
    torch.cuda.synchronize()
    for p in range(P):
        o = oracle.ransac_fh(pts[p], ref[p], sets[p], 1.0)
        assert np.array_equal(d_best[p].cpu().numpy(), o["best_it"])
        assert np.array_equal(d_F[p].cpu().numpy().reshape(3, 3), o["F"])
        assert np.array_equal(d_H[p].cpu().numpy().reshape(3, 3), o["H"])
        assert np.array_equal(d_if[p].cpu().numpy(), o["inliers_f"])
        assert np.array_equal(d_ih[p].cpu().numpy(), o["inliers_h"])
        assert d_sc[p, 0].item() == o["score_f"] and d_sc[p, 1].item() == o["score_h"]
    # device-resident sets cannot be inspected by the call: an index that names no correspondence is never used as
    # an address, its pair comes back as invalid (-2, score 0, no inliers, matrices untouched), the others as before
    bad = sets.copy()
    bad[2, 17, 5] = n
    bad[2, 100, 0] = 0xFFFFFFF0
    d_bad = torch.from_numpy(bad.astype(np.int64)).to(torch.int32).cuda()
    want = [t.clone() for t in (d_F, d_H, d_if, d_ih, d_sc, d_best)]
    d_F[2] = -5.0
    d_H[2] = -5.0
    torch.cuda.synchronize()
    ransac.fh_batch(d_pts.data_ptr(), d_ref.data_ptr(), n, d_bad.data_ptr(), it, P, 1.0, d_F.data_ptr(),
                    d_H.data_ptr(), d_if.data_ptr(), d_ih.data_ptr(), d_sc.data_ptr(), d_best.data_ptr())
    torch.cuda.synchronize()
    keep = [p for p in range(P) if p != 2]
    for got, exp in zip((d_F, d_H, d_if, d_ih, d_sc, d_best), want):
        assert torch.equal(got[keep], exp[keep])
    assert (d_best[2] == -2).all() and (d_sc[2] == 0).all() and not d_if[2].any() and not d_ih[2].any()
    assert (d_F[2] == -5.0).all() and (d_H[2] == -5.0).all()


TUM_DIST = [0.2624, -0.9531, -0.0054, 0.0026, 1.1633]
# a lens that moves points outward: cv::ORB::compute then drops keypoints near the border AFTER the grid was built
# (mono_frame.cpp:97-103), so the grid keeps entries for slots past the end of the shortened keypoint list
OUTWARD_DIST = [-0.25, 0.05, 0.001, -0.002, 0.0]


@pytest.mark.parametrize("cap,dist", [(1280, TUM_DIST), (6500, TUM_DIST), (1280, OUTWARD_DIST), (6500, OUTWARD_DIST)],
                         ids=["tum-1280", "tum-6500", "outward-1280", "outward-6500"])
def test_frame_model_batch_and_batched_projection(oracle, test_images, cap, dist):
    """orbx_extract_batch with the MonoFrame::extractFeatures frame model (undistort + grid + compute) and the
    batched projection matcher on its device results, against the oracle run frame by frame.  The larger row
    capacity takes the host selection and the matcher's sequential resolve (their LDS-resident forms stop earlier).
    The batch runs twice on the handle (host selection, then device selection), so the second run meets whatever the
    first left in the keypoint buffer beyond each frame's count."""
    import torch
    B = 5
    cam = orbx.make_camera(517.3, 516.5, 318.6, 255.3, dist)
    ocam = oracle.camera(517.3, 516.5, 318.6, 255.3, dist, 640, 480)
    bounds = oracle.image_bounds(ocam)
    grid = orbx.make_grid(bounds)
    og = oracle.grid_params(bounds)
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=B)
    ex.set_frame_model(cam, grid)
    mt = orbx.Matcher()
    p = oracle.orb_params(nfeatures=1000)
    rng = np.random.default_rng(31)
    frames = [test_images[0], test_images[1], np.roll(test_images[0], 3, axis=1).copy(),
              np.roll(test_images[1], -2, axis=0).copy(), test_images[0]]
    dev = torch.from_numpy(np.stack(frames)).cuda()
    torch.cuda.synchronize()
    ex.set_device_select(False)
    kps_h, counts_h, desc_h = ex.extract_batch(dev.data_ptr(), B, 640 * 480 * 3, 640, 480, 640 * 3, 3, cap)
    import ctypes
    cs_h, ci_h = ex.device_grid()
    ncell = grid.grid_cols * grid.grid_rows
    grid_host = (torch.as_tensor(_DevView(cs_h, (B, ncell + 1), "<i4"), device="cuda").cpu().numpy().copy(),
                 torch.as_tensor(_DevView(ci_h, (B, cap), "<i4"), device="cuda").cpu().numpy().copy())
    ex.set_device_select(True)
    kps, counts, desc = ex.extract_batch(dev.data_ptr(), B, 640 * 480 * 3, 640, 480, 640 * 3, 3, cap)
    cs_d, ci_d = ex.device_grid()
    grid_dev = (torch.as_tensor(_DevView(cs_d, (B, ncell + 1), "<i4"), device="cuda").cpu().numpy(),
                torch.as_tensor(_DevView(ci_d, (B, cap), "<i4"), device="cuda").cpu().numpy())
    assert np.array_equal(counts, counts_h) and np.array_equal(grid_dev[0], grid_host[0])
    for f in range(B):
        assert np.array_equal(kps[f, :counts[f]], kps_h[f, :counts[f]]) and np.array_equal(desc[f, :counts[f]], desc_h[f, :counts[f]])
        nk = grid_dev[0][f, -1]
        assert np.array_equal(grid_dev[1][f, :nk], grid_host[1][f, :nk])
    ora = []
    for f, img in enumerate(frames):
        k = oracle.detect(p, img)
        xy = oracle.undistort_points(ocam, np.stack([k["x"], k["y"]], 1))
        k["x"], k["y"] = xy[:, 0], xy[:, 1]
        k, start, items = oracle.assign_grid(og, k)
        k2, d = oracle.compute(p, img, k)
        if dist is OUTWARD_DIST:
            assert len(k2) < len(k) and items.max() >= len(k2)   # the case this camera is here for
        assert counts[f] == len(k2)
        assert np.array_equal(kps[f, :counts[f]], k2.astype(orbx.KP_DTYPE)) and np.array_equal(desc[f, :counts[f]], d)
        assert np.array_equal(orbx.undistort_points(cam, np.stack([k["x"], k["y"]], 1)),
                              oracle.undistort_points(ocam, np.stack([k["x"], k["y"]], 1)))
        ora.append((k2, d, start, items))
    _, sf, _, _, _ = ex.scale_tables()
    pairs = [(1, 0), (2, 1), (3, 2), (4, 3), (0, 4), (4, 0)]
    P = len(pairs)
    proj = np.zeros((P, cap, 2), np.float32)
    for i, (f, r) in enumerate(pairs):
        rk = ora[r][0]
        proj[i, :len(rk), 0] = rk["x"] + rng.normal(0, 3, len(rk)).astype(np.float32)
        proj[i, :len(rk), 1] = rk["y"] + rng.normal(0, 3, len(rk)).astype(np.float32)
    d_fi = torch.tensor([a for a, _ in pairs], dtype=torch.int32, device="cuda")
    d_ri = torch.tensor([b for _, b in pairs], dtype=torch.int32, device="cuda")
    d_proj = torch.from_numpy(proj).cuda()
    d_matched = torch.zeros((P, cap), dtype=torch.int32, device="cuda")
    d_dist = torch.zeros((P, cap), dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(P, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    kp_ptr, desc_ptr, cnt_ptr, kcap = ex.device_results()
    cs_ptr, ci_ptr = ex.device_grid()
    assert kcap == cap
    # radius 40 and 90 make windows longer than the pre-gathered lists hold; level modes 1 and 2 are the
    # "in front" / "behind" window rules
    for radius, orient, level_mode in ((15.0, True, 0), (40.0, False, 0), (25.0, True, 1), (25.0, False, 2),
                                       (90.0, True, 0)):
        mt.match_projection_frame_batch(grid, P, d_fi.data_ptr(), d_ri.data_ptr(), kp_ptr, desc_ptr, cnt_ptr, cap,
                                        cs_ptr, ci_ptr, 0, d_proj.data_ptr(), 0, sf, level_mode, radius, orient,
                                        d_matched.data_ptr(), d_dist.data_ptr(), d_nm.data_ptr())
        torch.cuda.synchronize()
        hm, hd, hn = d_matched.cpu().numpy(), d_dist.cpu().numpy(), d_nm.cpu().numpy()
        for i, (f, r) in enumerate(pairs):
            fk, fd, start, items = ora[f]
            rk, rd = ora[r][0], ora[r][1]
            om = oracle.match_projection_frame(og, start, items, fk, fd, rk, np.ones(len(rk), np.uint8),
                                               proj[i, :len(rk)], np.ones(len(rk), np.float32), rd, sf, level_mode,
                                               radius, orient)
            got_q = np.nonzero(hm[i, :len(fk)] > 0)[0]
            assert hn[i] == len(om) == len(got_q), (i, radius, level_mode)
            assert np.array_equal(got_q, om["query"]) and np.array_equal(hm[i, got_q], om["train"])
            assert np.array_equal(hd[i, got_q].astype(np.float32), om["distance"])
        if level_mode != 1:  # the 4-argument window test compares pixels with cell indices: it accepts nothing here
            # (with the outward lens the grid's entries have shifted against the shortened keypoint list, as in the
            # reference: windows then hold the wrong features and fewer matches survive)
            assert hn.min() > (100 if dist is TUM_DIST else 20)
    ex.close()
    mt.close()


def test_batched_map_point_projection(oracle, test_images):
    """orbx_match_projection_points_batch: each pair matches its own list of map points (ragged lengths, some not in
    view, radii large enough for windows of more than 64 candidates) against one frame of an extracted batch; every
    pair equals the oracle's BruteForceWithProjectionMatcher::match(frame, points) on the same data."""
    import torch
    B, cap, pcap = 4, 1280, 1400
    cam = orbx.make_camera(517.3, 516.5, 318.6, 255.3, TUM_DIST)
    ocam = oracle.camera(517.3, 516.5, 318.6, 255.3, TUM_DIST, 640, 480)
    bounds = oracle.image_bounds(ocam)
    grid, og = orbx.make_grid(bounds), oracle.grid_params(bounds)
    frames = [test_images[0], test_images[1], np.roll(test_images[0], 4, axis=1).copy(), synth_image(55, 640, 480, 3)]
    p = oracle.orb_params(nfeatures=1000)
    ora = []
    for img in frames:
        k = oracle.detect(p, img)
        xy = oracle.undistort_points(ocam, np.stack([k["x"], k["y"]], 1))
        k["x"], k["y"] = xy[:, 0], xy[:, 1]
        k, start, items = oracle.assign_grid(og, k)
        k2, d = oracle.compute(p, img, k)
        ora.append((k2, d, start, items))
    rng = np.random.default_rng(45)
    pairs = [0, 1, 2, 3, 0, 2]                       # frame of each pair
    n_pts = [1000, 1400, 37, 0, 640, 999]
    P = len(pairs)
    in_view = np.zeros((P, pcap), np.uint8)
    proj = np.zeros((P, pcap, 2), np.float32)
    pred = np.zeros((P, pcap), np.int32)
    vcos = np.zeros((P, pcap), np.float32)
    pdesc = np.zeros((P, pcap, 32), np.uint8)
    for i, (f, n) in enumerate(zip(pairs, n_pts)):
        src_k, src_d = ora[(f + 1) % 4][0], ora[(f + 1) % 4][1]     # points "seen" in the next frame
        pick = rng.integers(0, len(src_k), n)
        in_view[i, :n] = rng.uniform(size=n) > 0.15
        proj[i, :n, 0] = src_k["x"][pick] + rng.normal(0, 3, n)
        proj[i, :n, 1] = src_k["y"][pick] + rng.normal(0, 3, n)
        pred[i, :n] = np.clip(src_k["octave"][pick] + rng.integers(-1, 2, n), 0, 7)
        vcos[i, :n] = rng.uniform(0.99, 1.0, n)
        pdesc[i, :n] = src_d[pick] ^ (rng.integers(0, 256, (n, 32), dtype=np.uint8) & 0x10)
    dev = torch.from_numpy(np.stack(frames)).cuda()
    torch.cuda.synchronize()
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=B)
    ex.set_frame_model(cam, grid)
    mt = orbx.Matcher()
    kps, counts, desc = ex.extract_batch(dev.data_ptr(), B, 640 * 480 * 3, 640, 480, 640 * 3, 3, cap)
    kp_ptr, desc_ptr, cnt_ptr, kcap = ex.device_results()
    cs_ptr, ci_ptr = ex.device_grid()
    _, sf, _, _, _ = ex.scale_tables()
    d_fi = torch.tensor(pairs, dtype=torch.int32, device="cuda")
    d_np = torch.tensor(n_pts, dtype=torch.int32, device="cuda")
    d_iv, d_proj = torch.from_numpy(in_view).cuda(), torch.from_numpy(proj).cuda()
    d_pred, d_vcos, d_pd = torch.from_numpy(pred).cuda(), torch.from_numpy(vcos).cuda(), torch.from_numpy(pdesc).cuda()
    d_matched = torch.full((P, cap), 55, dtype=torch.int32, device="cuda")
    d_dist = torch.zeros((P, cap), dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(P, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for radius, ratio in ((1.0, 0.6), (4.0, 0.8), (9.0, 0.95)):
        mt.match_projection_points_batch(grid, P, d_fi.data_ptr(), kp_ptr, desc_ptr, cnt_ptr, cap, cs_ptr, ci_ptr, pcap,
                                         d_np.data_ptr(), d_iv.data_ptr(), d_proj.data_ptr(), d_pred.data_ptr(),
                                         d_vcos.data_ptr(), d_pd.data_ptr(), sf, radius, ratio, d_matched.data_ptr(),
                                         d_dist.data_ptr(), d_nm.data_ptr())
        torch.cuda.synchronize()
        hm, hd, hn = d_matched.cpu().numpy(), d_dist.cpu().numpy(), d_nm.cpu().numpy()
        for i, (f, n) in enumerate(zip(pairs, n_pts)):
            fk, fd, start, items = ora[f]
            om = oracle.match_projection_points(og, start, items, fk, fd, in_view[i, :n], proj[i, :n], pred[i, :n],
                                                vcos[i, :n], pdesc[i, :n], sf, radius, ratio)
            got_q = np.nonzero(hm[i, :len(fk)] > 0)[0]
            assert hn[i] == len(om) == len(got_q), (i, radius)
            assert np.array_equal(got_q, om["query"]) and np.array_equal(hm[i, got_q], om["train"]), (i, radius)
            assert np.array_equal(hd[i, got_q].astype(np.float32), om["distance"]), (i, radius)
        assert hn[3] == 0 and hn[0] > 50
    ex.close()
    mt.close()


def test_batched_projection_crowded_windows(oracle):
    """Windows that hold more acceptable candidates than a pre-sorted list has entries (a cluster of features with
    near-identical descriptors): those points are resolved by the in-kernel window scan, at their turn in the
    sequential order, and the result still equals the oracle's."""
    import torch
    rng = np.random.default_rng(77)
    N, cap, B = 700, 768, 2
    bounds = (0.0, 640.0, 0.0, 480.0)
    grid = orbx.make_grid(bounds)
    og = oracle.grid_params(bounds)
    sf = np.array([1.2 ** l for l in range(8)], np.float32)
    base_desc = rng.integers(0, 256, (4, 32), dtype=np.uint8)
    frames = []
    for f in range(B):
        k = np.zeros(N, orbx.KP_DTYPE)
        crowd = 300
        k["x"][:crowd] = 300 + rng.uniform(0, 24, crowd)
        k["y"][:crowd] = 200 + rng.uniform(0, 24, crowd)
        k["x"][crowd:] = rng.uniform(5, 635, N - crowd)
        k["y"][crowd:] = rng.uniform(5, 440, N - crowd)   # the grid (64 rows of 7 pixels) ends at y = 448
        k["size"], k["octave"], k["class_id"] = 31.0, 0, -1
        k["angle"] = rng.uniform(0, 360, N).astype(np.float32)
        d = rng.integers(0, 256, (N, 32), dtype=np.uint8)
        d[:crowd] = base_desc[rng.integers(0, 4, crowd)]
        flip = rng.integers(0, 32, crowd)
        d[np.arange(crowd), flip] ^= (1 << rng.integers(0, 8, crowd)).astype(np.uint8)   # distances 0..2 inside the crowd
        k2, start, items = oracle.assign_grid(og, k)
        assert len(k2) == N                       # all inside the grid: the order is unchanged
        frames.append((k2, d, start, items))
    ncell = grid.grid_cols * grid.grid_rows
    kps = np.zeros((B, cap), orbx.KP_DTYPE)
    desc = np.zeros((B, cap, 32), np.uint8)
    cs = np.zeros((B, ncell + 1), np.int32)
    ci = np.zeros((B, cap), np.int32)
    for f, (k2, d, start, items) in enumerate(frames):
        kps[f, :N], desc[f, :N], cs[f], ci[f, :N] = k2, d, start, items
    counts = np.full(B, N, np.int32)
    pairs = [(1, 0), (0, 1)]
    proj = np.zeros((len(pairs), cap, 2), np.float32)
    for i, (f, r) in enumerate(pairs):
        proj[i, :N, 0], proj[i, :N, 1] = frames[r][0]["x"], frames[r][0]["y"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()
    d_kps, d_desc, d_cnt, d_cs, d_ci, d_proj = t(kps), t(desc), t(counts), t(cs), t(ci), t(proj)
    d_fi = torch.tensor([a for a, _ in pairs], dtype=torch.int32, device="cuda")
    d_ri = torch.tensor([b for _, b in pairs], dtype=torch.int32, device="cuda")
    d_matched = torch.zeros((len(pairs), cap), dtype=torch.int32, device="cuda")
    d_dist = torch.zeros((len(pairs), cap), dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(len(pairs), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    mt = orbx.Matcher()
    for radius, orient in ((15.0, False), (30.0, True)):
        mt.match_projection_frame_batch(grid, len(pairs), d_fi.data_ptr(), d_ri.data_ptr(), d_kps.data_ptr(),
                                        d_desc.data_ptr(), d_cnt.data_ptr(), cap, d_cs.data_ptr(), d_ci.data_ptr(), 0,
                                        d_proj.data_ptr(), 0, sf, 0, radius, orient, d_matched.data_ptr(),
                                        d_dist.data_ptr(), d_nm.data_ptr())
        torch.cuda.synchronize()
        hm, hd, hn = d_matched.cpu().numpy(), d_dist.cpu().numpy(), d_nm.cpu().numpy()
        for i, (f, r) in enumerate(pairs):
            fk, fd, start, items = frames[f]
            rk, rd = frames[r][0], frames[r][1]
            om = oracle.match_projection_frame(og, start, items, fk, fd, rk, np.ones(N, np.uint8), proj[i, :N],
                                               np.ones(N, np.float32), rd, sf, 0, radius, orient)
            got_q = np.nonzero(hm[i, :N] > 0)[0]
            assert hn[i] == len(om) == len(got_q), (i, radius)
            assert np.array_equal(got_q, om["query"]) and np.array_equal(hm[i, got_q], om["train"])
            assert np.array_equal(hd[i, got_q].astype(np.float32), om["distance"])
        assert hn.min() > 50
    mt.close()


def test_wave_parallel_retain_best_matches_libstdcxx(oracle):
    """The wave-parallel nth_element + partition of the device selection leaves the kept responses in the order the
    real libstdc++ calls do: ties, sorted / reversed / constant inputs, and the heap-select fallback."""
    from tests.test_abi import retain_best_cases
    rng = np.random.default_rng(12)
    cases = retain_best_cases()
    for n in (63, 64, 65, 129, 2500, 8192):
        cases.append((rng.integers(20, 40, n).astype(np.float32), n // 4))
        cases.append((rng.normal(size=n).astype(np.float32), n // 2))
    for keys, npts in cases:
        a = orbx.selftest_retain_best_device(keys, npts)
        b = oracle.retain_best(keys, npts)
        assert np.array_equal(a, b), (len(keys), npts)


def test_4k_extraction_bit_exact(oracle):
    """3840x2160 gray frames, 8000 keypoints, 12 levels: single-frame detect / compute and the batch path against the
    oracle.  The batch's selection does not fit the selection kernel's LDS at this size: it runs in the kernel's
    global-memory form, still on the device (finish() reports no hand-over to the host selection), with and without
    the frame model; the host selection gives the same."""
    import torch
    w, h = 3840, 2160
    rng = np.random.default_rng(41)
    frames = []
    for f in range(2):
        img = np.full((h, w), 100, np.float32)
        for _ in range(6000):
            x0, y0 = rng.integers(0, w), rng.integers(0, h)
            ww, hh = rng.integers(3, 400), rng.integers(3, 300)
            img[y0:y0 + hh, x0:x0 + ww] = rng.integers(0, 256)
        img += rng.normal(0, 2.5, img.shape)
        frames.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    p = oracle.orb_params(nfeatures=8000, nlevels=12)
    ex = orbx.Extractor(n_key_points=8000, level_pyramid=12, max_width=w, max_height=h, max_batch=2)
    ora = []
    for img in frames:
        ok_ = oracle.detect(p, img)
        ora.append((ok_,) + tuple(oracle.compute(p, img, ok_)))
    k = ex.detect(frames[0], capacity=12000)
    assert len(k) > 7000 and np.array_equal(k, ora[0][0].astype(orbx.KP_DTYPE))
    k2, d2 = ex.compute(frames[0], k)
    assert np.array_equal(k2, ora[0][1].astype(orbx.KP_DTYPE)) and np.array_equal(d2, ora[0][2])
    dev = torch.from_numpy(np.stack(frames)).cuda()
    torch.cuda.synchronize()
    for device_select in (True, False):
        ex.set_device_select(device_select)
        kps, counts, desc = ex.extract_batch(dev.data_ptr(), 2, w * h, w, h, w, 1, 9000)
        for f in range(2):
            assert counts[f] == len(ora[f][1])
            assert np.array_equal(kps[f, :counts[f]], ora[f][1].astype(orbx.KP_DTYPE))
            assert np.array_equal(desc[f, :counts[f]], ora[f][2])
    ex.set_device_select(True)
    ex.extract_batch(dev.data_ptr(), 2, w * h, w, h, w, 1, 9000, to_host=False, deferred=True)
    c2, redone = ex.finish()
    assert not redone and np.array_equal(c2, counts)            # completed on the device
    # with the frame model (undistortion, 64 x 64 grid, border filter) at this size, device against host selection
    cam = orbx.make_camera(2800.0, 2800.0, 1919.5, 1079.5, [-0.05, 0.01, 0.0005, -0.0005, 0.0])
    grid = orbx.make_grid((0.0, float(w), 0.0, float(h)))
    ex.set_frame_model(cam, grid)
    res = []
    for device_select in (True, False):
        ex.set_device_select(device_select)
        kps, counts, desc = ex.extract_batch(dev.data_ptr(), 2, w * h, w, h, w, 1, 9000)
        cs, ci = ex.device_grid()
        ncell = grid.grid_cols * grid.grid_rows
        starts = torch.as_tensor(_DevView(cs, (2, ncell + 1), "<i4"), device="cuda").cpu().numpy().copy()
        items = torch.as_tensor(_DevView(ci, (2, 9000), "<i4"), device="cuda").cpu().numpy().copy()
        res.append((kps.copy(), counts.copy(), desc.copy(), starts, items))
    a, b = res
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) and a[1].min() > 7000
    for f in range(2):
        n, nk = a[1][f], a[3][f, -1]
        assert np.array_equal(a[0][f, :n], b[0][f, :n]) and np.array_equal(a[2][f, :n], b[2][f, :n])
        assert np.array_equal(a[4][f, :nk], b[4][f, :nk])
    ex.close()


def _feature_vector(desc, rng, n_nodes, drop=0.0):
    """A stand-in for DBoW2::FeatureVector: node id from a few descriptor bits (so matching features tend to share
    a node), ascending node ids, feature indices in ascending order inside a node; some nodes removed."""
    node = (desc[:, 0].astype(np.uint32) * 7 + desc[:, 5]) % n_nodes * 13 + 5
    ids = np.unique(node)
    if drop > 0:
        ids = ids[rng.uniform(size=len(ids)) >= drop]
    start, items = [0], []
    for i in ids:
        idx = np.nonzero(node == i)[0]
        items.extend(idx.tolist())
        start.append(len(items))
    return ids.astype(np.uint32), np.array(start, np.int32), np.array(items, np.int32)


def test_bow_matcher_bit_exact(oracle, matcher, ex1000, test_images):
    """BowOrbMatcher::match (orb_matcher.cpp:386-497) with synthetic feature vectors, against the oracle."""
    rng = np.random.default_rng(61)
    k0, d0 = ex1000.compute(test_images[0], ex1000.detect(test_images[0]))
    k1, d1 = ex1000.compute(test_images[1], ex1000.detect(test_images[1]))
    d1 = d1.copy()
    take = rng.choice(len(d1), min(len(d0), len(d1)) // 2, replace=False)
    d1[take] = d0[take % len(d0)] ^ (rng.uniform(size=(len(take), 32)) < 0.03).astype(np.uint8)  # near-duplicates
    has = (rng.uniform(size=len(k1)) > 0.15).astype(np.uint8)
    for n_nodes, drop in ((8, 0.0), (40, 0.2), (300, 0.3)):
        nodes = _feature_vector(d0, rng, n_nodes, drop)
        ref_nodes = _feature_vector(d1, rng, n_nodes, drop)
        for low, ratio, orient in ((50, 0.6, True), (80, 0.9, False), (30, 0.75, True)):
            m = matcher.match_bow(k0, d0, k1, d1, has, nodes, ref_nodes, low, ratio, orient)
            om = oracle.match_bow(k0, d0, k1, d1, has, nodes, ref_nodes, low, ratio, orient)
            assert np.array_equal(m, om.astype(orbx.MATCH_DTYPE)), (n_nodes, low, ratio, orient)
        if n_nodes == 8:
            assert len(m) > 30
    # no common node / empty vectors
    e = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    assert len(matcher.match_bow(k0, d0, k1, d1, has, e, ref_nodes)) == 0
    bad = (nodes[0][::-1].copy(), nodes[1], nodes[2])
    with pytest.raises(orbx.OrbxError):
        matcher.match_bow(k0, d0, k1, d1, has, bad, ref_nodes)


def test_epipolar_matcher_bit_exact(oracle, matcher, ex1000, test_images):
    """EpipolarConstraintWithBowMatcher::match (orb_matcher.cpp:520-661), quirks included: the distance is taken
    between row idx of both descriptor matrices, so the reference frame is a perturbed copy of the frame (same row
    order) for the test to produce matches at all."""
    rng = np.random.default_rng(83)
    k0, d0 = ex1000.compute(test_images[0], ex1000.detect(test_images[0]))
    n = len(k0)
    _, _, _, sig, _ = ex1000.scale_tables()
    flips = (rng.uniform(size=(n, 256)) < rng.uniform(0.0, 0.25, (n, 1))).astype(np.uint8)
    d1 = d0 ^ np.packbits(flips, axis=1)
    k1 = k0.copy()
    k1["x"] += rng.normal(0, 3, n).astype(np.float32)
    k1["y"] += rng.normal(0, 1.5, n).astype(np.float32)
    k1["angle"] = np.mod(k1["angle"] + rng.normal(0, 20, n), 360).astype(np.float32)
    k1["octave"] = np.clip(k1["octave"] + rng.integers(-1, 2, n), 0, 7)
    has = (rng.uniform(size=n) < 0.2).astype(np.uint8)
    ref_has = (rng.uniform(size=n) < 0.2).astype(np.uint8)
    F_tx = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)  # pure x translation: rows are epipolar lines
    A = rng.normal(size=(3, 3))
    u, w, vt = np.linalg.svd(A)
    F_gen = (u @ np.diag([w[0], w[1], 0]) @ vt).astype(np.float32) * 1e-3
    total = 0
    for n_nodes, drop in ((1, 0.0), (8, 0.0), (40, 0.2), (300, 0.3)):
        nodes = _feature_vector(d0, rng, n_nodes, drop)
        ref_nodes = _feature_vector(d1, rng, n_nodes, drop)
        for F in (F_tx, F_gen, np.zeros((3, 3), np.float32)):
            for low, orient in ((50, True), (80, False), (20, True)):
                m = matcher.match_epipolar(k0, d0, k1, d1, has, ref_has, F, sig, nodes, ref_nodes, low, orient)
                om = oracle.match_epipolar(k0, d0, k1, d1, has, ref_has, F, sig, nodes, ref_nodes, low, orient)
                assert np.array_equal(m, om.astype(orbx.MATCH_DTYPE)), (n_nodes, low, orient)
                total += len(m) if F is F_tx else 0
    assert total > 200
    # a node index the reference would use beyond the other frame's arrays
    nodes = _feature_vector(d0, rng, 8)
    short = _feature_vector(d1[:n // 2], rng, 8)
    assert oracle.match_epipolar(k0, d0, k1[:n // 2], d1[:n // 2], has * 0, ref_has, F_tx, sig, nodes, short, 50,
                                 True) is None
    with pytest.raises(orbx.OrbxError):
        matcher.match_epipolar(k0, d0, k1[:n // 2], d1[:n // 2], has * 0, ref_has, F_tx, sig, nodes, short)
    e = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    assert len(matcher.match_epipolar(k0, d0, k1, d1, has, ref_has, F_tx, sig, e, ref_nodes)) == 0


def test_fuse_search_bit_exact(oracle, matcher, ex1000, test_images):
    """Candidate search of ORBMatcher::fuse (orb_matcher.cpp:806-867) against the oracle."""
    rng = np.random.default_rng(87)
    bounds = (10.803785, 626.0599, 14.68408, 473.3248)
    g = orbx.make_grid(bounds)
    og = oracle.grid_params(bounds)
    k0, start, items = orbx.assign_grid(g, ex1000.detect(test_images[0]))
    k0, d0 = ex1000.compute(test_images[0], k0)
    _, sf, _, _, inv_sig = ex1000.scale_tables()
    n = len(k0)
    # map points: most sit near a feature of the key frame and carry a noisy copy of its descriptor
    n_pts = 3000
    src = rng.integers(0, n, n_pts)
    proj = np.stack([k0["x"][src], k0["y"][src]], 1) + rng.normal(0, 1.5, (n_pts, 2)).astype(np.float32)
    proj[:7] = [[-3, 5], [700, 100], [320, -2], [320, 500], [11, 15], [626, 473], [5000, 5000]]
    flips = (rng.uniform(size=(n_pts, 256)) < rng.uniform(0.0, 0.3, (n_pts, 1))).astype(np.uint8)
    pd = d0[src] ^ np.packbits(flips, axis=1)
    pred = np.clip(k0["octave"][src] + rng.integers(-1, 2, n_pts), 0, 7).astype(np.int32)
    in_view = (rng.uniform(size=n_pts) > 0.1).astype(np.uint8)
    hits = 0
    for mult in (3.0, 1.0, 10.0, 0.2):
        bi, bd = matcher.fuse_search(g, start, items, k0, d0, in_view, proj, pred, pd, sf, inv_sig, mult)
        oi, od = oracle.fuse_search(og, start, items, k0, d0, in_view, proj, pred, pd, sf, inv_sig, mult)
        assert np.array_equal(bi, oi) and np.array_equal(bd, od), mult
        hits = max(hits, int((bd <= 50).sum()))
    assert hits > 500
    assert (bi[in_view == 0] == -1).all()
    # ties: identical descriptors everywhere -> the first feature in scan order wins
    same = np.zeros_like(d0)
    bi, bd = matcher.fuse_search(g, start, items, k0, same, in_view, proj, pred, np.zeros_like(pd), sf, inv_sig, 10.0)
    oi, od = oracle.fuse_search(og, start, items, k0, same, in_view, proj, pred, np.zeros_like(pd), sf, inv_sig, 10.0)
    assert np.array_equal(bi, oi) and np.array_equal(bd, od)
    # no points / no features
    bi, bd = matcher.fuse_search(g, start, items, k0, d0, in_view[:0], proj[:0], pred[:0], pd[:0], sf, inv_sig, 3.0)
    assert len(bi) == 0


def _fundamental_of(K, R, t):
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.linalg.inv(K)
    F = Ki.T @ tx @ R @ Ki
    return (F / np.linalg.norm(F)).astype(np.float32)


def test_pose_recovery_and_triangulation_bit_exact(oracle, ransac):
    """Initializer::findRtWithFundamental / triangulatePoints (Initializer.cpp:165-252) for a batch of pairs:
    rotation, translation, cheirality count and every triangulated point equal the oracle's doubles bit for bit."""
    rng = np.random.default_rng(97)
    K = np.array([[517.3, 0, 318.6], [0, 516.5, 255.3], [0, 0, 1]], np.float32)
    Kd = K.astype(np.float64)
    sizes = [0, 1, 7, 8, 63, 64, 65, 255, 256, 257, 700, 2000] + rng.integers(8, 400, 20).tolist()
    Fs, pts_l, ref_l = [], [], []
    for p, n in enumerate(sizes):
        a, b = rng.uniform(-0.15, 0.15, 2)
        Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        R = Ry @ Rx
        t = rng.normal(size=3) * [0.4, 0.1, 0.2]
        X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n),
                      np.full(n, 4.0) if p % 5 == 4 else rng.uniform(1.5, 7, n)], 1)
        x1 = (Kd @ X.T).T
        x2 = (Kd @ (R @ X.T + t[:, None])).T
        ref = x1[:, :2] / x1[:, 2:]
        pts = x2[:, :2] / x2[:, 2:] + rng.normal(0, 0.5, (n, 2))
        # the reference widens float keypoint coordinates to double (Initializer.cpp:70-71)
        pts = pts.astype(np.float32).astype(np.float64)
        ref = ref.astype(np.float32).astype(np.float64)
        if n > 20:
            bad = rng.choice(n, n // 10, replace=False)
            pts[bad] = np.stack([rng.uniform(0, 640, len(bad)), rng.uniform(0, 480, len(bad))], 1)
        F = _fundamental_of(Kd, R, t)
        if p % 7 == 3:
            F = rng.normal(size=(3, 3)).astype(np.float32)  # not a fundamental matrix at all
        if p % 11 == 5:
            F = np.zeros((3, 3), np.float32)                 # every singular value vanishes: the completion path
        Fs.append(F)
        pts_l.append(pts)
        ref_l.append(ref)
    start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    pts, ref = np.concatenate(pts_l), np.concatenate(ref_l)
    R, t, good = ransac.recover_pose_batch(np.stack(Fs), K, pts, ref, start)
    n_right = 0
    for p, n in enumerate(sizes):
        oR, ot, og = oracle.recover_pose_fundamental(Fs[p], K, pts_l[p], ref_l[p])
        assert np.array_equal(R[p], oR, equal_nan=True) and np.array_equal(t[p], ot, equal_nan=True), p
        assert good[p] == og, p
        n_right += og > 0.7 * n
    assert n_right > len(sizes) // 2
    xyz = ransac.triangulate_batch(K, R, t, pts, ref, start)
    for p, n in enumerate(sizes):
        o = oracle.triangulate_initial(K, R[p], t[p], pts_l[p], ref_l[p])
        assert np.array_equal(xyz[start[p]:start[p + 1]], o, equal_nan=True), p
    # the building blocks, against the oracle's front ends
    P1 = np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = np.hstack([R[11], t[11][:, None]])
    q = oracle.triangulate_points(P1, P2, ref_l[11][:5], pts_l[11][:5])
    assert np.isfinite(q).all()


def _planar_pair(rng, K, n, rectified):
    """A plane seen from two poses: the homography in pixels (float32, arbitrary scale), the true motion, and n
    correspondences - rectified (K^-1 x, what the visibility filter is specified for) or in pixels (what the
    reference passes)."""
    Kd = K.astype(np.float64)
    a, b = rng.uniform(-0.2, 0.2, 2)
    Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    R = Ry @ Rx
    t = rng.normal(size=3) * 0.3
    nrm = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 1.0])
    nrm /= np.linalg.norm(nrm)
    d = rng.uniform(2, 5)
    Hn = R + np.outer(t, nrm) / d
    H = Kd @ Hn @ np.linalg.inv(Kd)
    H = (H / H[2, 2] * rng.uniform(0.3, 3)).astype(np.float32)
    X = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), np.zeros(n)], 1)
    X[:, 2] = (d - X[:, :2] @ nrm[:2]) / nrm[2]
    X2 = (R @ X.T).T + t
    x1, x2 = X / X[:, 2:], X2 / X2[:, 2:]
    if not rectified:
        x1, x2 = (Kd @ x1.T).T, (Kd @ x2.T).T
    f32 = lambda v: v[:, :2].astype(np.float32).astype(np.float64)
    return H, R, t / d, nrm, f32(x1), f32(x2)


def test_homography_pose_recovery_bit_exact(oracle, ransac):
    """Initializer::findRtWithHomography (Initializer.cpp:112-163) as its statements read - cv::decomposeHomographyMat,
    cv::filterHomographyDecompByVisibleRefpoints, first solution kept - for a batch of pairs: every motion, the
    number of motions, the solution chosen and its R, t equal the oracle's doubles bit for bit.  Covers rectified
    points (the filter's contract; the true motion must be among the visible ones), pixel points (what the
    reference passes), pure rotations (one motion), arbitrary 3 x 3 matrices, a zero matrix and pairs without points."""
    rng = np.random.default_rng(211)
    K = np.array([[517.3, 0, 318.6], [0, 516.5, 255.3], [0, 0, 1]], np.float32)
    sizes = [0, 1, 5, 63, 64, 65, 128, 700, 2000] + rng.integers(8, 400, 30).tolist()
    Hs, pts_l, ref_l, truth = [], [], [], []
    for p, n in enumerate(sizes):
        rectified = p % 2 == 0
        H, R, t, nrm, before, after = _planar_pair(rng, K, n, rectified)
        kind = "plane"
        if p % 9 == 4:
            a = rng.uniform(-0.2, 0.2)
            Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
            Kd = K.astype(np.float64)
            H = (Kd @ Rz @ np.linalg.inv(Kd) * 1.7).astype(np.float32)
            kind = "rotation"
        elif p % 9 == 6:
            H = rng.normal(size=(3, 3)).astype(np.float32)
            kind = "random"
        elif p % 13 == 8:
            H = np.zeros((3, 3), np.float32)
            kind = "zero"
        Hs.append(H)
        pts_l.append(before)
        ref_l.append(after)
        truth.append((kind, rectified, R, t, nrm))
    start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    R, t, idx, sols, nsol = ransac.recover_pose_homography_batch(
        np.stack(Hs), K, np.concatenate(pts_l), np.concatenate(ref_l), start)
    n_true = n_planes = 0
    for p, n in enumerate(sizes):
        oR, ot, on = oracle.decompose_homography(Hs[p], K)
        assert nsol[p] == len(oR), p
        got = sols[p, :nsol[p]]
        assert np.array_equal(got[:, :9].reshape(-1, 3, 3), oR, equal_nan=True), p
        assert np.array_equal(got[:, 9:12], ot, equal_nan=True) and np.array_equal(got[:, 12:], on, equal_nan=True), p
        keep = oracle.filter_homography_visible(oR, on, pts_l[p], ref_l[p])
        fR, ft, fi = oracle.recover_pose_homography(Hs[p], K, pts_l[p], ref_l[p])
        assert fi == (keep[0] if len(keep) else -1) == idx[p], p
        if fi >= 0:
            assert np.array_equal(R[p], fR, equal_nan=True) and np.array_equal(t[p], ft, equal_nan=True), p
        else:
            assert not R[p].any() and not t[p].any()
        kind, rectified, Rg, tg, ng = truth[p]
        if kind == "rotation":
            assert nsol[p] == 1
        if kind == "plane" and rectified and n >= 5:
            n_planes += 1
            n_true += any(np.abs(oR[k] - Rg).max() < 1e-5 and np.abs(on[k] - ng).max() < 1e-4 for k in keep)
    assert n_planes > 5 and n_true == n_planes


def test_backend_matchers_batched_equal_oracle(oracle, matcher, ex1000, test_images):
    """orbx_match_bow_batch / orbx_match_epipolar_batch: key-frame pairs of different sizes side by side, each
    equal to the oracle's result for that pair (and so to the single-pair entry points)."""
    rng = np.random.default_rng(101)
    _, _, _, sig, _ = ex1000.scale_tables()
    F_tx = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    k_all, d_all = ex1000.compute(test_images[0], ex1000.detect(test_images[0]))
    e = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    bow_pairs, epi_pairs = [], []
    for n, n_nodes in ((len(k_all), 40), (700, 8), (64, 3), (0, 5), (333, 120), (900, 1), (500, 40)):
        k0, d0 = k_all[:n], d_all[:n]
        flips = (rng.uniform(size=(n, 256)) < rng.uniform(0.0, 0.25, (n, 1))).astype(np.uint8)
        d1 = d0 ^ np.packbits(flips, axis=1) if n else d0
        k1 = k0.copy()
        k1["y"] += rng.normal(0, 1.5, n).astype(np.float32)
        k1["angle"] = np.mod(k1["angle"] + rng.normal(0, 20, n), 360).astype(np.float32)
        has = (rng.uniform(size=n) < 0.2).astype(np.uint8)
        ref_has = (rng.uniform(size=n) < 0.2).astype(np.uint8)
        nodes = _feature_vector(d0, rng, n_nodes, 0.1) if n else e
        ref_nodes = _feature_vector(d1, rng, n_nodes, 0.1) if n else e
        if n == 500:
            ref_nodes = (ref_nodes[0] + 100000, ref_nodes[1], ref_nodes[2])   # no common node
        bow_pairs.append((k0, d0, k1, d1, 1 - ref_has, nodes, ref_nodes))
        epi_pairs.append((k0, d0, k1, d1, has, ref_has, F_tx, nodes, ref_nodes))
    for low, ratio, orient in ((50, 0.6, True), (80, 0.9, False)):
        got = matcher.match_bow_batch(bow_pairs, low, ratio, orient)
        for p, g in zip(bow_pairs, got):
            want = oracle.match_bow(*p, low, ratio, orient) if len(p[0]) else np.zeros(0, orbx.MATCH_DTYPE)
            assert np.array_equal(g, want.astype(orbx.MATCH_DTYPE)), (len(p[0]), low)
        got = matcher.match_epipolar_batch(epi_pairs, sig, low, orient)
        for p, g in zip(epi_pairs, got):
            k0, d0, k1, d1, has, ref_has, F, nodes, ref_nodes = p
            want = (oracle.match_epipolar(k0, d0, k1, d1, has, ref_has, F, sig, nodes, ref_nodes, low, orient)
                    if len(k0) else np.zeros(0, orbx.MATCH_DTYPE))
            assert np.array_equal(g, want.astype(orbx.MATCH_DTYPE)), (len(k0), low)
    assert sum(len(g) for g in got) > 300
    assert matcher.match_bow_batch([]) == []


def test_best_descriptor_batch_bit_exact(oracle, matcher):
    """MapPoint::computeBestDescriptor (map_point.cpp:109-166) for a batch of observation sets."""
    rng = np.random.default_rng(71)
    sizes = [0, 1, 2, 3, 4, 5, 8, 17, 63, 64, 65, 200] + rng.integers(1, 40, 300).tolist()
    start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    desc = np.zeros((start[-1], 32), np.uint8)
    for p, n in enumerate(sizes):
        centre = rng.integers(0, 256, 32, dtype=np.uint8)
        noise = (rng.uniform(size=(n, 256)) < rng.uniform(0.02, 0.3)).astype(np.uint8)
        desc[start[p]:start[p + 1]] = centre ^ np.packbits(noise, axis=1)
    got = matcher.best_descriptor_batch(desc, start)
    want = oracle.best_descriptor(desc, start)
    assert np.array_equal(got, want)
    assert got[0] == -1 and got[1] == 0


def test_full_size_batches_are_consistent(oracle, ransac, test_images):
    """The bench-sized batches (792 VGA frames; 1024 pairs x 200 iterations x 2000 correspondences): every copy of a
    repeated unit gives the unit's result, and the distinct units equal the oracle."""
    import torch
    # extraction: 792 frames tiled from 4 distinct ones
    distinct = [test_images[0], test_images[1], synth_image(91, 640, 480, 3), synth_image(92, 640, 480, 3)]
    B, cap = 792, 1280
    dev = torch.from_numpy(np.stack([distinct[f % 4] for f in range(B)])).cuda()
    torch.cuda.synchronize()
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=B)
    kps, counts, desc = ex.extract_batch(dev.data_ptr(), B, 640 * 480 * 3, 640, 480, 640 * 3, 3, cap)
    p = oracle.orb_params(nfeatures=1000)
    for u in range(4):
        ok_, od = oracle.compute(p, distinct[u], oracle.detect(p, distinct[u]))
        for f in range(u, B, 4):
            assert counts[f] == len(ok_)
        f_all = np.arange(u, B, 4)
        assert (kps[f_all, :len(ok_)] == ok_.astype(orbx.KP_DTYPE)[None]).all()
        assert (desc[f_all, :len(ok_)] == od[None]).all()
    ex.close()
    # the bench's own configuration: frame model on (TUM lens, 64 x 64 grid), selection on the device, and the windowed
    # match of all 791 consecutive pairs (radius 15, orientation check): every pair equals the oracle's result for
    # its kind (the frames repeat with period 4, so there are four kinds of pair)
    cam = orbx.make_camera(517.3, 516.5, 318.6, 255.3, TUM_DIST)
    ocam = oracle.camera(517.3, 516.5, 318.6, 255.3, TUM_DIST, 640, 480)
    bounds = oracle.image_bounds(ocam)
    grid, og = orbx.make_grid(bounds), oracle.grid_params(bounds)
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=B)
    ex.set_frame_model(cam, grid)
    mt = orbx.Matcher()
    ts = torch.cuda.Stream()
    ex.set_stream(ts.cuda_stream)
    mt.set_stream(ts.cuda_stream)
    distinct = [test_images[0], np.roll(test_images[0], 3, axis=1).copy(), test_images[1],
                np.roll(test_images[1], -2, axis=0).copy()]           # neighbours show (nearly) the same scene
    del dev
    dev = torch.from_numpy(np.stack([distinct[f % 4] for f in range(B)])).cuda()
    torch.cuda.synchronize()
    ora = []
    for img in distinct:
        k = oracle.detect(p, img)
        xy = oracle.undistort_points(ocam, np.stack([k["x"], k["y"]], 1))
        k["x"], k["y"] = xy[:, 0], xy[:, 1]
        k, start, items = oracle.assign_grid(og, k)
        k2, d = oracle.compute(p, img, k)
        ora.append((k2, d, start, items))
    rng = np.random.default_rng(97)
    shift = [rng.normal(0, 3, (cap, 2)).astype(np.float32) for _ in range(4)]   # per kind of pair
    proj = np.zeros((B - 1, cap, 2), np.float32)
    for i in range(B - 1):
        rk = ora[i % 4][0]                                                      # pair i: frame i + 1 against frame i
        proj[i, :len(rk), 0] = rk["x"] + shift[i % 4][:len(rk), 0]
        proj[i, :len(rk), 1] = rk["y"] + shift[i % 4][:len(rk), 1]
    with torch.cuda.stream(ts):
        d_fi = torch.arange(1, B, dtype=torch.int32, device="cuda")
        d_ri = torch.arange(0, B - 1, dtype=torch.int32, device="cuda")
        d_proj = torch.from_numpy(proj).cuda()
        d_matched = torch.zeros((B - 1, cap), dtype=torch.int32, device="cuda")
        d_dist = torch.zeros((B - 1, cap), dtype=torch.int32, device="cuda")
        d_nm = torch.zeros(B - 1, dtype=torch.int32, device="cuda")
        ex.extract_batch(dev.data_ptr(), B, 640 * 480 * 3, 640, 480, 640 * 3, 3, cap, to_host=False, deferred=True)
        kp_ptr, desc_ptr, cnt_ptr, kcap = ex.device_results()
        cs_ptr, ci_ptr = ex.device_grid()
        _, sf, _, _, _ = ex.scale_tables()
        mt.match_projection_frame_batch(grid, B - 1, d_fi.data_ptr(), d_ri.data_ptr(), kp_ptr, desc_ptr, cnt_ptr, cap,
                                        cs_ptr, ci_ptr, 0, d_proj.data_ptr(), 0, sf, 0, 15.0, True,
                                        d_matched.data_ptr(), d_dist.data_ptr(), d_nm.data_ptr())
    torch.cuda.synchronize()
    counts, redone = ex.finish()
    assert not redone
    hm, hd, hn = d_matched.cpu().numpy(), d_dist.cpu().numpy(), d_nm.cpu().numpy()
    for u in range(4):
        fk, fd, start, items = ora[(u + 1) % 4]
        rk, rd = ora[u][0], ora[u][1]
        assert (counts[np.arange((u + 1) % 4, B, 4)] == len(fk)).all()
        om = oracle.match_projection_frame(og, start, items, fk, fd, rk, np.ones(len(rk), np.uint8),
                                           proj[u, :len(rk)], np.ones(len(rk), np.float32), rd, sf, 0, 15.0, True)
        assert len(om) > 100
        want_m = np.zeros(cap, np.int32)
        want_d = np.zeros(cap, np.int32)
        want_m[om["query"]] = om["train"]
        want_d[om["query"]] = om["distance"].astype(np.int32)
        idx = np.arange(u, B - 1, 4)
        got_m = np.where(hm[idx] > 0, hm[idx], 0)
        assert (hn[idx] == len(om)).all()
        assert (got_m == want_m[None]).all()
        assert (np.where(hm[idx] > 0, hd[idx], 0) == want_d[None]).all()
    ex.close()
    mt.close()
    # RANSAC: 1024 pairs tiled from 8 distinct ones
    rng = np.random.default_rng(93)
    P, n, it, D = 1024, 2000, 200, 8
    data = [_two_view(rng, n, 600, planar=(i % 2 == 1)) for i in range(D)]
    sets_d = [orbx.ransac_make_sets(n, it, 300 + i) for i in range(D)]
    pts = np.ascontiguousarray(np.stack([data[i % D][0] for i in range(P)]))
    ref = np.ascontiguousarray(np.stack([data[i % D][1] for i in range(P)]))
    sets = np.stack([sets_d[i % D] for i in range(P)])
    d_pts, d_ref = torch.from_numpy(pts).cuda(), torch.from_numpy(ref).cuda()
    d_sets = torch.from_numpy(sets.astype(np.int64)).to(torch.int32).cuda()
    d_F = torch.zeros((P, 9), dtype=torch.float32, device="cuda")
    d_H = torch.zeros_like(d_F)
    d_if = torch.zeros((P, n), dtype=torch.uint8, device="cuda")
    d_ih = torch.zeros_like(d_if)
    d_sc = torch.zeros((P, 2), dtype=torch.float32, device="cuda")
    d_best = torch.zeros((P, 2), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ransac.fh_batch(d_pts.data_ptr(), d_ref.data_ptr(), n, d_sets.data_ptr(), it, P, 1.0, d_F.data_ptr(),
                    d_H.data_ptr(), d_if.data_ptr(), d_ih.data_ptr(), d_sc.data_ptr(), d_best.data_ptr())
    torch.cuda.synchronize()
    F, Hm, best, sc = d_F.cpu().numpy(), d_H.cpu().numpy(), d_best.cpu().numpy(), d_sc.cpu().numpy()
    inl_f, inl_h = d_if.cpu().numpy(), d_ih.cpu().numpy()
    for u in range(D):
        o = oracle.ransac_fh(data[u][0], data[u][1], sets_d[u], 1.0)
        idx = np.arange(u, P, D)
        assert (best[idx] == o["best_it"][None]).all()
        assert (F[idx] == o["F"].reshape(1, 9)).all() and (Hm[idx] == o["H"].reshape(1, 9)).all()
        assert (inl_f[idx] == o["inliers_f"][None]).all() and (inl_h[idx] == o["inliers_h"][None]).all()
        assert (sc[idx, 0] == np.float32(o["score_f"])).all() and (sc[idx, 1] == np.float32(o["score_h"])).all()
