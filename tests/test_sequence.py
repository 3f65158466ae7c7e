"""Sequence input: the TUM association list as TUMRGBDCamera::setupCameraStream parses it (reference
src/geometry/rgbd_camera.cpp:225-252), the PNG decoder against PIL for what cv::imread would return under the two
flag values the reference uses (rgbd_camera.h:211-226), the depth scaling, and (GPU) the batch feeder."""
import io
import os

import numpy as np
import pytest

from orb_slam_amd import orbx

PIL = pytest.importorskip("PIL.Image")


def _png_bytes(arr, mode=None, **kw):
    buf = io.BytesIO()
    PIL.fromarray(arr, mode).save(buf, format="PNG", **kw)
    return buf.getvalue()


def test_png_decoder_returns_what_imread_returns():
    rng = np.random.default_rng(5)
    h, w = 37, 53
    smooth = (np.add.outer(np.arange(h), np.arange(w)) * 3 % 256).astype(np.uint8)
    rgb = np.stack([smooth, rng.integers(0, 256, (h, w), dtype=np.uint8), smooth[::-1]], 2)
    # 8-bit RGB (the TUM colour images): BGR out, both modes
    for level in (0, 1, 9):
        data = _png_bytes(rgb, compress_level=level)
        assert np.array_equal(orbx.png_decode(data, orbx.IMREAD_UNCHANGED), rgb[:, :, ::-1])
        assert np.array_equal(orbx.png_decode(data, orbx.IMREAD_COLOR), rgb[:, :, ::-1])
    # RGBA: BGRA unchanged, alpha dropped in colour mode
    rgba = np.concatenate([rgb, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], 2)
    data = _png_bytes(rgba)
    assert np.array_equal(orbx.png_decode(data), rgba[:, :, [2, 1, 0, 3]])
    assert np.array_equal(orbx.png_decode(data, orbx.IMREAD_COLOR), rgb[:, :, ::-1])
    # 8-bit gray
    data = _png_bytes(smooth)
    assert np.array_equal(orbx.png_decode(data), smooth)
    assert np.array_equal(orbx.png_decode(data, orbx.IMREAD_COLOR), np.repeat(smooth[:, :, None], 3, 2))
    # 16-bit gray (the TUM depth images): 16UC1 unchanged; the reference reads it with the default flag and gets
    # three 8-bit channels holding the high byte
    depth = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    depth[:, :20] = (np.arange(20) * 2500).astype(np.uint16)
    data = _png_bytes(depth)
    assert np.array_equal(orbx.png_decode(data), depth)
    hi = (depth >> 8).astype(np.uint8)
    got = orbx.png_decode(data, orbx.IMREAD_COLOR)
    assert np.array_equal(got, np.repeat(hi[:, :, None], 3, 2))
    # depth scaling of TUMRGBDCamera::imageDepth: float product with (float)(255 * 0.0002)
    scale = np.float32(255 * 0.0002)
    assert np.array_equal(orbx.depth_to_float(got), got.astype(np.float32) * scale)
    # palette and gray + alpha
    pal = PIL.fromarray(rgb).quantize(17)
    buf = io.BytesIO()
    pal.save(buf, format="PNG")
    assert np.array_equal(orbx.png_decode(buf.getvalue()), np.asarray(pal.convert("RGB"))[:, :, ::-1])
    la = np.stack([smooth, rng.integers(0, 256, (h, w), dtype=np.uint8)], 2)
    data = _png_bytes(la, "LA")
    assert np.array_equal(orbx.png_decode(data), la[:, :, [0, 0, 0, 1]])
    # errors: not a PNG, truncated, interlaced, capacity
    with pytest.raises(orbx.OrbxError):
        orbx.png_decode(b"not a png at all, just some bytes of text....")
    with pytest.raises(orbx.OrbxError):
        orbx.png_decode(_png_bytes(rgb)[:200])


def test_association_list_parsing(tmp_path):
    d = str(tmp_path) + "/"
    lines = ["1305031102.175304 rgb/1305031102.175304.png 1305031102.160407 depth/1305031102.160407.png",
             "",
             "1305031102.211214 rgb/1305031102.211214.png 1305031102.226738 depth/1305031102.226738.png",
             "1305031102.243211   rgb/a.png\t1305031102.262886 depth/b.png", ""]
    with open(d + "assoc.txt", "w") as f:
        f.write("\n".join(lines))
    seq = orbx.Sequence(d)
    assert len(seq) == 3
    t, rgb, depth = seq.entry(0)
    assert t == 1305031102.175304 and rgb == "rgb/1305031102.175304.png" and depth == "depth/1305031102.160407.png"
    assert seq.entry(2) == (1305031102.243211, "rgb/a.png", "depth/b.png")
    with pytest.raises(orbx.OrbxError):
        seq.entry(3)
    seq.close()
    with pytest.raises(orbx.OrbxError):
        orbx.Sequence(d + "missing/")
    open(d + "empty.txt", "w").close()
    os.makedirs(d + "e")
    open(d + "e/assoc.txt", "w").close()
    with pytest.raises(orbx.OrbxError):   # the reference asserts on an empty list
        orbx.Sequence(d + "e/")


class _DevView:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}


@pytest.mark.gpu
def test_feeder_delivers_every_frame_in_order(tmp_path):
    import torch
    rng = np.random.default_rng(11)
    w, h, n, batch = 160, 120, 21, 8
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    frames[:, 10:60, 20:90] = (frames[:, 10:60, 20:90] // 8) + 100   # some compressible area
    d = str(tmp_path) + "/"
    os.makedirs(d + "rgb")
    with open(d + "assoc.txt", "w") as f:
        for i in range(n):
            PIL.fromarray(frames[i][:, :, ::-1]).save(d + f"rgb/{i:04d}.png")   # file holds RGB; decoder returns BGR
            f.write(f"{1000.0 + i / 30:.6f} rgb/{i:04d}.png {1000.0 + i / 30:.6f} depth/{i:04d}.png\n")
    seq = orbx.Sequence(d)
    stream = torch.cuda.Stream()
    for n_slots, use_stream in ((2, False), (3, True)):
        fd = orbx.Feeder(w, h, 3, batch, n_slots=n_slots, n_threads=4)
        for job in range(2):   # a feeder can run several jobs
            fd.start_sequence(seq, 2 * job, n - 2 * job)
            got = []
            while True:
                ptr, k = fd.next(stream.cuda_stream if use_stream else None)
                if k == 0:
                    break
                with torch.cuda.stream(stream):
                    t = torch.as_tensor(_DevView(ptr, (k, h, w, 3), "|u1"), device="cuda").clone()
                if not use_stream:
                    stream.synchronize()
                got.append(t)
            stream.synchronize()
            out = torch.cat(got).cpu().numpy()
            assert np.array_equal(out, frames[2 * job:])
        # frames already decoded in host memory
        fd.start_memory(frames)
        got = []
        while True:
            ptr, k = fd.next(None)
            if k == 0:
                break
            got.append(torch.as_tensor(_DevView(ptr, (k, h, w, 3), "|u1"), device="cuda").clone())
            torch.cuda.synchronize()
        assert np.array_equal(torch.cat(got).cpu().numpy(), frames)
        fd.close()
    # a frame of the wrong size ends the job with an error
    PIL.fromarray(frames[0][:50]).save(d + "rgb/0003.png")
    fd = orbx.Feeder(w, h, 3, batch)
    fd.start_sequence(seq, 0, n)
    with pytest.raises(orbx.OrbxError):
        while fd.next(None)[1]:
            pass
    fd.close()


@pytest.mark.gpu
def test_fed_batches_extract_like_uploaded_frames(oracle, test_images):
    """Feeder -> orbx_extract_batch on the consumer stream == extraction of the same frames uploaded with torch."""
    import torch
    frames = np.stack([test_images[i % 2] for i in range(6)])
    ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=4)
    dev = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    want_k, want_c, want_d = ex.extract_batch(dev.data_ptr(), 4, 640 * 480 * 3, 640, 480, 640 * 3, 3, 1280)
    fd = orbx.Feeder(640, 480, 3, 4, n_slots=2)
    fd.start_memory(frames)
    ptr, k = fd.next(None)
    assert k == 4
    kps, counts, desc = ex.extract_batch(ptr, 4, 640 * 480 * 3, 640, 480, 640 * 3, 3, 1280)
    fed = torch.as_tensor(_DevView(ptr, (4, 480, 640, 3), "|u1"), device="cuda").cpu().numpy()
    assert np.array_equal(fed, frames[:4]), "uploaded frames differ"
    assert np.array_equal(counts, want_c), (counts, want_c)
    for b in range(4):
        c = counts[b]
        assert np.array_equal(kps[b, :c], want_k[b, :c]), b
        assert np.array_equal(desc[b, :c], want_d[b, :c]), b
    ptr, k = fd.next(None)
    assert k == 2
    assert fd.next(None)[1] == 0
    fd.close()
    ex.close()
