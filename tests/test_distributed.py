"""CPU tests of the N > 1 path: 2 gloo ranks shard frames in contiguous blocks; the concatenation of the shards'
results equals the single-process result, and the key-frame descriptor all-gather replicates the full set."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _per_frame(f):
    """A deterministic per-frame unit of work through the C ABI's host entry points (no GPU)."""
    from orb_slam_amd import orbx
    sets = orbx.ransac_make_sets(100 + f, 4, f)
    m = np.zeros(50, orbx.MATCH_DTYPE)
    rng = np.random.default_rng(f)
    m["query"] = np.arange(50)
    m["train"] = rng.integers(0, 20, 50)
    m["distance"] = rng.integers(0, 80, 50).astype(np.float32)
    return int(sets.sum()) * 1000 + len(orbx.filter_matches_host(m, 50))


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from orb_slam_amd.shard import shard_range, all_gather_descriptors
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_frames, rank, world)
    local = [_per_frame(f) for f in range(lo, hi)]
    k_local = 3
    desc = torch.full((k_local, 5, 32), rank, dtype=torch.uint8)
    desc[:, :, 0] = torch.arange(k_local, dtype=torch.uint8)[:, None] + rank * k_local
    full = all_gather_descriptors(desc, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, local))
    dist.barrier()
    if rank == 0:
        q.put((gathered, full.numpy()))
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process():
    import torch.multiprocessing as mp
    n_frames, world = 11, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    ranges = [(lo, hi) for lo, hi, _ in gathered]
    assert ranges[0][0] == 0 and ranges[-1][1] == n_frames and ranges[0][1] == ranges[1][0]
    merged = [v for _, _, loc in gathered for v in loc]
    assert merged == [_per_frame(f) for f in range(n_frames)]
    assert full.shape == (6, 5, 32)
    assert full[:, 0, 0].tolist() == [0, 1, 2, 3, 4, 5] and full[3:, :, 1].min() == 1 and full[:3, :, 1].max() == 0


def _kfdb_descriptors(first, n, cap):
    """Deterministic per-frame descriptor blocks keyed by the GLOBAL frame index (so any sharding sees the same data):
    frame g has 40 + (7 g mod 17) rows; consecutive frames share most rows, as consecutive video frames do."""
    import torch
    base = np.random.default_rng(1234).integers(0, 256, (cap + 64, 32), dtype=np.uint8)
    desc = np.zeros((n, cap, 32), np.uint8)
    counts = np.zeros(n, np.int32)
    for i in range(n):
        g = first + i
        counts[i] = 40 + (7 * g) % 17
        rows = base[g % 5:g % 5 + counts[i]].copy()
        flip = np.random.default_rng(g).integers(0, 256, rows.shape, dtype=np.uint8) & 0x11
        desc[i, :counts[i]] = rows ^ flip
    return torch.from_numpy(desc), torch.from_numpy(counts)


def _kfdb_worker(rank, world, port, n_frames, cap, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from orb_slam_amd import kfdb
    from orb_slam_amd.shard import shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_frames, rank, world)
    desc, counts = _kfdb_descriptors(lo, hi - lo, cap)
    db_desc, db_counts = kfdb.gather_database(desc, counts, kfdb.keyframe_rows(lo, hi - lo, 10), dist)
    idx, dd = kfdb.match_frames_against_database(kfdb.NumpyBackend(), desc, counts, db_desc, db_counts)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, db_counts.numpy(), idx.numpy(), dd.numpy()))
    dist.barrier()
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,world", [(40, 2), (45, 2), (31, 3)])
def test_keyframe_database_ranks_equal_single_process(n_frames, world):
    """The key-frame database step (all-gather of the ranks' key-frame descriptor blocks, then every local frame
    against the whole database) through the same driver functions the GPU path uses (orb_slam_amd/kfdb.py), with a
    CPU stand-in only for the matching call: the gloo ranks give, concatenated in frame order, what one process gives
    - also when the ranks hold different numbers of key frames (45 frames on 2 ranks: 3 + 2; 31 on 3: 1 + 1 + 2)."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from orb_slam_amd import kfdb
    cap = 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_kfdb_worker, args=(r, world, port, n_frames, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    desc, counts = _kfdb_descriptors(0, n_frames, cap)
    rows = kfdb.keyframe_rows(0, n_frames, 10)
    assert rows == list(range(0, n_frames, 10))
    db_desc, db_counts = kfdb.gather_database(desc, counts, rows)
    idx, dd = kfdb.match_frames_against_database(kfdb.NumpyBackend(), desc, counts, db_desc, db_counts)
    assert gathered[0][0] == 0 and gathered[-1][1] == n_frames
    assert all(gathered[r][1] == gathered[r + 1][0] for r in range(world - 1))
    for _, _, dbc, _, _ in gathered:
        assert np.array_equal(dbc, db_counts.numpy())       # every rank holds the whole database, in rank order
    assert np.array_equal(np.concatenate([g[3] for g in gathered]), idx.numpy())
    assert np.array_equal(np.concatenate([g[4] for g in gathered]), dd.numpy())
    # a frame that is a key frame matches its own database entry exactly
    assert (dd.numpy()[10, 1, :counts[10]] == 0).all() and (idx.numpy()[10, 1, :counts[10]] == np.arange(counts[10])).all()


def test_shard_range_covers_everything():
    from orb_slam_amd.shard import shard_range
    for n in (0, 1, 7, 10000):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        shard_range(5, 2, 2)


# ---- two ranks on one GPU: real frames through the HIP path ------------------------------------------------------------

def _gpu_sequence(first, n, w=320, h=240):
    """Frames first .. first + n - 1 of a deterministic sequence: a window sliding over one large synthetic image
    (one pixel right, two down per frame) plus per-frame noise keyed by the GLOBAL frame index."""
    from tests.conftest import synth_image
    base = synth_image(4242, w + 64, h + 96, 3)
    out = np.zeros((n, h, w, 3), np.uint8)
    for i in range(n):
        g = first + i
        crop = base[2 * g:2 * g + h, g:g + w].astype(np.int16)
        noise = np.random.default_rng(9000 + g).integers(-3, 4, crop.shape)
        out[i] = np.clip(crop + noise, 0, 255).astype(np.uint8)
    return out


def _gpu_extract_and_match(lo, hi, w=320, h=240, cap=1200):
    """Extraction of frames lo .. hi - 1 and the windowed match of each to its predecessor (frame lo - 1 is extracted
    as well when it exists: the pair (lo, lo - 1) belongs to the rank that owns lo).  Everything through the C ABI on
    the GPU; returns host arrays for frames lo .. hi - 1."""
    import torch
    from orb_slam_amd import orbx
    first = max(lo - 1, 0)
    frames = _gpu_sequence(first, hi - first, w, h)
    n = len(frames)
    ex = orbx.Extractor(n_key_points=600, level_pyramid=6, max_width=w, max_height=h, max_batch=n)
    grid = orbx.make_grid((0.0, float(w), 0.0, float(h)), 32, 32)
    ex.set_frame_model(None, grid)
    mt = orbx.Matcher()
    dev = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    kps, counts, desc = ex.extract_batch(dev.data_ptr(), n, w * h * 3, w, h, w * 3, 3, cap)
    kp_ptr, desc_ptr, cnt_ptr, kcap = ex.device_results()
    cs_ptr, ci_ptr = ex.device_grid()
    n_pairs = n - 1
    matched = np.zeros((0, kcap), np.int32)
    if n_pairs > 0:
        frame_idx = torch.arange(1, n, dtype=torch.int32, device="cuda")
        ref_idx = torch.arange(0, n - 1, dtype=torch.int32, device="cuda")
        # the scene moves one pixel left and two up per frame
        proj = np.zeros((n_pairs, kcap, 2), np.float32)
        proj[:, :min(cap, kcap), 0] = kps["x"][:-1, :kcap] - 1.0
        proj[:, :min(cap, kcap), 1] = kps["y"][:-1, :kcap] - 2.0
        d_proj = torch.from_numpy(proj).cuda()
        d_matched = torch.zeros((n_pairs, kcap), dtype=torch.int32, device="cuda")
        d_dist = torch.zeros((n_pairs, kcap), dtype=torch.int32, device="cuda")
        d_nm = torch.zeros(n_pairs, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        mt.match_projection_frame_batch(grid, n_pairs, frame_idx.data_ptr(), ref_idx.data_ptr(), kp_ptr, desc_ptr,
                                        cnt_ptr, kcap, cs_ptr, ci_ptr, 0, d_proj.data_ptr(), 0, ex.scale_tables()[1],
                                        0, 15.0, True, d_matched.data_ptr(), d_dist.data_ptr(), d_nm.data_ptr())
        torch.cuda.synchronize()
        matched = d_matched.cpu().numpy()
    own = lo - first   # drop the predecessor frame that was only extracted for the first pair
    out = []
    for i in range(own, n):
        c = int(counts[i])
        m = matched[i - 1, :c] if i >= 1 else np.zeros(0, np.int32)
        out.append((kps[i, :c].copy(), desc[i, :c].copy(), m.copy()))
    ex.close()
    mt.close()
    return out


def _gpu_worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from orb_slam_amd.shard import shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_frames, rank, world)
    res = _gpu_extract_and_match(lo, hi)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, res))
    dist.barrier()
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_equal_single_process():
    """Two ranks (gloo process group, both on GPU 0) extract their blocks of a sequence and match every frame to its
    predecessor through the HIP path; concatenated in frame order that is bit for bit what one process gives:
    keypoints, descriptors and window matches."""
    import torch.multiprocessing as mp
    n_frames, world = 9, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _gpu_extract_and_match(0, n_frames)
    merged = [fr for _, _, res in gathered for fr in res]
    assert len(merged) == len(single) == n_frames
    for f, ((k1, d1, m1), (k2, d2, m2)) in enumerate(zip(merged, single)):
        assert len(k2) > 200, f
        assert np.array_equal(k1, k2) and np.array_equal(d1, d2), f
        assert np.array_equal(m1, m2), f
        if f > 0:
            assert (m2 > 0).sum() > 100, f   # the sliding window really is matched frame to frame
