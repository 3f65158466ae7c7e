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


def test_keyframe_database_two_ranks_equal_single_process():
    """The key-frame database step (all-gather of the ranks' key-frame descriptor blocks, then every local frame
    against the whole database) through the same driver functions the GPU path uses (orb_slam_amd/kfdb.py), with a
    CPU stand-in only for the matching call: two gloo ranks give, concatenated in frame order, what one process gives."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from orb_slam_amd import kfdb
    n_frames, world, cap = 40, 2, 64
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
    assert rows == [0, 10, 20, 30]
    db_desc, db_counts = kfdb.gather_database(desc, counts, rows)
    idx, dd = kfdb.match_frames_against_database(kfdb.NumpyBackend(), desc, counts, db_desc, db_counts)
    assert gathered[0][0] == 0 and gathered[0][1] == gathered[1][0] and gathered[1][1] == n_frames
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
