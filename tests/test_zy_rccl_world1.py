"""The collective of the key-frame database on the GPU's own backend (RCCL), as far as one GPU can show it: a process
group of one rank is initialised with the nccl backend exactly as bench.py does for N > 1, and the database gather,
the barrier and the max-reduction of the timing run through it.  More ranks need more GPUs than the test box has."""
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_keyframe_database_gather_over_rccl_single_rank():
    import torch
    import torch.distributed as dist
    from orb_slam_amd import kfdb
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    device = torch.device("cuda", 0)
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=device)
    except Exception as e:  # the library is part of the image; a box where it cannot start says so instead of failing
        pytest.skip(f"RCCL did not initialise here: {e}")
    try:
        rng = np.random.default_rng(5)
        frames, cap = 20, 64
        desc = torch.from_numpy(rng.integers(0, 256, (frames, cap, 32), dtype=np.uint8)).to(device)
        counts = torch.from_numpy(rng.integers(1, cap + 1, frames).astype(np.int32)).to(device)
        rows = kfdb.keyframe_rows(0, frames, 10)
        db, dbc = kfdb.gather_database(desc, counts, rows, dist)
        assert torch.equal(db.cpu(), desc[rows].cpu()) and torch.equal(dbc.cpu(), counts[rows].cpu())
        dist.barrier()
        t = torch.tensor([1.5], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.5
    finally:
        dist.destroy_process_group()
