"""Key-frame descriptor database: many-to-many brute-force Hamming matching over all ranks' key frames.

Every rank extracts its own contiguous block of frames (no collective on that path).  Every `every`-th frame of the
sequence is a key frame; the ranks replicate their key frames' descriptor blocks with ONE all-gather of uint8 rows
into a preallocated buffer (RCCL over xGMI on GPUs, gloo on CPU) plus one of the per-key-frame row counts and one of
the per-rank key-frame numbers, and then each rank matches each of
its frames against every key frame of the database: per (frame, key frame) pair what CVORBMatcher::match returns
(reference src/geometry/orb_matcher.cpp:671-677: for every query row the train row of minimum Hamming distance,
lowest index on ties).

The matching call is a backend object, so that the same driver runs on the GPU (HipBackend: the C ABI's
orbx_match_hamming_cross, one call per block of frames: every frame of the block against every key frame) and in the
CPU tests of the multi-rank path (a stand-in backend with the same contract).
"""
import numpy as np


def keyframe_rows(first_global_index, n_local, every=10):
    """Local indices of the key frames among frames first_global_index .. first_global_index + n_local - 1."""
    first = (-first_global_index) % every
    return list(range(first, n_local, every))


def gather_database(desc, counts, key_rows, dist=None, out=None):
    """desc: [frames, cap, 32] uint8 tensor, counts: [frames] int32 tensor (same device), key_rows: local indices.
    Returns (db_desc [K, cap, 32], db_counts [K]) holding the key frames of all ranks in rank order.

    The ranks may hold different numbers of key frames (a sequence whose length is no multiple of the key-frame
    spacing times the number of ranks): the per-rank numbers are gathered first, every rank pads its block to the
    largest, ONE all_gather_into_tensor fills a [world * k_max, cap, 32] buffer in place (no per-rank list, no
    concatenation copy), and the padding is dropped only when there is any.  `out`: a (descriptor buffer, count
    buffer) pair from an earlier call to fill again when it is large enough - a caller that gathers every batch then
    allocates nothing."""
    import torch
    rows = torch.as_tensor(key_rows, dtype=torch.long, device=desc.device)
    k_local = len(key_rows)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return desc.index_select(0, rows).contiguous(), counts.index_select(0, rows).contiguous()
    world = dist.get_world_size()
    cap = desc.shape[1]
    per_rank = torch.empty(world, dtype=torch.int32, device=desc.device)
    dist.all_gather_into_tensor(per_rank, torch.tensor([k_local], dtype=torch.int32, device=desc.device))
    per_rank = per_rank.tolist()
    k_max = max(per_rank)
    if k_max == 0:
        return desc[:0].contiguous(), counts[:0].contiguous()
    send_d = torch.zeros((k_max, cap, 32), dtype=torch.uint8, device=desc.device)
    send_c = torch.zeros(k_max, dtype=torch.int32, device=desc.device)
    if k_local:
        torch.index_select(desc, 0, rows, out=send_d[:k_local])
        torch.index_select(counts, 0, rows, out=send_c[:k_local])
    if out is not None and out[0].shape[0] >= world * k_max and out[0].shape[1:] == send_d.shape[1:] and \
            out[0].device == desc.device:
        all_d, all_c = out[0][:world * k_max], out[1][:world * k_max]
    else:
        all_d = torch.empty((world * k_max, cap, 32), dtype=torch.uint8, device=desc.device)
        all_c = torch.empty(world * k_max, dtype=torch.int32, device=desc.device)
    dist.all_gather_into_tensor(all_d, send_d)
    dist.all_gather_into_tensor(all_c, send_c)
    if all(k == k_max for k in per_rank):
        return all_d, all_c
    keep = torch.as_tensor([r * k_max + i for r, k in enumerate(per_rank) for i in range(k)], dtype=torch.long,
                           device=desc.device)
    return all_d.index_select(0, keep), all_c.index_select(0, keep)


class HipBackend:
    """orbx_match_hamming_cross on device tensors: pair f * K + k of a call = (frame f's rows, key frame k's rows)."""

    def __init__(self, matcher):
        self.matcher = matcher

    def match(self, queries, n_query, db_desc, db_counts, out_idx, out_dist):
        """queries [F, cap, 32], n_query [F]; outputs [F, K, cap] int32 (rows past a frame's count are not written)."""
        F, K, cap = queries.shape[0], db_desc.shape[0], db_desc.shape[1]
        self.matcher.match_hamming_cross(queries.data_ptr(), n_query.data_ptr(), cap * 32, F, db_desc.data_ptr(),
                                         db_counts.data_ptr(), cap * 32, K, cap, cap, out_idx.data_ptr(),
                                         out_dist.data_ptr(), cap)


class NumpyBackend:
    """The same contract on CPU tensors (tests of the multi-rank driver; never the thing measured)."""

    _POP = np.array([bin(i).count("1") for i in range(256)], np.int32)

    def match(self, queries, n_query, db_desc, db_counts, out_idx, out_dist):
        import torch
        for f in range(queries.shape[0]):
            q = queries[f].numpy()
            nq = int(n_query[f])
            for k in range(db_desc.shape[0]):
                nt = int(db_counts[k])
                if nq == 0:
                    continue
                if nt == 0:
                    out_idx[f, k, :nq] = -1
                    out_dist[f, k, :nq] = -1
                    continue
                t = db_desc[k, :nt].numpy()
                d = self._POP[q[:nq, None, :] ^ t[None, :, :]].sum(2)
                best = d.argmin(1)  # first minimum: lowest train index on ties
                out_idx[f, k, :nq] = torch.from_numpy(best.astype(np.int32))
                out_dist[f, k, :nq] = torch.from_numpy(d[np.arange(nq), best].astype(np.int32))


def match_frames_against_database(backend, desc, counts, db_desc, db_counts, frames=None, consume=None, block=8):
    """Matches each local frame (all, or the indices in `frames`) against every key frame of the database, `block`
    frames per call of the backend.  consume(frame, idx [K, cap], dist [K, cap]) is called per frame with buffers that
    are reused by a later block; without it the per-frame results are stacked and returned as ([frames, K, cap],
    [frames, K, cap]), -1 in the rows past a frame's count.  On the GPU the matcher must launch on the stream torch is
    using (Matcher.set_stream)."""
    import torch
    K, cap = db_desc.shape[0], db_desc.shape[1]
    frames = list(range(desc.shape[0])) if frames is None else list(frames)
    block = max(1, min(block, 65535 // max(K, 1)))
    out_idx = torch.zeros((block, K, cap), dtype=torch.int32, device=desc.device)
    out_dist = torch.zeros((block, K, cap), dtype=torch.int32, device=desc.device)
    kept_i, kept_d = [], []
    for b0 in range(0, len(frames), block):
        fb = frames[b0:b0 + block]
        if fb == list(range(fb[0], fb[0] + len(fb))):
            q, nq = desc[fb[0]:fb[0] + len(fb)], counts[fb[0]:fb[0] + len(fb)]   # consecutive frames: views, no copy
        else:
            sel = torch.as_tensor(fb, dtype=torch.long, device=desc.device)
            q, nq = desc.index_select(0, sel), counts.index_select(0, sel)
        if consume is None:
            # rows past a frame's count are not written by the matcher: give them one value for every frame
            out_idx.fill_(-1)
            out_dist.fill_(-1)
        backend.match(q.contiguous(), nq.contiguous(), db_desc, db_counts, out_idx[:len(fb)], out_dist[:len(fb)])
        for j, f in enumerate(fb):
            if consume is not None:
                consume(f, out_idx[j], out_dist[j])
            else:
                kept_i.append(out_idx[j].clone())
                kept_d.append(out_dist[j].clone())
    if consume is not None:
        return None
    return torch.stack(kept_i), torch.stack(kept_d)
