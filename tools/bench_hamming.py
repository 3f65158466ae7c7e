"""Brute-force Hamming, 1024 pairs of 1000 x 1000 descriptors: the matrix-core kernel against the vector kernel
(same results bit for bit), descriptor pairs per second of each.  Run from the repository root on the GPU box."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from orb_slam_amd import orbx
import torch
P, N = 1024, 1000
rng = np.random.default_rng(5)
q = torch.from_numpy(rng.integers(0, 256, (P, N, 32), dtype=np.uint8)).cuda()
t = torch.from_numpy(rng.integers(0, 256, (P, N, 32), dtype=np.uint8)).cuda()
# ties and exact copies: the lowest train index must win in both kernels
t[:, 7] = t[:, 3]
q[:, 11] = t[:, 3]
nq_h = rng.integers(1, N + 1, P).astype(np.int32); nq_h[:4] = N
nt_h = rng.integers(1, N + 1, P).astype(np.int32); nt_h[:4] = N
out = {}
mt = orbx.Matcher()
for name, nq_a, nt_a in (("full", np.full(P, N, np.int32), np.full(P, N, np.int32)), ("ragged", nq_h, nt_h)):
    nq = torch.from_numpy(nq_a).cuda(); nt = torch.from_numpy(nt_a).cuda()
    for kind in (0, 1):
        mt.set_hamming_kernel(kind)
        idx = torch.full((P, N), -7, dtype=torch.int32, device='cuda')
        dist = torch.full((P, N), -7, dtype=torch.int32, device='cuda')
        torch.cuda.synchronize()
        def run():
            mt.match_hamming_batch(q.data_ptr(), nq.data_ptr(), N * 32, t.data_ptr(), nt.data_ptr(), N * 32, P, N, N,
                                   idx.data_ptr(), dist.data_ptr(), N)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 5
        for _ in range(R):
            run()
        torch.cuda.synchronize()   # the matcher launches on its own stream
        dt = (time.perf_counter() - t0) / R
        pairs = float((nq_a.astype(np.int64) * nt_a).sum())
        print('%-6s kernel %d (%s): %.3f ms -> %.3e descriptor pairs/s' % (name, kind, 'matrix cores' if kind == 0 else 'vector', dt * 1e3, pairs / dt))
        out[(name, kind)] = (idx.cpu().numpy(), dist.cpu().numpy())
    a, b = out[(name, 0)], out[(name, 1)]
    m = np.arange(N)[None, :] < nq_a[:, None]
    print(name, 'identical:', bool(np.array_equal(a[0][m], b[0][m]) and np.array_equal(a[1][m], b[1][m])),
          'mismatches', int((a[0][m] != b[0][m]).sum()), int((a[1][m] != b[1][m]).sum()))
