#!/usr/bin/env python3
"""Throughput of the ORB front-end hot path on MI355X.

One step = one pass of the hot path over one batch of synthetic VGA frames resident in HBM:
per frame ORB extract (detect -> undistort -> bucket grid -> compute, the sequence of
MonoFrame::extractFeatures) at 1000 keypoints / 8 levels x 1.2, then the windowed (projection)
match of every frame against its predecessor.  Frames are sharded over ranks with no collective on
the data path (weak scaling).  Prints ONE JSON line on rank 0.

The CPU oracle is used only for the `cpu_baseline` leg (a reported number, never the thing measured).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 640, 480
NKP, NLEVELS, CAP = 1000, 8, 1280
# cfg/camera.yaml `camera:` block of the reference (TUM freiburg1), distortion active
CAM = dict(fx=517.3, fy=516.5, cx=318.6, cy=255.3, dist=[0.2624, -0.9531, -0.0054, 0.0026, 1.1633])
RADIUS = 15.0  # Tracker::trackWithMotionModel, monocular (src/tracker.cpp:304-308)


class DevView:
    """Wraps a raw device pointer for torch.as_tensor (no copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}


def make_sequence(torch, base_bgr, n_frames, first_index, device):
    """Smooth random homography walk over the reference test image + Gaussian noise, rendered on the GPU.
    Returns uint8 [n, H, W, 3] frames and the 3x3 maps H_f with x_f = H_f x_base (float64, host)."""
    rng = np.random.default_rng(2)
    total = first_index + n_frames
    Hs = []
    a = np.zeros(8)
    for f in range(total):
        a = 0.98 * a + rng.normal(0, 1, 8) * np.array([2e-4, 2e-4, 0.15, 2e-4, 2e-4, 0.15, 2e-7, 2e-7])
        Hm = np.array([[1 + a[0], a[1], a[2] * 4], [a[3], 1 + a[4], a[5] * 4], [a[6], a[7], 1.0]])
        Hs.append(Hm)
    Hs = np.stack(Hs[first_index:])
    base = torch.from_numpy(base_bgr.astype(np.float32)).to(device).permute(2, 0, 1)[None]
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    ones = torch.ones_like(xs)
    pix = torch.stack([xs, ys, ones], 0).reshape(3, -1)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + first_index)
    out = torch.empty((n_frames, H, W, 3), dtype=torch.uint8, device=device)
    for f in range(n_frames):
        Hinv = torch.from_numpy(np.linalg.inv(Hs[f]).astype(np.float32)).to(device)
        src = Hinv @ pix
        sx = src[0] / src[2]
        sy = src[1] / src[2]
        grid = torch.stack([sx / (W - 1) * 2 - 1, sy / (H - 1) * 2 - 1], -1).reshape(1, H, W, 2)
        img = torch.nn.functional.grid_sample(base, grid, mode="bilinear", padding_mode="reflection",
                                              align_corners=True)
        img = img + torch.randn(img.shape, generator=gen, device=device) * 2.0
        out[f] = img.clamp(0, 255).round().to(torch.uint8)[0].permute(1, 2, 0)
    return out.contiguous(), Hs


def algorithmic_bytes(n_kp):
    """Per-frame algorithmic HBM bytes of the extraction stages (VGA colour, 8 levels x 1.2)."""
    lv = [(int(round(W / 1.2 ** l)), int(round(H / 1.2 ** l))) for l in range(NLEVELS)]
    P = sum(w * h for w, h in lv)
    per_kernel = {
        "gray": 3 * W * H + W * H,                      # read BGR, write level 0
        "resize": sum(lv[l - 1][0] * lv[l - 1][1] + lv[l][0] * lv[l][1] for l in range(1, NLEVELS)),
        "fast": P + P // 8,                             # fused FAST + NMS: read pyramid, write 1 bit / pixel
        "emit": P // 8,                                 # read mask
        "harris": 2 * n_kp * 81,                        # 9x9 per candidate above the retainBest(2 x quota) score
        "angle": n_kp * 709,                            # radius-15 disc per kept keypoint
        "blur": 2 * P,                                  # read pyramid, write blurred pyramid
        "brief": n_kp * (512 + 32),                     # 512 samples + 32 descriptor bytes per keypoint
    }
    frame_total = 3 * W * H + 4 * P + n_kp * (2 * 81 + 709 + 512)
    return per_kernel, frame_total


def cpu_baseline(frames_bgr, homs, grid_bounds, seconds=12.0, threads=16):
    """CPU oracle (scalar C++ port of the reference path) on a bounded sample of the same workload: frames in
    parallel on `threads` host threads (the oracle's C calls release the GIL), and one thread alone beside it."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as O
    p = O.orb_params(nfeatures=NKP, nlevels=NLEVELS)
    cam = O.camera(CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"], CAM["dist"], W, H)
    g = O.grid_params(grid_bounds)
    _, sf, _, _, _ = O.scale_tables(1.2, NLEVELS)

    def extract(f):
        k = O.detect(p, frames_bgr[f])
        xy = O.undistort_points(cam, np.stack([k["x"], k["y"]], 1))
        k["x"], k["y"] = xy[:, 0], xy[:, 1]
        k, start, items = O.assign_grid(g, k)
        k, d = O.compute(p, frames_bgr[f], k)
        return k, d, start, items

    def match(f, cur, prev):
        k, d, start, items = cur
        pk, pd = prev[0], prev[1]
        q = (homs[f] @ np.linalg.inv(homs[f - 1]) @ np.c_[pk["x"], pk["y"], np.ones(len(pk))].T).T
        proj = (q[:, :2] / q[:, 2:]).astype(np.float32)
        return O.match_projection_frame(g, start, items, k, d, pk, np.ones(len(pk), np.uint8), proj,
                                        np.ones(len(pk), np.float32), pd, sf, 0, RADIUS, True)

    # one thread, a few frames
    t0 = time.perf_counter()
    n1, prev = 0, None
    while n1 < len(frames_bgr) and (n1 < 4 or time.perf_counter() - t0 < seconds / 4):
        cur = extract(n1)
        if prev is not None:
            match(n1, cur, prev)
        prev = cur
        n1 += 1
    single = n1 / (time.perf_counter() - t0)
    # all threads: as many frames as the remaining time allows, in blocks
    threads = max(1, min(threads, os.cpu_count() or 1))
    block = 4 * threads
    done, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(threads) as pool:
        last = None
        while done < len(frames_bgr) and (done == 0 or time.perf_counter() - t0 < seconds * 3 / 4):
            idx = list(range(done, min(done + block, len(frames_bgr))))
            ex = list(pool.map(extract, idx))
            pairs = [(f, ex[i], ex[i - 1] if i else last) for i, f in enumerate(idx)]
            list(pool.map(lambda a: match(*a) if a[2] is not None else None, pairs))
            last = ex[-1]
            done += len(idx)
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "single_thread_value": single,
            "sample": f"{done} frames of the same synthetic VGA sequence on {threads} threads ({n1} frames on one), "
                      f"detect+undistort+grid+compute+windowed match, CPU oracle -O2"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=792,
                    help="frames per GPU per step (792 = one TUM fr1/xyz-length sequence)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ransac-pairs", type=int, default=1024)
    ap.add_argument("--streams", type=int, default=3,
                    help="independent batches in flight per GPU, each on a HIP stream of its own (high priority unless "
                         "ORBX_BENCH_PRIORITIES says otherwise)")
    ap.add_argument("--kfdb-frames", type=int, default=1250,
                    help="frames per GPU of the key-frame database leg (0 = skip): every 10th is a key frame, the "
                         "ranks all-gather their key frames' descriptors and match every frame against all of them")
    ap.add_argument("--frames-4k", type=int, default=32,
                    help="frames per GPU of the 4K extraction leg (3840x2160 gray, 8000 kp, 12 levels; 0 = skip)")
    ap.add_argument("--fed-frames", type=int, default=1024,
                    help="frames of the host-fed leg (PCIe-inclusive extraction rate, N = 1 only; 0 = skip)")
    ap.add_argument("--pyramid-kernel", type=int, default=0,
                    help="0 / 1: level by level (the default), 2: the one-pass kernel")
    ap.add_argument("--pyramid-rows", type=int, default=0, help="image rows per step of the one-pass pyramid kernel")
    ap.add_argument("--distribution", type=int, default=0,
                    help="1: the opt-in per-cell adaptive threshold + quad-tree distribution (low threshold 7, cells of "
                         "30 px) instead of cv::ORB's selection; not the headline configuration")
    ap.add_argument("--pyramid-tail", type=int, default=0,
                    help="levels above this one in one launch (orbx_extractor_set_pyramid_tail; 0 = level by level)")
    ap.add_argument("--pyramid-tail-rows", type=int, default=0)
    ap.add_argument("--profile-round", default="r03", help="prefix of the PMC summaries under profiles/ to quote")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from orb_slam_amd import orbx

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ORBX_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("ORBX_BENCH_BACKEND", "nccl")
    backend_is_nccl = backend == "nccl"
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    red_device = device if backend == "nccl" else torch.device("cpu")
    B = args.frames
    base = np.load(os.path.join(ROOT, "tests", "golden", "test_images.npz"))["bgr1"]
    frames, homs = make_sequence(torch, base, B, rank * B, device)
    torch.cuda.synchronize()

    cam = orbx.make_camera(CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"], CAM["dist"])
    # Camera::computeImageBounds (camera.cpp:129-162): undistorted corners, clipped to the image
    corners = orbx.undistort_points(cam, np.array([[0, 0], [W, 0], [0, H], [W, H]], np.float32))
    bounds = (max(min(corners[0, 0], corners[2, 0]), 0.0), min(max(corners[1, 0], corners[3, 0]), float(W)),
              max(min(corners[0, 1], corners[1, 1]), 0.0), min(max(corners[2, 1], corners[3, 1]), float(H)))
    grid = orbx.make_grid(bounds)
    n_pairs = B - 1
    frame_idx = torch.arange(1, B, dtype=torch.int32, device=device)
    ref_idx = torch.arange(0, B - 1, dtype=torch.int32, device=device)
    # x_f = H_f H_{f-1}^-1 x_{f-1}: the known inter-frame map stands in for pose + depth of the map points
    rel = torch.from_numpy(np.stack([homs[f] @ np.linalg.inv(homs[f - 1]) for f in range(1, B)])
                           .astype(np.float32)).to(device)
    torch.cuda.synchronize()

    stage_ms = {}
    match_ms = []

    class Pipeline:
        """One in-flight batch: its own HIP stream (torch's plumbing and the library's kernels share it; a NULL
        stream handle means "the handle's own stream" to orbx_*_set_stream), extractor, matcher and outputs.
        Batches are independent, so several pipelines keep the GPU busy while one of them sits in a
        latency-bound kernel (ordered emission, selection, the greedy resolve) or in its end-of-batch hand-off."""

        def __init__(self, priority=0):
            self.tstream = torch.cuda.Stream(device=device, priority=priority)
            self.ex = orbx.Extractor(n_key_points=NKP, level_pyramid=NLEVELS, max_width=W, max_height=H, max_batch=B,
                                     device=local_rank)
            self.ex.set_frame_model(cam, grid)
            self.ex.set_stream(self.tstream.cuda_stream)
            # with one batch in flight the blur runs beside the latency-bound middle of its own batch; with several, the
            # other batches fill those gaps and the blur runs in line
            self.ex.set_blur_fork(max(1, args.streams) == 1)
            if args.pyramid_kernel or args.pyramid_rows:
                self.ex.set_pyramid_kernel(args.pyramid_kernel, args.pyramid_rows)
            if args.pyramid_tail:
                self.ex.set_pyramid_tail(args.pyramid_tail, args.pyramid_tail_rows)
            if args.distribution:
                self.ex.set_distribution(args.distribution, 7, 30)
            self.mt = orbx.Matcher(device=local_rank)
            self.mt.set_stream(self.tstream.cuda_stream)
            self.sf = self.ex.scale_tables()[1]
            self.d_matched = torch.empty((n_pairs, CAP), dtype=torch.int32, device=device)
            self.d_dist = torch.empty((n_pairs, CAP), dtype=torch.int32, device=device)
            self.d_nm = torch.zeros(n_pairs, dtype=torch.int32, device=device)

        def step(self, record=False):
            ex, mt = self.ex, self.mt
            with torch.cuda.stream(self.tstream):
                # queued, not waited for: the matcher below consumes the device results on the same stream
                ex.extract_batch(frames.data_ptr(), B, W * H * 3, W, H, W * 3, 3, CAP, to_host=False, deferred=True)
                kp_ptr, desc_ptr, cnt_ptr, cap = ex.device_results()
                cs_ptr, ci_ptr = ex.device_grid()
                kp = torch.as_tensor(DevView(kp_ptr, (B, cap, 7), "<f4"), device=device)
                xy1 = torch.cat([kp[:-1, :, 0:2], torch.ones((B - 1, cap, 1), device=device)], 2)
                q = torch.bmm(xy1, rel.transpose(1, 2))
                proj = (q[:, :, :2] / q[:, :, 2:]).contiguous()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                mt.match_projection_frame_batch(grid, n_pairs, frame_idx.data_ptr(), ref_idx.data_ptr(), kp_ptr,
                                                desc_ptr, cnt_ptr, cap, cs_ptr, ci_ptr, 0, proj.data_ptr(), 0, self.sf,
                                                0, RADIUS, True, self.d_matched.data_ptr(), self.d_dist.data_ptr(),
                                                self.d_nm.data_ptr())
                e1.record()
                self.tstream.synchronize()  # the batch's results are complete when step() returns
                _, redone = ex.finish()     # status of the extraction (capacity, host-selection fallback)
                if redone:
                    raise RuntimeError("a frame fell back to the host selection: the match above ran on stale keypoints")
                if record:
                    for k, v in ex.last_timings().items():
                        stage_ms.setdefault(k, []).append(v)
                    match_ms.append((e0, e1))

    # Three batches on high-priority streams.  Measured on 792-frame VGA batches (ms per step, several runs each; the HIP
    # runtime spreads its streams over GPU_MAX_HW_QUEUES = 4 hardware queues by default): two default-priority streams
    # 3.42-3.60 depending on which queues they land on, three 3.56, three high-priority ones 3.34-3.44; with 3 hardware
    # queues two or three default streams give 3.38-3.48, with 1 or 2 queues every kernel runs alone (3.85), with 8 or 16
    # every stream has a queue of its own and the step takes 3.7-4.05: two kernels sharing the chip with a third
    # queued behind them without a gap is the arrangement that pays, free concurrency of everything is not.
    # (ORBX_BENCH_PRIORITIES="0,0,0": stream priority per pipeline, for that experiment.)
    prios = [int(v) for v in os.environ.get("ORBX_BENCH_PRIORITIES", "").split(",") if v.strip()]
    pipes = [Pipeline(prios[i] if i < len(prios) else -1) for i in range(max(1, args.streams))]

    def run_steps(n, record):
        """n steps in all, dealt round-robin to the pipelines; each pipeline is driven by its own host thread
        (the library calls release the GIL)."""
        import threading
        if len(pipes) == 1:
            for _ in range(n):
                pipes[0].step(record)
            return
        def work(pi):
            torch.cuda.set_device(local_rank)
            for _ in range(pi, n, len(pipes)):
                pipes[pi].step(record)
        th = [threading.Thread(target=work, args=(pi,)) for pi in range(len(pipes))]
        for t in th:
            t.start()
        for t in th:
            t.join()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup * len(pipes), False)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the same stage timings with one batch alone on the GPU (after the timed region, not part of `value`): with
    # several batches in flight a kernel shares the CUs with the other batch's kernels and its own duration stretches
    timed_stage_ms, timed_match_ms = stage_ms, match_ms
    alone_avg = None
    if len(pipes) > 1:
        stage_ms, match_ms = {}, []
        for _ in range(3):
            pipes[0].step(True)
        torch.cuda.synchronize()
        alone_avg = {k: float(np.mean(v)) for k, v in stage_ms.items()}
        alone_avg["match_window"] = float(np.mean([a.elapsed_time(b) for a, b in match_ms]))
        stage_ms, match_ms = timed_stage_ms, timed_match_ms

    ex, d_nm = pipes[0].ex, pipes[0].d_nm
    stream = pipes[0].tstream.cuda_stream
    torch.cuda.set_stream(pipes[0].tstream)
    mean_matches = float(d_nm.float().mean().item())
    mean_kp = float(torch.as_tensor(DevView(ex.device_results()[2], (B,), "<i4"), device=device).float().mean().item())
    total_frames = world * B * args.steps
    value = total_frames / dt

    per_kernel, frame_bytes = algorithmic_bytes(NKP)
    stage_avg = {k: float(np.mean(v)) for k, v in stage_ms.items()}
    stage_avg["match_window"] = float(np.mean([a.elapsed_time(b) for a, b in match_ms]))
    # launches per stage: the pyramid stage is one launch per level above 0, every other stage is one kernel launch
    launches = {k: (NLEVELS - 1 if k == "resize" else 1) for k in per_kernel}
    # the blur runs on the extractor's side stream as a filler beside emission / Harris / selection: its bracketed
    # duration is stretched by them (0.53 ms by itself, about 1.1 ms there), so it does not stand for the dominant kernel
    dom = max((k for k in stage_avg if k in per_kernel and k != "blur"), key=lambda k: stage_avg[k] / launches[k])
    achieved = per_kernel[dom] * B / (stage_avg[dom] * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes kept under profiles/ (separate rocprofv3 --pmc runs of this command)
    traffic = None
    kernel_of = {"gray": "k_gray_level0", "resize": "k_resize_strip", "fast": "k_fast_nms", "emit": "k_emit",
                 "harris": "k_harris", "angle": "k_angle_kept", "blur": "k_blur", "brief": "k_brief"}
    try:
        with open(os.path.join(ROOT, "profiles", args.profile_round + "_pmc_traffic.json")) as f:
            pmc = json.load(f).get(kernel_of[dom])
        if pmc and pmc.get("frames_per_launch") == B:
            traffic = pmc["traffic_bytes"] * launches[dom]
    except (OSError, ValueError, KeyError):
        pass
    roofline = {"bound": "hbm", "kernel": kernel_of[dom], "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": traffic,
                "algorithmic_bytes_per_launch": per_kernel[dom] * B,
                "stage_ms_per_batch": stage_avg, "algorithmic_bytes_per_frame": frame_bytes,
                "device_ms_per_batch": float(sum(stage_avg.values()))}
    if alone_avg:
        # With several batches in flight a HIP-event bracket around a launch also holds the time the launch waits behind
        # the other batches' kernels on the hardware queues, and the kernel itself is stretched by whatever shares the
        # chip with it: that figure says how the schedule treats the kernel, not how close the kernel is to its
        # roofline.  `achieved` / `frac` are therefore taken from the same batch run alone right after the timed region
        # (HIP events on the library's stream; `profiles/*_kernel_stats_streams1.csv` is the rocprofv3 view of the same
        # thing); the bracket inside the timed region is kept beside it.
        roofline["stage_ms_per_batch_alone"] = alone_avg
        alone = per_kernel[dom] * B / (alone_avg[dom] * 1e-3) / 1e9
        roofline["timed_region"] = {"achieved": achieved, "frac": achieved / 8000.0,
                                    "note": "HIP-event bracket of the launch with the other batches in flight"}
        roofline["achieved"] = alone
        roofline["frac"] = alone / 8000.0
        roofline["frac_alone"] = alone / 8000.0
    # what the dominant kernel's instruction pipes do while it runs alone: the counters of the PMC passes kept under
    # profiles/ (per launch of the same batch size) against this run's duration of the kernel alone.  Nominal
    # capacities: 1024 SIMDs issuing one vector instruction per 2.5 cycles (the probed full rate), one scalar unit and
    # one LDS per CU (256), 2.4 GHz.
    try:
        rd = lambda name: json.load(open(os.path.join(ROOT, "profiles", args.profile_round + name))).get(kernel_of[dom], {})
        sq_i, sq_m = rd("_pmc_sq.json"), rd("_pmc_sq_mem.json")
        ms_alone = (alone_avg or stage_avg)[dom] / launches[dom]
        cyc = ms_alone * 1e-3 * 2.4e9
        roofline["pipes_alone"] = {
            "vector_issue": sq_i["SQ_INSTS_VALU_per_launch"] * 2.5 / (cyc * 1024),
            "scalar_issue": sq_i["SQ_INSTS_SALU_per_launch"] / (cyc * 256),
            "lds_active": sq_m["SQ_LDS_IDX_ACTIVE_per_launch"] / (cyc * 256),
            "lds_bank_conflict_share": sq_m["SQ_LDS_BANK_CONFLICT_per_launch"] / max(sq_m["SQ_LDS_IDX_ACTIVE_per_launch"], 1.0),
            "note": "fractions of each pipe's cycles while the kernel runs alone; no pipe saturated means the kernel "
                    "waits on the dependencies between its phases"}
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        pass

    # RANSAC hypotheses/s (second half of the headline metric): 200 iterations x 2000 correspondences per pair
    ransac = None
    if args.ransac_pairs > 0:
        P, n, it = args.ransac_pairs, 2000, 200
        rng = np.random.default_rng(3 + rank)
        ref = np.stack([rng.uniform(0, W, (P, n)), rng.uniform(0, H, (P, n))], 2).astype(np.float32)
        Hm = np.array([[1.01, 0.01, 3.0], [-0.01, 0.99, -2.0], [1e-5, -1e-5, 1.0]])
        q = ref.astype(np.float64) @ Hm[:2, :2].T + Hm[:2, 2]
        w = ref.astype(np.float64) @ Hm[2, :2] + 1.0
        pts = (q / w[..., None] + rng.normal(0, 1.0, (P, n, 2))).astype(np.float32)
        out = rng.uniform(size=(P, n)) < 0.3
        pts[out] = np.stack([rng.uniform(0, W, out.sum()), rng.uniform(0, H, out.sum())], 1).astype(np.float32)
        sets = np.stack([np.stack([rng.choice(n, 8, replace=False) for _ in range(it)]) for _ in range(min(P, 64))])
        sets = np.tile(sets, (P // len(sets) + 1, 1, 1))[:P].astype(np.int32)
        d_pts = torch.from_numpy(np.ascontiguousarray(pts)).to(device)
        d_ref = torch.from_numpy(np.ascontiguousarray(ref)).to(device)
        d_sets = torch.from_numpy(np.ascontiguousarray(sets)).to(device)
        d_F = torch.empty((P, 9), dtype=torch.float32, device=device)
        d_H = torch.empty((P, 9), dtype=torch.float32, device=device)
        d_if = torch.empty((P, n), dtype=torch.uint8, device=device)
        d_ih = torch.empty((P, n), dtype=torch.uint8, device=device)
        d_sc = torch.empty((P, 2), dtype=torch.float32, device=device)
        d_b = torch.empty((P, 2), dtype=torch.int32, device=device)
        rs = orbx.Ransac(device=local_rank)
        rs.set_stream(stream)

        def rstep():
            rs.fh_batch(d_pts.data_ptr(), d_ref.data_ptr(), n, d_sets.data_ptr(), it, P, 1.0, d_F.data_ptr(),
                        d_H.data_ptr(), d_if.data_ptr(), d_ih.data_ptr(), d_sc.data_ptr(), d_b.data_ptr())
        rstep()
        barrier()
        r0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            rstep()
        barrier()
        rdt = time.perf_counter() - r0
        if world > 1:
            t = torch.tensor([rdt], dtype=torch.float64, device=red_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rdt = float(t.item())
        ransac = {"hypotheses_per_s": world * P * it * 2 * reps / rdt, "pairs": P, "iterations": it,
                  "correspondences": n, "stage_ms": rs.last_timings(),
                  "mean_inliers_h": float(d_ih.float().sum(1).mean().item())}
        # the same kernels one after the other on one stream (after the timed region): each has the chip to itself,
        # which is the duration a roofline fraction of a single kernel needs
        rs.set_fork(False)
        rstep()
        rstep()
        torch.cuda.synchronize()
        ransac["stage_ms_alone"] = rs.last_timings()
        rs.set_fork(True)

    # ---- issue-rate rooflines of the two halves that are bound by the vector pipes, not by HBM ----------------------
    # Peaks: nominal = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz lane-operations/s for single-cycle-class instructions
    # (half of it for fp64); measured = this GPU's sustained rate for the very instruction mix, taken now by
    # orbx_probe_issue_rate (8 wavefronts per SIMD, independent chains).
    lanes_nominal = 256 * 4 * 32 * 2.4e9
    probe = {k: orbx.probe_issue_rate(i, local_rank) * 64 for i, k in
             enumerate(["add_u32", "xor_bcnt", "mul_add_f64", "mul_add_f32"])}
    sq = {}
    try:
        with open(os.path.join(ROOT, "profiles", args.profile_round + "_pmc_sq.json")) as f:
            sq = json.load(f)
    except (OSError, ValueError):
        pass
    roofline_ransac = None
    if ransac:
        P, n, it = ransac["pairs"], ransac["correspondences"], ransac["iterations"]
        st = ransac["stage_ms_alone"]
        # lane-operations of the scorers per (hypothesis, correspondence): Initializer.cpp:410-438 (48 fp64 multiplies,
        # adds and compares + 2 fp64 divisions) and :552-577 (44 fp32 operations + 2 divisions)
        def entry(kernel, ms, ops, nominal, measured):
            e = {"kernel": kernel, "bound": "valu", "ms": ms, "algorithmic_lane_ops": ops,
                 "achieved": ops / (ms * 1e-3) / 1e12, "peak": nominal / 1e12, "peak_measured": measured / 1e12,
                 "unit": "T lane-ops/s", "frac": ops / (ms * 1e-3) / nominal,
                 "frac_of_measured": ops / (ms * 1e-3) / measured,
                 "note": "duration of the kernel alone on the chip (fundamental and homography chains one after the "
                         "other); in the timed run the two chains share the chip on two streams"}
            k = sq.get(kernel)
            if k and "SQ_INSTS_VALU_per_launch" in k:
                # every issued vector instruction (address arithmetic, divisions' expansions, LDS traffic included)
                e["valu_wave_instructions_per_launch"] = k["SQ_INSTS_VALU_per_launch"]
                e["issue_frac_of_measured"] = k["SQ_INSTS_VALU_per_launch"] * 64 / (ms * 1e-3) / measured
            return e
        roofline_ransac = {
            "score_f_fp64": entry("k_score_f", st["score_f"], float(P) * it * n * 48, lanes_nominal / 2,
                                  probe["mul_add_f64"]),
            "score_h_fp32": entry("k_score_h", st["score_h"], float(P) * it * n * 44, lanes_nominal,
                                  probe["mul_add_f32"]),
        }

    # brute-force Hamming (cv::BFMatcher of Frame::match): 1024 pairs of 1000 x 1000 descriptors resident in HBM
    mt = pipes[0].mt
    HP, HN = 1024, 1000
    gq = torch.Generator(device=device)
    gq.manual_seed(99)
    hq = torch.randint(0, 256, (HP, HN, 32), dtype=torch.uint8, device=device, generator=gq)
    ht = torch.randint(0, 256, (HP, HN, 32), dtype=torch.uint8, device=device, generator=gq)
    hn = torch.full((HP,), HN, dtype=torch.int32, device=device)
    hidx = torch.empty((HP, HN), dtype=torch.int32, device=device)
    hdist = torch.empty((HP, HN), dtype=torch.int32, device=device)

    def hstep():
        mt.match_hamming_batch(hq.data_ptr(), hn.data_ptr(), HN * 32, ht.data_ptr(), hn.data_ptr(), HN * 32, HP, HN,
                               HN, hidx.data_ptr(), hdist.data_ptr(), HN)
    def htime(kind):
        mt.set_hamming_kernel(kind)
        hstep()
        he0, he1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        he0.record()
        for _ in range(5):
            hstep()
        he1.record()
        torch.cuda.synchronize()
        return he0.elapsed_time(he1) / 5
    hms_v = htime(1)   # the vector kernel (xor + popcount)
    hms = htime(0)     # the matrix-core kernel (the default): |query| + int8 dot product of {0,1} and {+1,-1} bits
    hpairs = float(HP) * HN * HN
    hops = hpairs * 16  # vector form: 8 xor + 8 popcount-accumulate per descriptor pair
    int8_peak = 5000.0  # T int8 ops/s dense: twice the bf16 rate of 2.5 PFLOP/s (v_mfma_i32_32x32x32_i8, 32 cycles)
    roofline_match = {"kernel": "k_bf_hamming_mfma", "bound": "mfma", "ms": hms, "pairs_per_s": hpairs / (hms * 1e-3),
                      "achieved": hpairs * 512 / (hms * 1e-3) / 1e12, "peak": int8_peak, "unit": "T int8 ops/s",
                      "frac": hpairs * 512 / (hms * 1e-3) / 1e12 / int8_peak,
                      "workload": f"{HP} pairs of {HN} x {HN} descriptors, 256 multiply-adds per descriptor pair",
                      "note": "two vector instructions per descriptor pair and lane (key = distance << 16 | row, running "
                              "minimum) share the SIMD's issue with the products; they, not the matrix pipe, set the "
                              "pace",
                      "vector_kernel": {
                          "kernel": "k_bf_hamming", "bound": "valu", "ms": hms_v, "pairs_per_s": hpairs / (hms_v * 1e-3),
                          "achieved": hops / (hms_v * 1e-3) / 1e12, "peak": lanes_nominal / 1e12,
                          "peak_measured": probe["xor_bcnt"] / 1e12, "unit": "T lane-ops/s",
                          "frac": hops / (hms_v * 1e-3) / lanes_nominal,
                          "frac_of_measured": hops / (hms_v * 1e-3) / probe["xor_bcnt"],
                          "note": "16 lane-operations per descriptor pair; v_bcnt_u32_b32 issues at half the rate of "
                                  "v_xor_b32 on gfx950 (probe): the measured peak is the sustained rate of the 1:1 "
                                  "xor/popcount mix"}}
    del hq, ht, hidx, hdist

    # ---- key-frame database leg (BASELINE config 5): the one step of the path with a collective -----------------------
    kfdb_line = None
    if args.kfdb_frames > 0:
        from orb_slam_amd import kfdb
        KF = args.kfdb_frames
        kframes, _ = make_sequence(torch, base, KF, rank * KF, device)
        exk = orbx.Extractor(n_key_points=NKP, level_pyramid=NLEVELS, max_width=W, max_height=H, max_batch=KF,
                             device=local_rank)
        exk.set_frame_model(cam, grid)
        exk.set_stream(stream)
        mt.set_stream(stream)
        exk.extract_batch(kframes.data_ptr(), KF, W * H * 3, W, H, W * 3, 3, CAP, to_host=False)
        kp_ptr, desc_ptr, cnt_ptr, cap = exk.device_results()
        kdesc = torch.as_tensor(DevView(desc_ptr, (KF, cap, 32), "|u1"), device=device)
        kcnt = torch.as_tensor(DevView(cnt_ptr, (KF,), "<i4"), device=device)
        rows = kfdb.keyframe_rows(rank * KF, KF, 10)
        backend = kfdb.HipBackend(mt)
        # one untimed block against the local key frames: the matcher's workspace is allocated on first use
        kfdb.match_frames_against_database(backend, kdesc, kcnt, kdesc[rows].contiguous(), kcnt[rows].contiguous(),
                                           frames=range(min(8, KF)), consume=lambda f, i, d: None)
        # one untimed gather: the first collective on a process group sets up its communicator, and the first
        # index_select of this shape allocates - neither belongs to the all-gather time reported below
        gather_out = kfdb.gather_database(kdesc, kcnt, rows, dist if world > 1 else None)
        barrier()
        k0 = time.perf_counter()
        db_desc, db_counts = kfdb.gather_database(kdesc, kcnt, rows, dist if world > 1 else None, out=gather_out)
        torch.cuda.synchronize()
        k1 = time.perf_counter()
        kfdb.match_frames_against_database(backend, kdesc, kcnt, db_desc, db_counts, consume=lambda f, i, d: None)
        barrier()
        k2 = time.perf_counter()
        times = torch.tensor([k1 - k0, k2 - k0], dtype=torch.float64, device=red_device)
        work = torch.tensor([float(kcnt.double().sum().item())], dtype=torch.float64, device=red_device)
        if world > 1:
            dist.all_reduce(times, op=dist.ReduceOp.MAX)
            dist.all_reduce(work, op=dist.ReduceOp.SUM)
        db_rows = float(db_counts.double().sum().item())
        kfdb_line = {"frames_per_gpu": KF, "key_frames_total": int(db_desc.shape[0]),
                     "database_bytes": int(db_desc.numel()),
                     "all_gather_ms": float(times[0].item()) * 1e3, "total_ms": float(times[1].item()) * 1e3,
                     "descriptor_pairs_per_s": float(work.item()) * db_rows / float(times[1].item()),
                     "collective": "all_gather (RCCL)" if world > 1 and backend_is_nccl else
                                   ("all_gather (gloo rehearsal)" if world > 1 else "none at 1 GPU"),
                     "scaling_note": "frames and key frames per GPU fixed; the database every GPU matches against "
                                     "grows with the number of GPUs"}
        del kframes, exk

    # ---- 4K extraction leg (BASELINE config 4): 3840x2160 gray, 8000 keypoints, 12 levels, extraction only -------------
    line_4k = None
    if args.frames_4k > 0:
        W4, H4, B4, CAP4, L4 = 3840, 2160, args.frames_4k, 9000, 12
        g4 = torch.Generator(device=device)
        g4.manual_seed(4 + rank)
        distinct = []
        for _ in range(min(4, B4)):   # rectangles of random gray on a flat ground + noise: corners at every scale
            img = torch.full((H4, W4), 100.0, device=device)
            geo = torch.randint(0, 1 << 30, (6000, 5), generator=g4, device=device).cpu().numpy()
            for x0, y0, ww, hh, v in geo:
                x0, y0, ww, hh = int(x0 % W4), int(y0 % H4), int(3 + ww % 397), int(3 + hh % 297)
                img[y0:y0 + hh, x0:x0 + ww] = float(v % 256)
            img = img + torch.randn((H4, W4), generator=g4, device=device) * 2.5
            distinct.append(img.clamp(0, 255).round().to(torch.uint8))
        f4 = torch.stack([distinct[i % len(distinct)] for i in range(B4)]).contiguous()
        ex4 = orbx.Extractor(n_key_points=8000, level_pyramid=L4, max_width=W4, max_height=H4, max_batch=B4,
                             device=local_rank)
        ex4.set_stream(stream)

        def step4():
            ex4.extract_batch(f4.data_ptr(), B4, W4 * H4, W4, H4, W4, 1, CAP4, to_host=False, deferred=True)
            torch.cuda.synchronize()
            return ex4.finish()
        step4()
        barrier()
        q0 = time.perf_counter()
        reps4 = 3
        for _ in range(reps4):
            cnt4, redone4 = step4()
        barrier()
        dt4 = (time.perf_counter() - q0) / reps4
        if world > 1:
            t = torch.tensor([dt4], dtype=torch.float64, device=red_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt4 = float(t.item())
        lv4 = [(int(round(W4 / 1.2 ** l)), int(round(H4 / 1.2 ** l))) for l in range(L4)]
        P4 = sum(a * b for a, b in lv4)
        bytes4 = W4 * H4 + 4 * P4 + 8000 * (2 * 81 + 709 + 512)   # gray input: read once, pyramid written and read twice
        line_4k = {"workload": "3840x2160 gray, 8000 kp, 12 levels x1.2, detect + compute, frames resident in HBM",
                   "frames_per_gpu_per_batch": B4, "frames_per_s": world * B4 / dt4, "ms_per_batch": dt4 * 1e3,
                   "mean_keypoints": float(np.mean(cnt4)), "selection_on_device": not redone4,
                   "algorithmic_bytes_per_frame": bytes4,
                   "achieved_GBps": bytes4 * B4 / dt4 / 1e9, "frac_of_8TBps": bytes4 * B4 / dt4 / 8e12,
                   "stage_ms_per_batch": ex4.last_timings()}
        del f4, ex4, distinct

    # ---- host-fed leg: the same frames start in pinned HOST memory; the feeder uploads one batch ahead on its own
    # stream while the previous batch is extracted.  Extraction only.  This is the PCIe-inclusive rate, never `value`.
    fed_line = None
    if args.fed_frames > 0 and world == 1:
        FB = 256
        nb = max(1, args.fed_frames // FB)
        host = frames[:FB].cpu().repeat((nb, 1, 1, 1)).pin_memory()
        exf = orbx.Extractor(n_key_points=NKP, level_pyramid=NLEVELS, max_width=W, max_height=H, max_batch=FB,
                             device=local_rank)
        fstream = torch.cuda.Stream(device=device)
        exf.set_stream(fstream.cuda_stream)
        fd = orbx.Feeder(W, H, 3, FB, n_slots=3)

        def run_fed():
            fd.start_memory(host.numpy(), already_pinned=True)
            n = 0
            while True:
                ptr, k = fd.next(fstream.cuda_stream)
                if k == 0:
                    break
                exf.extract_batch(ptr, k, W * H * 3, W, H, W * 3, 3, CAP, to_host=False)
                n += k
            fstream.synchronize()
            return n
        run_fed()
        q0 = time.perf_counter()
        n_fed = run_fed()
        dtf = time.perf_counter() - q0
        fed_line = {"workload": "640x480 BGR frames in pinned host memory -> upload (double-buffered) -> detect + compute",
                    "frames": n_fed, "frames_per_s": n_fed / dtf, "bus_GBps": n_fed * W * H * 3 / dtf / 1e9}
        fd.close()
        del host, exf, fd

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # up to the whole batch, cut off after about 12 s of CPU work
        cpu = cpu_baseline(frames.cpu().numpy(), homs, bounds)

    if rank == 0:
        line = {
            "metric": "frames/s ORB extract+match @640x480/1k kp; RANSAC hyp/s; 1-8 GPU scaling",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "TUM-fr1-like monocular 640x480 BGR sequence, one sequence per step (homography walk of the reference "
                                   "test image + noise), 1000 kp, 8 levels x1.2: per-frame extract (detect, "
                                   "undistort, 64x64 grid, compute) + windowed match (radius 15) to the previous frame",
                       "frames_per_gpu_per_step": B, "batches_in_flight": len(pipes),
                       "stream_priorities": [(prios[i] if i < len(prios) else -1) for i in range(len(pipes))],
                       "mean_keypoints": mean_kp,
                       "keypoint_selection": "adaptive threshold + quad-tree (opt-in)" if args.distribution else "cv::ORB",
                       "mean_window_matches": mean_matches,
                       "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
            "roofline": roofline, "roofline_ransac": roofline_ransac, "roofline_match": roofline_match,
            "ransac": ransac, "extra": {"key_frame_database": kfdb_line, "extraction_4k": line_4k, "fed_from_host_memory": fed_line, "issue_rate_probe_lane_ops_per_s": probe},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
