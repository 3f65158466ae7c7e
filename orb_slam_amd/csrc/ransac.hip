// Two-view F/H RANSAC behind orbx_ransac_fh / orbx_ransac_fh_batch.
// Replaces Initializer::findFundamentalAndHomography and its callees (reference src/Initializer.cpp:254-585,
// src/geometry/utils.cpp:18-154).
#include <algorithm>
#include <mutex>

#include "common.h"
#include "ransac_kernels.h"
#include "pose_kernels.h"

using namespace orbx;

namespace {
enum Stage { RT_NORMALIZE = 0, RT_SOLVE, RT_SCORE_F, RT_SCORE_H, RT_SELECT, RT_SOLVE_H, RT_COUNT };
}

struct orbx_ransac {
    CallGate gate;  // one call at a time (ORBX_ENTER)
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // the homography chain runs beside the fundamental chain (the reference uses two threads, Initializer.cpp:291-303)
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool fork_h = true;                       // homography chain on the side stream (orbx_ransac_set_fork)
    unsigned char* static_scratch = nullptr;  // this GPU's copy of the file-scope workspace
    DevBuf<float> d_pts, d_ref, d_pn, d_rn, d_T, d_Fall, d_Hall, d_Hiall, d_sf, d_sh, d_F, d_H, d_scores;
    DevBuf<uint32_t> d_sets;
    DevBuf<uint8_t> d_inlf, d_inlh;
    DevBuf<int32_t> d_best, d_bad;
    DevBuf<double> d_pd;
    // pose recovery
    DevBuf<double> d_ppts, d_pref, d_cand, d_norm, d_R, d_t, d_xyz;
    DevBuf<float> d_K, d_Fp;
    DevBuf<int32_t> d_pstart, d_good, d_count;
    StageTimer timer;
    float last_ms[RT_COUNT] = {};
};

namespace {

// Intermediate device arrays of one batch: normalised points [pairs][n][2] x2, T [pairs][2][9], hypotheses
// [pairs][iterations][9] x3, scores [pairs][iterations] x2, and one word per pair that k_solve_f raises when a set
// names a correspondence that does not exist (it must be zero when the batch starts).
struct Workspace {
    float *pn, *rn, *T, *Fall, *Hall, *Hiall, *sf, *sh;
    int32_t* bad;
    double* pd;  // [pairs][n][4]: the correspondences widened to double for the fundamental scorer
};

// The handle's own (growing) buffers as the workspace of a batch.
orbx_status handle_workspace(orbx_ransac* h, int n, int iterations, int pairs, Workspace* w) {
    const size_t pn = (size_t)pairs * n * 2, hyp = (size_t)pairs * iterations;
    ORBX_TRY(h->d_pn.ensure(pn));
    ORBX_TRY(h->d_rn.ensure(pn));
    ORBX_TRY(h->d_T.ensure((size_t)pairs * 18));
    ORBX_TRY(h->d_Fall.ensure(hyp * 9));
    ORBX_TRY(h->d_Hall.ensure(hyp * 9));
    ORBX_TRY(h->d_Hiall.ensure(hyp * 9));
    ORBX_TRY(h->d_sf.ensure(hyp));
    ORBX_TRY(h->d_sh.ensure(hyp));
    ORBX_TRY(h->d_bad.ensure(pairs));
    ORBX_TRY(h->d_pd.ensure(pn * 2));
    *w = Workspace{h->d_pn.p, h->d_rn.p, h->d_T.p, h->d_Fall.p, h->d_Hall.p, h->d_Hiall.p, h->d_sf.p, h->d_sh.p,
                   h->d_bad.p, h->d_pd.p};
    return ORBX_OK;
}

orbx_status run_batch(orbx_ransac* h, const Workspace& w, const float* d_points, const float* d_ref, int n,
                      const uint32_t* d_sets8, int iterations, int pairs, double sigma, float* d_F, float* d_H,
                      uint8_t* d_inl_f, uint8_t* d_inl_h, float* d_scores, int32_t* d_best) {
    hipStream_t s = h->stream;
    ORBX_HIP(hipMemsetAsync(w.bad, 0, sizeof(int32_t) * pairs, s));
    StageTimer& T = h->timer;
    T.clear();
    T.begin(RT_NORMALIZE, s);
    hipLaunchKernelGGL(k_normalize, dim3(pairs, 2), dim3(64), 0, s, d_points, d_ref, n, w.pn, w.rn, w.T, w.pd);
    T.end(RT_NORMALIZE, s);
    // hypotheses are numbered through across pairs, 64 per scoring workgroup; with very few iterations per pair a
    // workgroup would span many pairs - the scorers then simply run pair by pair
    const bool flat = iterations >= kScoreMinIterations;
    const int hyp_blocks64 = (int)(((long long)pairs * iterations + 63) / 64);
    const int score_blocks = flat ? hyp_blocks64 : div_up(iterations, 64);
    // F chain on the handle's stream, H chain on the side stream (the reference uses two threads,
    // Initializer.cpp:291-303): the solvers hold a whole system in registers and run two wavefronts per SIMD, so
    // neither chain fills the chip alone.  (Cutting a large batch into parts of pairs, solve and score alternating
    // on each stream, was measured slower: 2.76 ms whole, 2.78 in two parts, 3.08 in four.)
    hipStream_t sh = h->fork_h ? h->side_stream : s;
    if (h->fork_h) {
        ORBX_HIP(hipEventRecord(h->ev_fork, s));
        ORBX_HIP(hipStreamWaitEvent(sh, h->ev_fork, 0));
    }
    T.begin(RT_SOLVE, s);
    hipLaunchKernelGGL(k_solve_f, dim3(hyp_blocks64), dim3(64), 0, s, w.pn, w.rn, n, w.T, d_sets8,
                       iterations, pairs, w.Fall, w.bad);
    T.end(RT_SOLVE, s);
    T.begin(RT_SOLVE_H, sh);
    hipLaunchKernelGGL(k_solve_h, dim3(hyp_blocks64), dim3(64), 0, sh, w.pn, w.rn, n, w.T, d_sets8,
                       iterations, pairs, w.Hall, w.Hiall);
    T.end(RT_SOLVE_H, sh);
    T.begin(RT_SCORE_F, s);
    if (flat)
        hipLaunchKernelGGL(k_score_f, dim3(score_blocks), dim3(256), 0, s, w.pd, n, w.Fall, iterations, pairs, sigma,
                           w.sf);
    else
        for (int p = 0; p < pairs; p++)
            hipLaunchKernelGGL(k_score_f, dim3(score_blocks), dim3(256), 0, s, w.pd + (size_t)p * n * 4, n,
                               w.Fall + (size_t)p * iterations * 9, iterations, 1, sigma, w.sf + (size_t)p * iterations);
    T.end(RT_SCORE_F, s);
    T.begin(RT_SCORE_H, sh);
    if (flat)
        hipLaunchKernelGGL(k_score_h, dim3(score_blocks), dim3(256), 0, sh, d_points, d_ref, n, w.Hall,
                           w.Hiall, iterations, pairs, sigma, w.sh);
    else
        for (int p = 0; p < pairs; p++)
            hipLaunchKernelGGL(k_score_h, dim3(score_blocks), dim3(256), 0, sh, d_points + (size_t)p * n * 2,
                               d_ref + (size_t)p * n * 2, n, w.Hall + (size_t)p * iterations * 9,
                               w.Hiall + (size_t)p * iterations * 9, iterations, 1, sigma,
                               w.sh + (size_t)p * iterations);
    T.end(RT_SCORE_H, sh);
    if (h->fork_h) {
        ORBX_HIP(hipEventRecord(h->ev_join, sh));
        ORBX_HIP(hipStreamWaitEvent(s, h->ev_join, 0));
    }
    T.begin(RT_SELECT, s);
    hipLaunchKernelGGL(k_select, dim3(pairs, 2), dim3(256), 0, s, d_points, d_ref, n, w.Fall, w.Hall,
                       w.Hiall, w.sf, w.sh, w.bad, iterations, sigma, d_F, d_H, d_inl_f, d_inl_h,
                       d_scores, d_best);
    T.end(RT_SELECT, s);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

// ---- workspace of the entry point that allocates nothing ------------------------------------------------------
// One file-scope device array per GPU holds what handle_workspace() would otherwise allocate.  Limits of a call:
constexpr size_t kStaticMaxHyp = 32768;      // n_pairs * iterations
constexpr size_t kStaticMaxPoints = 131072;  // n_pairs * n
constexpr size_t kStaticMaxPairs = 1024;
constexpr size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr size_t kOffPn = 0;
constexpr size_t kOffRn = kOffPn + align256(kStaticMaxPoints * 2 * sizeof(float));
constexpr size_t kOffT = kOffRn + align256(kStaticMaxPoints * 2 * sizeof(float));
constexpr size_t kOffFall = kOffT + align256(kStaticMaxPairs * 18 * sizeof(float));
constexpr size_t kOffHall = kOffFall + align256(kStaticMaxHyp * 9 * sizeof(float));
constexpr size_t kOffHiall = kOffHall + align256(kStaticMaxHyp * 9 * sizeof(float));
constexpr size_t kOffSf = kOffHiall + align256(kStaticMaxHyp * 9 * sizeof(float));
constexpr size_t kOffSh = kOffSf + align256(kStaticMaxHyp * sizeof(float));
constexpr size_t kOffBad = kOffSh + align256(kStaticMaxHyp * sizeof(float));
constexpr size_t kOffPd = kOffBad + align256(kStaticMaxPairs * sizeof(int32_t));
constexpr size_t kScratchBytes = kOffPd + align256(kStaticMaxPoints * 4 * sizeof(double));  // about 10.1 MB

// Every GPU the library runs on holds its own copy of this array until the process ends.  In host code the name
// is only a handle for hipGetSymbolAddress: the device address it yields (one per GPU) is what kernels and
// hipMemsetAsync receive.
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = {};
// One event per GPU marks the end of the last call that used that GPU's array: the next call, on whichever handle
// and stream, makes its stream wait for it first, so calls that use the array run one after the other on the device
// even when their streams would let them overlap.  Created with the address, at set-up: a call creates nothing.
hipEvent_t g_scratch_done[kMaxDevices] = {};

// Address of `scratch` on `device` (which must be the current device); looked up once per device.
orbx_status resolve_scratch(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return ORBX_BAD_ARG;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        unsigned char* ptr = nullptr;
        ORBX_HIP(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return ORBX_DEVICE;
        hipEvent_t ev = nullptr;
        ORBX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g_scratch_done[device] = ev;
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return ORBX_OK;
}

orbx_status check_sets(const uint32_t* sets8, int iterations, int n) {
    for (size_t i = 0; i < (size_t)iterations * 8; i++)
        if (sets8[i] >= (uint32_t)n) return ORBX_BAD_ARG;
    return ORBX_OK;
}

orbx_status upload_pair(orbx_ransac* h, const float* points, const float* ref_points, int n, const uint32_t* sets8,
                        int iterations) {
    hipStream_t s = h->stream;
    ORBX_TRY(h->d_pts.ensure((size_t)n * 2));
    ORBX_TRY(h->d_ref.ensure((size_t)n * 2));
    ORBX_TRY(h->d_sets.ensure((size_t)iterations * 8));
    ORBX_HIP(hipMemcpyAsync(h->d_pts.p, points, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_ref.p, ref_points, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_sets.p, sets8, sizeof(uint32_t) * 8 * iterations, hipMemcpyHostToDevice, s));
    return ORBX_OK;
}

}  // namespace

extern "C" {

orbx_status orbx_ransac_create(int device, orbx_ransac** out) {
    if (!out) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(device));
    orbx_ransac* h = new orbx_ransac();
    h->device = device;
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return ORBX_DEVICE;
    }
    h->stream = h->own_stream;
    if (hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
        orbx_ransac_destroy(h);
        return ORBX_DEVICE;
    }
    orbx_status st = h->timer.init();
    if (st == ORBX_OK) st = resolve_scratch(device, &h->static_scratch);
    if (st != ORBX_OK) {
        orbx_ransac_destroy(h);
        return st;
    }
    *out = h;
    return ORBX_OK;
}

void orbx_ransac_destroy(orbx_ransac* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    h->d_pts.release(); h->d_ref.release(); h->d_pn.release(); h->d_rn.release(); h->d_T.release();
    h->d_Fall.release(); h->d_Hall.release(); h->d_Hiall.release(); h->d_sf.release(); h->d_sh.release();
    h->d_F.release(); h->d_H.release(); h->d_scores.release(); h->d_sets.release(); h->d_inlf.release();
    h->d_inlh.release(); h->d_best.release(); h->d_bad.release(); h->d_pd.release();
    h->d_ppts.release(); h->d_pref.release(); h->d_cand.release(); h->d_norm.release(); h->d_R.release();
    h->d_t.release(); h->d_xyz.release(); h->d_K.release(); h->d_Fp.release(); h->d_pstart.release();
    h->d_good.release(); h->d_count.release();
    h->timer.destroy();
    if (h->side_stream) { (void)hipStreamSynchronize(h->side_stream); (void)hipStreamDestroy(h->side_stream); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

orbx_status orbx_ransac_set_stream(orbx_ransac* h, void* hip_stream) {
    ORBX_ENTER(h);
    if (!h) return ORBX_BAD_ARG;
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return ORBX_OK;
}

orbx_status orbx_ransac_set_fork(orbx_ransac* h, int enabled) {
    ORBX_ENTER(h);
    h->fork_h = enabled != 0;
    return ORBX_OK;
}

int orbx_ransac_last_timings(const orbx_ransac* hc, float* ms, int cap) {
    orbx_ransac* h = const_cast<orbx_ransac*>(hc);
    if (!h || !ms) return 0;
    int n = std::min(cap, (int)RT_COUNT);
    for (int i = 0; i < n; i++) ms[i] = h->timer.ms(i);
    return n;
}

orbx_status orbx_ransac_fh_batch(orbx_ransac* h, const float* d_points, const float* d_ref_points, int n,
                                 const uint32_t* d_sets8, int iterations, int n_pairs, double sigma, float* d_F,
                                 float* d_H, uint8_t* d_inliers_f, uint8_t* d_inliers_h, float* d_scores,
                                 int32_t* d_best_iteration) {
    ORBX_ENTER(h);
    if (!h || !d_points || !d_ref_points || !d_sets8 || !d_F || !d_H || !d_inliers_f || !d_inliers_h || !d_scores ||
        !d_best_iteration)
        return ORBX_BAD_ARG;
    if (n < 8 || iterations <= 0 || n_pairs <= 0 || !(sigma > 0)) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    Workspace w;
    ORBX_TRY(handle_workspace(h, n, iterations, n_pairs, &w));
    return run_batch(h, w, d_points, d_ref_points, n, d_sets8, iterations, n_pairs, sigma, d_F, d_H, d_inliers_f,
                     d_inliers_h, d_scores, d_best_iteration);
}

orbx_status orbx_ransac_fh_batch_static(orbx_ransac* h, const float* d_points, const float* d_ref_points, int n,
                                        const uint32_t* d_sets8, int iterations, int n_pairs, double sigma,
                                        float* d_F, float* d_H, uint8_t* d_inliers_f, uint8_t* d_inliers_h,
                                        float* d_scores, int32_t* d_best_iteration) {
    ORBX_ENTER(h);
    if (!h || !d_points || !d_ref_points || !d_sets8 || !d_F || !d_H || !d_inliers_f || !d_inliers_h || !d_scores ||
        !d_best_iteration)
        return ORBX_BAD_ARG;
    if (n < 8 || iterations <= 0 || n_pairs <= 0 || !(sigma > 0)) return ORBX_BAD_ARG;
    if ((size_t)n_pairs > kStaticMaxPairs || (size_t)n_pairs * iterations > kStaticMaxHyp ||
        (size_t)n_pairs * n > kStaticMaxPoints) {
        char msg[256];
        std::snprintf(msg, sizeof(msg),
                      "orbx_ransac_fh_batch_static: %d pairs x %d iterations x %d correspondences exceed the fixed "
                      "workspace (at most %zu pairs, %zu hypotheses and %zu correspondences per call)",
                      n_pairs, iterations, n, kStaticMaxPairs, kStaticMaxHyp, kStaticMaxPoints);
        set_last_error_text(msg);
        return ORBX_CAPACITY;
    }
    ORBX_HIP(hipSetDevice(h->device));
    unsigned char* base = h->static_scratch;
    const Workspace w{(float*)(base + kOffPn),   (float*)(base + kOffRn),    (float*)(base + kOffT),
                      (float*)(base + kOffFall), (float*)(base + kOffHall),  (float*)(base + kOffHiall),
                      (float*)(base + kOffSf),   (float*)(base + kOffSh),    (int32_t*)(base + kOffBad),
                      (double*)(base + kOffPd)};
    // queue behind the previous user of this GPU's array, and mark this call's end for the next one; the lock keeps
    // wait -> launches -> record of two host threads from interleaving
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    hipEvent_t done = g_scratch_done[h->device];
    ORBX_HIP(hipStreamWaitEvent(h->stream, done, 0));
    const orbx_status st = run_batch(h, w, d_points, d_ref_points, n, d_sets8, iterations, n_pairs, sigma, d_F, d_H,
                                     d_inliers_f, d_inliers_h, d_scores, d_best_iteration);
    ORBX_HIP(hipEventRecord(done, h->stream));
    return st;
}

orbx_status orbx_ransac_fh(orbx_ransac* h, const float* points, const float* ref_points, int n, const uint32_t* sets8,
                           int iterations, double sigma, float* F, float* H, uint8_t* inliers_f, uint8_t* inliers_h,
                           double* score_f, double* score_h, int32_t* best_iteration) {
    ORBX_ENTER(h);
    if (!h || !points || !ref_points || !sets8 || !F || !H || !inliers_f || !inliers_h || !score_f || !score_h ||
        !best_iteration)
        return ORBX_BAD_ARG;
    if (n < 8 || iterations <= 0 || !(sigma > 0)) return ORBX_BAD_ARG;
    ORBX_TRY(check_sets(sets8, iterations, n));
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    ORBX_TRY(upload_pair(h, points, ref_points, n, sets8, iterations));
    ORBX_TRY(h->d_F.ensure(9));
    ORBX_TRY(h->d_H.ensure(9));
    ORBX_TRY(h->d_inlf.ensure(n));
    ORBX_TRY(h->d_inlh.ensure(n));
    ORBX_TRY(h->d_scores.ensure(2));
    ORBX_TRY(h->d_best.ensure(2));
    Workspace w;
    ORBX_TRY(handle_workspace(h, n, iterations, 1, &w));
    ORBX_TRY(run_batch(h, w, h->d_pts.p, h->d_ref.p, n, h->d_sets.p, iterations, 1, sigma, h->d_F.p, h->d_H.p,
                       h->d_inlf.p, h->d_inlh.p, h->d_scores.p, h->d_best.p));
    float hF[9], hH[9], sc[2];
    int32_t best[2];
    ORBX_HIP(hipMemcpyAsync(hF, h->d_F.p, sizeof(hF), hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(hH, h->d_H.p, sizeof(hH), hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(sc, h->d_scores.p, sizeof(sc), hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(best, h->d_best.p, sizeof(best), hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(inliers_f, h->d_inlf.p, n, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(inliers_h, h->d_inlh.p, n, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    if (best[0] >= 0) std::memcpy(F, hF, sizeof(hF));
    if (best[1] >= 0) std::memcpy(H, hH, sizeof(hH));
    *score_f = sc[0];
    *score_h = sc[1];
    best_iteration[0] = best[0];
    best_iteration[1] = best[1];
    return ORBX_OK;
}

orbx_status orbx_ransac_fh_all(orbx_ransac* h, const float* points, const float* ref_points, int n,
                               const uint32_t* sets8, int iterations, double sigma, float* F_all, float* H_all,
                               float* Hinv_all, float* score_f, float* score_h) {
    ORBX_ENTER(h);
    if (!h || !points || !ref_points || !sets8) return ORBX_BAD_ARG;
    if (n < 8 || iterations <= 0 || !(sigma > 0)) return ORBX_BAD_ARG;
    ORBX_TRY(check_sets(sets8, iterations, n));
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    ORBX_TRY(upload_pair(h, points, ref_points, n, sets8, iterations));
    ORBX_TRY(h->d_F.ensure(9));
    ORBX_TRY(h->d_H.ensure(9));
    ORBX_TRY(h->d_inlf.ensure(n));
    ORBX_TRY(h->d_inlh.ensure(n));
    ORBX_TRY(h->d_scores.ensure(2));
    ORBX_TRY(h->d_best.ensure(2));
    Workspace w;
    ORBX_TRY(handle_workspace(h, n, iterations, 1, &w));
    ORBX_TRY(run_batch(h, w, h->d_pts.p, h->d_ref.p, n, h->d_sets.p, iterations, 1, sigma, h->d_F.p, h->d_H.p,
                       h->d_inlf.p, h->d_inlh.p, h->d_scores.p, h->d_best.p));
    const size_t m = (size_t)iterations * 9 * sizeof(float);
    if (F_all) ORBX_HIP(hipMemcpyAsync(F_all, h->d_Fall.p, m, hipMemcpyDeviceToHost, s));
    if (H_all) ORBX_HIP(hipMemcpyAsync(H_all, h->d_Hall.p, m, hipMemcpyDeviceToHost, s));
    if (Hinv_all) ORBX_HIP(hipMemcpyAsync(Hinv_all, h->d_Hiall.p, m, hipMemcpyDeviceToHost, s));
    if (score_f) ORBX_HIP(hipMemcpyAsync(score_f, h->d_sf.p, sizeof(float) * iterations, hipMemcpyDeviceToHost, s));
    if (score_h) ORBX_HIP(hipMemcpyAsync(score_h, h->d_sh.p, sizeof(float) * iterations, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    return ORBX_OK;
}

// ---- the public solvers of geometry/utils.h for any number of correspondences ---------------------------------

orbx_status orbx_normalize_points(orbx_ransac* h, const float* points, int n, float* normalized, float* T9) {
    ORBX_ENTER(h);
    if (!h || !points || !normalized || !T9 || n <= 0) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    ORBX_TRY(h->d_pts.ensure((size_t)n * 2));
    ORBX_TRY(h->d_pn.ensure((size_t)n * 2));
    ORBX_TRY(h->d_T.ensure(18));
    ORBX_HIP(hipMemcpyAsync(h->d_pts.p, points, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
    // one point set: side 0 of a single pair (the reference-side arguments are never touched)
    hipLaunchKernelGGL(k_normalize, dim3(1, 1), dim3(64), 0, s, h->d_pts.p, h->d_pts.p, n, h->d_pn.p, h->d_pn.p,
                       h->d_T.p, (double*)nullptr);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(normalized, h->d_pn.p, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(T9, h->d_T.p, sizeof(float) * 9, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    return ORBX_OK;
}

}  // extern "C"

namespace {
orbx_status solve_npt(orbx_ransac* h, const float* points, const float* ref_points, int n, int n_systems, int model,
                      float* out9) {
    if (!h || !points || !ref_points || !out9 || n <= 0 || n_systems <= 0) return ORBX_BAD_ARG;
    if ((long long)n * n_systems > (1ll << 28)) return ORBX_CAPACITY;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const size_t total = (size_t)n * n_systems;
    const int rows = model == 0 ? n : 2 * n;
    ORBX_TRY(h->d_pts.ensure(total * 2));
    ORBX_TRY(h->d_ref.ensure(total * 2));
    ORBX_TRY(h->d_Fall.ensure((size_t)9 * std::max(rows, 9) * n_systems));  // the systems' rows, worked on in place
    ORBX_TRY(h->d_F.ensure((size_t)9 * n_systems));
    ORBX_HIP(hipMemcpyAsync(h->d_pts.p, points, sizeof(float) * 2 * total, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_ref.p, ref_points, sizeof(float) * 2 * total, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_solve_npt, dim3(n_systems), dim3(64), 0, s, h->d_pts.p, h->d_ref.p, n, model, h->d_Fall.p,
                       h->d_F.p);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(out9, h->d_F.p, sizeof(float) * 9 * n_systems, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    return ORBX_OK;
}
}  // namespace

extern "C" {

orbx_status orbx_fundamental_npt(orbx_ransac* h, const float* points, const float* ref_points, int n, int n_systems,
                                 float* F9) {
    ORBX_ENTER(h);
    return solve_npt(h, points, ref_points, n, n_systems, 0, F9);
}

orbx_status orbx_homography_dlt(orbx_ransac* h, const float* points, const float* ref_points, int n, int n_systems,
                                float* H9) {
    ORBX_ENTER(h);
    return solve_npt(h, points, ref_points, n, n_systems, 1, H9);
}

}  // extern "C"

namespace {

// Validates pair_start and uploads the two double point sets; returns the total and the longest pair.
orbx_status upload_pose_points(orbx_ransac* h, int n_pairs, const double* points, const double* ref_points,
                               const int32_t* pair_start, int* total_out, int* longest_out) {
    if (pair_start[0] != 0) return ORBX_BAD_ARG;
    int longest = 0;
    for (int p = 0; p < n_pairs; p++) {
        if (pair_start[p + 1] < pair_start[p]) return ORBX_BAD_ARG;
        longest = std::max(longest, pair_start[p + 1] - pair_start[p]);
    }
    const int total = pair_start[n_pairs];
    if (total > 0 && (!points || !ref_points)) return ORBX_BAD_ARG;
    hipStream_t s = h->stream;
    ORBX_TRY(h->d_ppts.ensure(std::max((size_t)total * 2, (size_t)2)));
    ORBX_TRY(h->d_pref.ensure(std::max((size_t)total * 2, (size_t)2)));
    ORBX_TRY(h->d_pstart.ensure((size_t)n_pairs + 1));
    if (total > 0) {
        ORBX_HIP(hipMemcpyAsync(h->d_ppts.p, points, sizeof(double) * 2 * total, hipMemcpyHostToDevice, s));
        ORBX_HIP(hipMemcpyAsync(h->d_pref.p, ref_points, sizeof(double) * 2 * total, hipMemcpyHostToDevice, s));
    }
    ORBX_HIP(hipMemcpyAsync(h->d_pstart.p, pair_start, sizeof(int32_t) * (n_pairs + 1), hipMemcpyHostToDevice, s));
    *total_out = total;
    *longest_out = longest;
    return ORBX_OK;
}

}  // namespace

extern "C" {

orbx_status orbx_recover_pose_batch(orbx_ransac* h, int n_pairs, const float* fundamental, const float* intrinsics,
                                    const double* points, const double* ref_points, const int32_t* pair_start,
                                    double* rotation, double* translation, int32_t* n_good) {
    ORBX_ENTER(h);
    if (!h || n_pairs < 0) return ORBX_BAD_ARG;
    if (n_pairs == 0) return ORBX_OK;
    if (!fundamental || !intrinsics || !pair_start || !rotation || !translation || !n_good) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int total = 0, longest = 0;
    ORBX_TRY(upload_pose_points(h, n_pairs, points, ref_points, pair_start, &total, &longest));
    ORBX_TRY(h->d_K.ensure(9));
    ORBX_TRY(h->d_Fp.ensure((size_t)n_pairs * 9));
    ORBX_TRY(h->d_cand.ensure((size_t)n_pairs * 21));
    ORBX_TRY(h->d_norm.ensure(4));
    ORBX_TRY(h->d_good.ensure((size_t)n_pairs * 4));
    ORBX_TRY(h->d_R.ensure((size_t)n_pairs * 9));
    ORBX_TRY(h->d_t.ensure((size_t)n_pairs * 3));
    ORBX_TRY(h->d_count.ensure(n_pairs));
    ORBX_HIP(hipMemcpyAsync(h->d_K.p, intrinsics, sizeof(float) * 9, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_Fp.p, fundamental, sizeof(float) * 9 * n_pairs, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemsetAsync(h->d_good.p, 0, sizeof(int32_t) * 4 * n_pairs, s));
    hipLaunchKernelGGL(k_pose_candidates, dim3(div_up(n_pairs, 64)), dim3(64), 0, s, h->d_Fp.p, h->d_K.p, n_pairs,
                       h->d_cand.p, h->d_norm.p);
    if (longest > 0)
        hipLaunchKernelGGL(k_pose_cheirality, dim3(div_up(longest, 256), n_pairs), dim3(256), 0, s, h->d_ppts.p,
                           h->d_pref.p, h->d_pstart.p, h->d_cand.p, h->d_norm.p, h->d_good.p);
    hipLaunchKernelGGL(k_pose_select, dim3(div_up(n_pairs, 64)), dim3(64), 0, s, h->d_cand.p, h->d_good.p, n_pairs,
                       h->d_R.p, h->d_t.p, h->d_count.p);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(rotation, h->d_R.p, sizeof(double) * 9 * n_pairs, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(translation, h->d_t.p, sizeof(double) * 3 * n_pairs, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(n_good, h->d_count.p, sizeof(int32_t) * n_pairs, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    return ORBX_OK;
}

orbx_status orbx_triangulate_batch(orbx_ransac* h, int n_pairs, const float* intrinsics, const double* rotation,
                                   const double* translation, const double* points, const double* ref_points,
                                   const int32_t* pair_start, double* points_3d) {
    ORBX_ENTER(h);
    if (!h || n_pairs < 0) return ORBX_BAD_ARG;
    if (n_pairs == 0) return ORBX_OK;
    if (!intrinsics || !rotation || !translation || !pair_start) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int total = 0, longest = 0;
    ORBX_TRY(upload_pose_points(h, n_pairs, points, ref_points, pair_start, &total, &longest));
    if (total == 0) return ORBX_OK;
    if (!points_3d) return ORBX_BAD_ARG;
    ORBX_TRY(h->d_K.ensure(9));
    ORBX_TRY(h->d_R.ensure((size_t)n_pairs * 9));
    ORBX_TRY(h->d_t.ensure((size_t)n_pairs * 3));
    ORBX_TRY(h->d_xyz.ensure((size_t)total * 3));
    ORBX_HIP(hipMemcpyAsync(h->d_K.p, intrinsics, sizeof(float) * 9, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_R.p, rotation, sizeof(double) * 9 * n_pairs, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_t.p, translation, sizeof(double) * 3 * n_pairs, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_triangulate_initial, dim3(div_up(longest, 256), n_pairs), dim3(256), 0, s, h->d_K.p,
                       h->d_R.p, h->d_t.p, h->d_ppts.p, h->d_pref.p, h->d_pstart.p, h->d_xyz.p);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(points_3d, h->d_xyz.p, sizeof(double) * 3 * total, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    return ORBX_OK;
}

orbx_status orbx_recover_pose_homography_batch(orbx_ransac* h, int n_pairs, const float* homography,
                                               const float* intrinsics, const double* points, const double* ref_points,
                                               const int32_t* pair_start, double* rotation, double* translation,
                                               int32_t* solution, double* all_solutions, int32_t* n_solutions) {
    ORBX_ENTER(h);
    if (!h || n_pairs < 0) return ORBX_BAD_ARG;
    if (n_pairs == 0) return ORBX_OK;
    if (!homography || !intrinsics || !pair_start || !rotation || !translation || !solution) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int total = 0, longest = 0;
    ORBX_TRY(upload_pose_points(h, n_pairs, points, ref_points, pair_start, &total, &longest));
    ORBX_TRY(h->d_K.ensure(9));
    ORBX_TRY(h->d_Fp.ensure((size_t)n_pairs * 9));
    ORBX_TRY(h->d_cand.ensure((size_t)n_pairs * 60));
    ORBX_TRY(h->d_good.ensure((size_t)n_pairs * 2));
    ORBX_HIP(hipMemcpyAsync(h->d_K.p, intrinsics, sizeof(float) * 9, hipMemcpyHostToDevice, s));
    ORBX_HIP(hipMemcpyAsync(h->d_Fp.p, homography, sizeof(float) * 9 * n_pairs, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pose_homography, dim3(n_pairs), dim3(64), 0, s, h->d_Fp.p, h->d_K.p, h->d_ppts.p,
                       h->d_pref.p, h->d_pstart.p, h->d_cand.p, h->d_good.p);
    ORBX_HIP(hipGetLastError());
    std::vector<double> sol((size_t)n_pairs * 60);
    std::vector<int32_t> info((size_t)n_pairs * 2);
    ORBX_HIP(hipMemcpyAsync(sol.data(), h->d_cand.p, sizeof(double) * sol.size(), hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(info.data(), h->d_good.p, sizeof(int32_t) * info.size(), hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    for (int p = 0; p < n_pairs; p++) {
        // `if (!sols.empty()) idx = sols.at<int>(0, 0)`: the lowest visible solution
        const int mask = info[2 * p + 1];
        const int idx = mask ? __builtin_ctz((unsigned)mask) : -1;
        solution[p] = idx;
        for (int i = 0; i < 9; i++) rotation[(size_t)p * 9 + i] = idx >= 0 ? sol[(size_t)p * 60 + idx * 15 + i] : 0.0;
        for (int i = 0; i < 3; i++)
            translation[(size_t)p * 3 + i] = idx >= 0 ? sol[(size_t)p * 60 + idx * 15 + 9 + i] : 0.0;
        if (n_solutions) n_solutions[p] = info[2 * p];
    }
    if (all_solutions) std::memcpy(all_solutions, sol.data(), sizeof(double) * sol.size());
    return ORBX_OK;
}

}  // extern "C"
