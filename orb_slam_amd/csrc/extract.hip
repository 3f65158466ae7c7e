// ORB extractor: host orchestration of the gfx950 kernels behind orbx_detect / orbx_compute /
// orbx_extract_batch.  Replaces cv::ORB::detect / compute as called by ORBExtractor
// (reference include/orb_slam/geometry/orb_extractor.h:57-80,100-119).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <thread>

#include "common.h"
#include "extract_kernels.h"
#include "distribute_kernels.h"
#include "geometry_host.h"
#include "host_logic.h"

using namespace orbx;

namespace {

const int kBitPattern31[256 * 4] = {
#include "bit_pattern_31.inc"
};

enum Stage { ST_GRAY = 0, ST_RESIZE, ST_FAST, ST_NMS, ST_EMIT, ST_HARRIS, ST_BLUR, ST_BRIEF, ST_SELECT, ST_ANGLE, ST_COUNT };

constexpr size_t kStreamMaxLds = 64 * 1024;  // rings of the one-pass pyramid kernel: at least two workgroups per CU
static_assert(kResizeRows == geomdetail::kResizeRowsHost && kResizeSrcRows == geomdetail::kResizeSrcRowsHost,
              "geometry_host.h plans for the strip kernels' block shape");

inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

}  // namespace

struct orbx_extractor {
    CallGate gate;  // one call at a time (ORBX_ENTER)
    orbx_extractor_params prm;
    int device = 0;
    int max_w = 0, max_h = 0, max_batch = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // the blur needs only the pyramid: it runs on a side stream next to FAST / emission / selection
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_pyramid = nullptr, ev_blur = nullptr;
    bool blur_forked = false;
    bool fork_blur = true;   // orbx_extract_batch runs the blur on the side stream beside emission / Harris / selection

    // geometry of the current image size
    int cur_w = -1, cur_h = -1, cur_levels = -1;
    Geom geom;
    std::vector<Tile> tiles;
    std::vector<Strip> strips;
    std::vector<Coef> coefs;
    std::vector<CoefX> coefx;
    std::vector<CoefY> coefy;
    ResizePlan rplan[kMaxLevels];
    // the one-pass pyramid kernel (pyramid_stream.h): its plan for the current geometry and the tables on the device
    StreamPlan splan;
    DevBuf<StreamCol> d_scols;
    DevBuf<StreamRow> d_srows;
    DevBuf<short> d_ssched;
    DevBuf<StreamCol> d_tcols;
    DevBuf<StreamRow> d_trows;
    DevBuf<short> d_tsched;
    // opt-in distribution mode (orbx_extractor_set_distribution): 0 = cv::ORB's own selection
    int distribute_mode = 0, distribute_low = 7, distribute_cell = 30;
    DevBuf<uint16_t> d_labels;       // box label per candidate (k_distribute_level's work array)
    DevBuf<int> d_picked_idx, d_picked_n;
    int picked_stride = 0;
    int pyramid_kernel = 0;   // 0 / 1: level by level; 2: the one-pass kernel when the plan suits the geometry
    int stream_rows = 8;      // level-0 rows per step of the one-pass kernel
    StreamPlan tplan;         // the tail form: levels above tail_base in one launch
    int tail_base = 0;        // 0: off
    int tail_rows = 16;       // rows of level tail_base per step
    DevBuf<CoefX> d_coefx;
    DevBuf<CoefY> d_coefy;
    DevBuf<Tile> d_tiles;
    DevBuf<Strip> d_strips;
    DevBuf<BriefPattern> d_pattern;
    UMax umax;
    BlurTaps blur_taps;
    float harris_scale_sq_sq = 0.f;

    size_t atlas_bytes = 0;
    int cand_cap = 0;  // candidate corners per frame
    int kp_cap = 0;    // final keypoints per frame

    DevBuf<uint8_t> d_in, d_atlas, d_score, d_blur, d_desc;
    DevBuf<MaskWord> d_mask;
    DevBuf<Steer> d_steer;
    DevBuf<Cand> d_cand, d_packed;
    DevBuf<int> d_thresholds, d_totals, d_offsets, d_kp_counts;
    // accumulators zeroed once per batch, back to back in one allocation: per-level counts [frames][kMaxLevels],
    // per-run corner counts [frames][kEmitMaxChunks], per-level score histograms [frames][kMaxLevels][256]
    DevBuf<int> d_acc;
    int* d_level_counts_p = nullptr;
    int* d_chunk_counts_p = nullptr;
    int* d_hist_p = nullptr;
    size_t acc_ints = 0;
    int emit_chunks = 1, emit_per = 256;
    DevBuf<orbx_keypoint> d_kp;
    PinBuf<Cand> h_packed;
    PinBuf<int> h_ints;
    PinBuf<orbx_keypoint> h_kp;
    PinBuf<uint8_t> h_desc;
    // optional frame model of MonoFrame::extractFeatures for the batch path
    bool has_camera = false, has_grid = false;
    orbx_camera camera;
    orbx_grid grid;
    DevBuf<int32_t> d_cell_start, d_cell_items;
    PinBuf<int32_t> h_cell_start, h_cell_items;
    DevBuf<int> d_status;
    DevBuf<int> d_kp_src;               // batch path: where each kept keypoint was detected (k_select_frame -> k_angle_kept)
    bool harris_split = false;          // the last detect stage computed responses only (k_harris): no angles in d_cand
    DevBuf<unsigned char> d_select_ws;  // workspace of the selection kernel's global form
    FrameModel sel_fm;                  // what the last device selection ran with (more_global_passes repeats it)
    size_t sel_ws_stride = 0;
    int sel_ws_slices = 0;
    bool device_select = true;     // batch path: select / undistort / grid on the device (k_select_frame)
    int last_frames = 0;
    bool pyramid_valid = false;  // atlas holds the pyramid of the frames of the last detect stage

    // detect() followed by compute() on the same image (MonoFrame::extractFeatures, mono_frame.cpp:75,102): the
    // second call finds the uploaded image and its pyramid still on the device
    struct ImageKey {
        const void* ptr = nullptr;
        int w = 0, h = 0, stride = 0, ch = 0;
        uint64_t sample = 0;
        bool valid = false;
    } cached_image;
    bool image_cache_enabled = true;

    // a batch whose caller asked for no host copies returns before the GPU has finished; its status words are
    // looked at by orbx_extractor_finish (or dropped by the next call on the handle)
    struct Pending {
        bool active = false;
        const uint8_t* d_images = nullptr;
        int n_frames = 0;
        size_t frame_stride = 0;
        int width = 0, height = 0, stride = 0, channels = 0, capacity = 0;
    } pending;
    hipEvent_t ev_done = nullptr;
    std::vector<int32_t> last_counts;  // keypoints per frame of the most recent batch, once it has finished
    bool batch_dropped = false;        // a deferred batch was waited for and dropped by a later call, never finished

    StageTimer timer;
    float last_ms[ST_COUNT] = {};
};

namespace {

// cv::ORB detectAndCompute(): border, level sizes and atlas placement; computeKeyPoints(): per-level quotas
// (compute_geometry, geometry_host.h); then the tables go to the device.
orbx_status build_geometry(orbx_extractor* h, int w, int hgt, int nlevels) {
    if (h->cur_w == w && h->cur_h == hgt && h->cur_levels == nlevels) return ORBX_OK;
    {
        HostGeometry hg;
        ORBX_TRY(compute_geometry(h->prm, w, hgt, nlevels, hg));
        h->geom = hg.g;
        h->tiles.swap(hg.tiles);
        h->strips.swap(hg.strips);
        h->coefs.swap(hg.coefs);
        h->coefx.swap(hg.coefx);
        h->coefy.swap(hg.coefy);
        for (int l = 0; l < kMaxLevels; l++) h->rplan[l] = hg.rplan[l];
        h->atlas_bytes = hg.atlas_bytes;
        // the one-pass pyramid: planned and checked on the host here; a geometry it does not suit leaves splan.ok false
        // (swap the vectors back for the planner: it reads the coefficient tables)
        hg.coefx.swap(h->coefx);
        hg.coefy.swap(h->coefy);
        build_stream_plan(hg, w, hgt, h->stream_rows, kStreamMaxLds, h->splan);
        h->tplan = StreamPlan();
        if (h->tail_base > 0 && h->tail_base < hg.g.nlevels - 1)
            build_stream_plan(hg, hg.g.L[h->tail_base].w, hg.g.L[h->tail_base].h, h->tail_rows, kStreamMaxLds, h->tplan,
                              h->tail_base);
        hg.coefx.swap(h->coefx);
        hg.coefy.swap(h->coefy);
    }
    if (h->splan.ok) {
        ORBX_TRY(h->d_scols.ensure(std::max<size_t>(h->splan.cols.size(), 1)));
        ORBX_TRY(h->d_srows.ensure(h->splan.rows.size()));
        ORBX_TRY(h->d_ssched.ensure(h->splan.sched.size()));
        if (!h->splan.cols.empty())
            ORBX_HIP(hipMemcpyAsync(h->d_scols.p, h->splan.cols.data(), h->splan.cols.size() * sizeof(StreamCol),
                                    hipMemcpyHostToDevice, h->stream));
        ORBX_HIP(hipMemcpyAsync(h->d_srows.p, h->splan.rows.data(), h->splan.rows.size() * sizeof(StreamRow),
                                hipMemcpyHostToDevice, h->stream));
        ORBX_HIP(hipMemcpyAsync(h->d_ssched.p, h->splan.sched.data(), h->splan.sched.size() * sizeof(short),
                                hipMemcpyHostToDevice, h->stream));
        if (ensure_dynamic_lds((const void*)k_pyramid_stream<false>, h->splan.lds_bytes) != ORBX_OK) h->splan.ok = false;
    }
    if (h->tplan.ok) {
        ORBX_TRY(h->d_tcols.ensure(std::max<size_t>(h->tplan.cols.size(), 1)));
        ORBX_TRY(h->d_trows.ensure(h->tplan.rows.size()));
        ORBX_TRY(h->d_tsched.ensure(h->tplan.sched.size()));
        if (!h->tplan.cols.empty())
            ORBX_HIP(hipMemcpyAsync(h->d_tcols.p, h->tplan.cols.data(), h->tplan.cols.size() * sizeof(StreamCol),
                                    hipMemcpyHostToDevice, h->stream));
        ORBX_HIP(hipMemcpyAsync(h->d_trows.p, h->tplan.rows.data(), h->tplan.rows.size() * sizeof(StreamRow),
                                hipMemcpyHostToDevice, h->stream));
        ORBX_HIP(hipMemcpyAsync(h->d_tsched.p, h->tplan.sched.data(), h->tplan.sched.size() * sizeof(short),
                                hipMemcpyHostToDevice, h->stream));
        if (ensure_dynamic_lds((const void*)k_pyramid_stream<true>, h->tplan.lds_bytes) != ORBX_OK) h->tplan.ok = false;
    }
    ORBX_TRY(h->d_tiles.ensure(h->tiles.size()));
    ORBX_TRY(h->d_strips.ensure(h->strips.size()));
    ORBX_HIP(hipMemcpyAsync(h->d_strips.p, h->strips.data(), h->strips.size() * sizeof(Strip), hipMemcpyHostToDevice, h->stream));
    ORBX_TRY(h->d_coefx.ensure(std::max<size_t>(h->coefx.size(), 4)));
    ORBX_TRY(h->d_coefy.ensure(std::max<size_t>(h->coefy.size(), 1)));
    if (!h->coefx.empty()) {
        ORBX_HIP(hipMemcpyAsync(h->d_coefx.p, h->coefx.data(), h->coefx.size() * sizeof(CoefX), hipMemcpyHostToDevice, h->stream));
        ORBX_HIP(hipMemcpyAsync(h->d_coefy.p, h->coefy.data(), h->coefy.size() * sizeof(CoefY), hipMemcpyHostToDevice, h->stream));
    }
    ORBX_HIP(hipMemcpyAsync(h->d_tiles.p, h->tiles.data(), h->tiles.size() * sizeof(Tile), hipMemcpyHostToDevice, h->stream));
    ORBX_HIP(hipStreamSynchronize(h->stream));
    h->cur_w = w;
    h->cur_h = hgt;
    h->cur_levels = nlevels;
    h->pyramid_valid = false;
    return ORBX_OK;
}

// The pyramid of `frames` device-resident images: gray conversion + one resize launch per level, or - on request - the
// one-pass kernel (one workgroup per frame, one launch).  Measured on 792 VGA frames, alone on the chip: 0.77 ms level
// by level (8 launches) against 0.98 ms in one pass; with two batches in flight the step time is the same.  One
// workgroup per frame gives every SIMD three wavefronts that each walk a dependent chain of steps, and the lanes of a
// wavefront own columns of levels that emit different numbers of rows per step: it reads and writes less than half
// the bytes and still loses to kernels that spread one level over the whole chip.  The automatic choice is therefore
// level by level.
bool use_stream_pyramid(const orbx_extractor* h, int frames) {
    (void)frames;
    return h->splan.ok && h->pyramid_kernel == 2;
}

void launch_pyramid(orbx_extractor* h, const uint8_t* d_images, int frames, size_t frame_stride, int w, int hgt,
                    int stride, int ch);

orbx_status ensure_buffers(orbx_extractor* h, int frames) {
    const Geom& g = h->geom;
    size_t A = h->atlas_bytes;
    ORBX_TRY(h->d_atlas.ensure(A * frames + 256));  // slack: the segment-wise kernels read up to 96 bytes past a row end
    ORBX_TRY(h->d_score.ensure(A * frames));
    ORBX_TRY(h->d_blur.ensure(A * frames + 64));
    ORBX_TRY(h->d_mask.ensure((size_t)g.total_words * frames));
    // candidate capacity: a strict 3x3 maximum can occur at most once per 2x2 pixels; a quarter of that bound is
    // ample for natural images, and a frame that exceeds it makes the list grow (regrow_candidates).
    size_t px = 0;
    for (int l = 0; l < g.nlevels; l++) px += (size_t)g.L[l].w * g.L[l].h;
    int cap = (int)std::min<size_t>(std::max<size_t>(px / 16, 4096), 1u << 20);
    if (cap > h->cand_cap) h->cand_cap = cap;
    ORBX_TRY(h->d_cand.ensure((size_t)h->cand_cap * frames));
    ORBX_TRY(h->d_packed.ensure((size_t)h->cand_cap * frames));
    h->acc_ints = (size_t)frames * (kMaxLevels + kEmitMaxChunks + kMaxLevels * 256);
    ORBX_TRY(h->d_acc.ensure(h->acc_ints));
    h->d_level_counts_p = h->d_acc.p;
    h->d_chunk_counts_p = h->d_level_counts_p + (size_t)frames * kMaxLevels;
    h->d_hist_p = h->d_chunk_counts_p + (size_t)frames * kEmitMaxChunks;
    h->emit_chunks = std::min(std::max(div_up(g.total_words, 4096), 1), kEmitMaxChunks);
    h->emit_per = div_up(div_up(g.total_words, h->emit_chunks), 256) * 256;
    ORBX_TRY(h->d_thresholds.ensure((size_t)kMaxLevels * frames));
    ORBX_TRY(h->d_totals.ensure(frames));
    ORBX_TRY(h->d_offsets.ensure(frames + 1));
    ORBX_TRY(h->h_ints.ensure((size_t)frames * (2 * kMaxLevels + 2) + 1));
    return ORBX_OK;
}

// A batch call forks the blur onto the side stream (run_detect_stage with fork_blur) and run_describe_stage joins it.
// Every other way out of that window - a capacity error, a failed allocation, the next call arriving first - must
// join it as well, or the next describe stage would wait on a stale event and skip its own blur.
void join_blur_fork(orbx_extractor* h) {
    if (!h->blur_forked) return;
    (void)hipStreamWaitEvent(h->stream, h->ev_blur, 0);
    h->blur_forked = false;
}
struct BlurForkGuard {
    orbx_extractor* h;
    ~BlurForkGuard() { join_blur_fork(h); }
};

// Fingerprint of a host image for the detect -> compute pyramid reuse: its address, geometry and a hash of EVERY
// pixel (four independent multiply chains over 32 bytes at a time: some 40 microseconds for a VGA colour frame, against
// the upload and the pyramid launches the reuse saves).  A sampled hash would let a caller who rewrites part of the
// buffer between two calls get descriptors of the old pixels; cv::ORB::compute has no such precondition.
uint64_t sample_rows(const uint8_t* image, int w, int hgt, int stride, int ch) {
    const size_t row_bytes = (size_t)w * ch;
    const int step = 1;
    uint64_t acc[4] = {0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull, 0x27d4eb2f165667c5ull};
    auto row = [&](int y) {
        const uint8_t* p = image + (size_t)y * stride;
        size_t i = 0;
        for (; i + 32 <= row_bytes; i += 32) {
            uint64_t v[4];
            std::memcpy(v, p + i, 32);
            for (int k = 0; k < 4; k++) acc[k] = (acc[k] ^ v[k]) * 0x100000001b3ull + (acc[k] >> 29);
        }
        for (; i < row_bytes; i++) acc[i & 3] = (acc[i & 3] ^ p[i]) * 0x100000001b3ull;
    };
    for (int y = 0; y < hgt; y += step) row(y);
    row(hgt - 1);
    return acc[0] ^ (acc[1] << 1) ^ (acc[2] << 2) ^ (acc[3] << 3);
}

bool same_image(const orbx_extractor* h, const uint8_t* image, int w, int hgt, int stride, int ch, uint64_t sample) {
    const orbx_extractor::ImageKey& k = h->cached_image;
    return h->image_cache_enabled && k.valid && h->pyramid_valid && k.ptr == image && k.w == w && k.h == hgt &&
           k.stride == stride && k.ch == ch && k.sample == sample;
}

void remember_image(orbx_extractor* h, const uint8_t* image, int w, int hgt, int stride, int ch, uint64_t sample) {
    h->cached_image.ptr = image;
    h->cached_image.w = w;
    h->cached_image.h = hgt;
    h->cached_image.stride = stride;
    h->cached_image.ch = ch;
    h->cached_image.sample = sample;
    h->cached_image.valid = true;
}

// A deferred batch (see orbx_extractor_finish) that the caller never finished: wait for it and drop it.
orbx_status drain_pending(orbx_extractor* h) {
    if (!h->pending.active) return ORBX_OK;
    h->pending.active = false;
    h->batch_dropped = true;
    h->last_counts.clear();
    ORBX_HIP(hipEventSynchronize(h->ev_done));
    return ORBX_OK;
}

// Launch shape of k_gray_level0: x covers the larger of its two regions (interior 16-pixel groups, edge pixels).
dim3 gray_grid(const Geom& g, int w, int hgt, int frames) {
    const int border = g.border, rx = g.L[0].x;
    const int ax_begin = (rx - border) & ~15;
    const int nq = (rx + w + border - ax_begin + 15) >> 4;
    const int n_in = std::max(((rx + w - ax_begin) >> 4) - ((rx - ax_begin + 15) >> 4), 0);
    const int per_row = std::max(n_in, (nq - n_in) * 16);
    return dim3(div_up(per_row * (hgt + 2 * border), 256), 2, frames);
}

void launch_resize(orbx_extractor* h, int l, int frames, hipStream_t s) {
    const Geom& g = h->geom;
    const ResizePlan& rp = h->rplan[l];
    const int rows = g.L[l].h + 2 * g.border;
    if (rp.narrow) {
        // lanes across a strip: 64, or 32 / 16 (with 2 / 4 row blocks per wavefront) when that leaves clearly fewer
        // lanes idle past the level's right edge
        auto used = [&](int gw) { return (double)rp.n_dwords / (div_up(rp.n_dwords, gw) * gw); };
        int gw = 64;
        // (the per-lane-group form costs about a tenth more per lane: measured, it pays from a quarter fewer idle lanes)
        if (used(32) > used(64) * 1.25) {
            gw = 32;
            if (used(16) > used(32) * 1.15) gw = 16;
        }
        const int row_blocks = div_up(rows, kResizeRows);
        dim3 grid(div_up(rp.n_dwords, gw), div_up(row_blocks, 4 * (64 / gw)), frames);
        auto kern = gw == 64 ? k_resize_strip : gw == 32 ? k_resize_strip_sub<32> : k_resize_strip_sub<16>;
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, h->d_atlas.p, h->atlas_bytes, g.atlas_w, g.L[l],
                           g.border, h->d_coefx.p + rp.cx_off, h->d_coefy.p + rp.cy_off, rp.ax_begin, rp.n_dwords);
    } else {
        dim3 grid(div_up(rp.n_dwords, 64), div_up(rows, 4), frames);
        hipLaunchKernelGGL(k_resize_level4, grid, dim3(256), 0, s, h->d_atlas.p, h->atlas_bytes, g.atlas_w, g.L[l],
                           g.border, h->d_coefx.p + rp.cx_off, h->d_coefy.p + rp.cy_off, rp.ax_begin, rp.n_dwords);
    }
}

// Harris response and orientation of every candidate above the retainBest(2 * quota) score, in one kernel.
void launch_harris_angle(orbx_extractor* h, int frames) {
    const int chunks = frames >= 64 ? 32 : 256;
    hipLaunchKernelGGL(k_harris_angle, dim3(8 * chunks * div_up(frames, 8)), dim3(256), 0, h->stream, h->d_atlas.p,
                       h->atlas_bytes, h->geom, h->d_cand.p, (const uint4*)h->d_cand.p, h->cand_cap, h->d_totals.p,
                       h->d_thresholds.p, h->umax, h->harris_scale_sq_sq, chunks, frames);
    h->harris_split = false;
}

// Orientation of the keypoints the device selection kept (after k_harris and k_select_frame).
void launch_angle_kept(orbx_extractor* h, int frames) {
    // (8 to 64 chunks of four wavefronts per frame measured the same, for this kernel and for k_harris)
    const int chunks = frames >= 64 ? 16 : 128;
    h->timer.begin(ST_ANGLE, h->stream);
    hipLaunchKernelGGL(k_angle_kept, dim3(8 * chunks * div_up(frames, 8)), dim3(256), 0, h->stream, h->d_atlas.p,
                       h->atlas_bytes, h->geom, h->d_kp.p, h->d_kp_src.p, h->kp_cap, h->d_kp_counts.p, h->d_status.p,
                       h->umax, chunks, frames);
    h->timer.end(ST_ANGLE, h->stream);
}

// Ordered emission of the corners the FAST kernel marked, then Harris response and orientation of the candidates
// (split: the response only - the caller runs the device selection and then launch_angle_kept).
void run_emit_and_harris(orbx_extractor* h, int frames, bool split = false) {
    const Geom& g = h->geom;
    hipStream_t s = h->stream;
    size_t A = h->atlas_bytes;
    StageTimer& T = h->timer;
    T.begin(ST_EMIT, s);
    {
        const int chunks = h->emit_chunks, per = h->emit_per;
        hipLaunchKernelGGL(k_emit_write, dim3(chunks, frames), dim3(256), 0, s, h->d_mask.p, h->d_score.p, A, g, chunks,
                           per, h->d_chunk_counts_p, h->d_cand.p, h->cand_cap, h->d_level_counts_p, h->d_hist_p);
        hipLaunchKernelGGL(k_emit_finish, dim3(frames), dim3(256), 0, s, g, chunks, h->d_chunk_counts_p,
                           h->d_level_counts_p, h->d_hist_p, h->d_thresholds.p, h->d_totals.p);
    }
    T.end(ST_EMIT, s);
    T.begin(ST_HARRIS, s);
    if (split && h->distribute_mode) {
        // the distribution mode ranks by corner score: no Harris response is needed
        h->harris_split = true;
    } else if (split) {
        const int chunks = frames >= 64 ? 16 : 128;
        hipLaunchKernelGGL(k_harris, dim3(8 * chunks * div_up(frames, 8)), dim3(256), 0, s, h->d_atlas.p, A, g,
                           h->d_cand.p, h->cand_cap, h->d_totals.p, h->d_thresholds.p, h->harris_scale_sq_sq, chunks,
                           frames);
        h->harris_split = true;
    } else {
        launch_harris_angle(h, frames);
    }
    T.end(ST_HARRIS, s);
}

// A frame with more corners than the candidate list holds (cv::ORB has no such limit): the list is enlarged to the
// largest per-frame total and emission + Harris/angle run again on the masks the FAST kernel left.  The per-run
// corner counts come from the FAST kernel and stay; the per-level counts and histograms are accumulated by the
// emission and start again from zero.
orbx_status regrow_candidates(orbx_extractor* h, int frames, int needed) {
    h->cand_cap = (needed + 1023) & ~1023;
    ORBX_TRY(h->d_cand.ensure((size_t)h->cand_cap * frames));
    ORBX_TRY(h->d_packed.ensure((size_t)h->cand_cap * frames));
    hipStream_t s = h->stream;
    ORBX_HIP(hipMemsetAsync(h->d_level_counts_p, 0, sizeof(int) * (size_t)frames * kMaxLevels, s));
    ORBX_HIP(hipMemsetAsync(h->d_hist_p, 0, sizeof(int) * (size_t)frames * kMaxLevels * 256, s));
    run_emit_and_harris(h, frames);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

void launch_pyramid(orbx_extractor* h, const uint8_t* d_images, int frames, size_t frame_stride, int w, int hgt,
                    int stride, int ch) {
    const Geom& g = h->geom;
    hipStream_t s = h->stream;
    StageTimer& T = h->timer;
    if (use_stream_pyramid(h, frames)) {
        // (the stage timers book the whole pass under "gray"; "resize" stays zero)
        T.begin(ST_GRAY, s);
        hipLaunchKernelGGL(k_pyramid_stream<false>, dim3(frames), dim3(256), h->splan.lds_bytes, s, d_images, frame_stride,
                           stride, ch, h->d_atlas.p, h->atlas_bytes, h->splan.prm, h->d_scols.p, h->d_srows.p,
                           h->d_ssched.p);
        T.end(ST_GRAY, s);
        return;
    }
    T.begin(ST_GRAY, s);
    {
        dim3 grid = gray_grid(g, w, hgt, frames);
        hipLaunchKernelGGL(k_gray_level0, grid, dim3(256), 0, s, d_images, frame_stride, w, hgt, stride, ch,
                           h->d_atlas.p, h->atlas_bytes, g.atlas_w, g.L[0].x, g.L[0].y, g.border);
    }
    T.end(ST_GRAY, s);
    T.begin(ST_RESIZE, s);
    // On request the small levels above tail_base take one launch together (orbx_extractor_set_pyramid_tail).
    // Measured on 792 VGA frames, eight levels, alone on the chip: 0.565 ms for the seven launches level by level;
    // 0.58 / 0.60 / 0.60 ms with levels 5-7 / 4-7 / 3-7 in one launch (16, 8, 8 rows a step) - the workgroup-per-frame
    // walk needs some 20 to 40 dependent steps of a few microseconds each, while a small level's own launch takes 40
    // to 70 us.  Off by default.
    const int last_single = h->tplan.ok ? h->tail_base : g.nlevels - 1;
    for (int l = 1; l <= last_single; l++) launch_resize(h, l, frames, s);
    if (h->tplan.ok) {
        const LevelInfo& Lb = g.L[h->tail_base];
        hipLaunchKernelGGL(k_pyramid_stream<true>, dim3(frames), dim3(256), h->tplan.lds_bytes, s,
                           h->d_atlas.p + (size_t)Lb.y * g.atlas_w + Lb.x, h->atlas_bytes, g.atlas_w, 1, h->d_atlas.p,
                           h->atlas_bytes, h->tplan.prm, h->d_tcols.p, h->d_trows.p, h->d_tsched.p);
    }
    T.end(ST_RESIZE, s);
}

// Pyramid + FAST + NMS + ordered emission + Harris/angle for `frames` device-resident images.
orbx_status run_detect_stage(orbx_extractor* h, const uint8_t* d_images, int frames, size_t frame_stride, int w,
                             int hgt, int stride, int ch, bool fork_blur = false, bool split_harris = false) {
    const Geom& g = h->geom;
    hipStream_t s = h->stream;
    size_t A = h->atlas_bytes;
    StageTimer& T = h->timer;
    T.clear();
    join_blur_fork(h);
    launch_pyramid(h, d_images, frames, frame_stride, w, hgt, stride, ch);
    T.begin(ST_FAST, s);
    ORBX_HIP(hipMemsetAsync(h->d_acc.p, 0, sizeof(int) * h->acc_ints, s));
    hipLaunchKernelGGL(k_fast_nms<false>, dim3(div_up(g.total_tiles, 8) * 8, frames), dim3(256), 0, s, h->d_atlas.p, h->d_score.p,
                       h->d_mask.p, A, g, h->d_tiles.p,
                       std::min(std::max(h->distribute_mode ? h->distribute_low : h->prm.score_threshold, 0), 255), 0,
                       h->d_chunk_counts_p, h->emit_per);
    T.end(ST_FAST, s);
    // forked here, not right after the pyramid: FAST fills the chip by itself, emission and selection do not
    if (fork_blur) {
        ORBX_HIP(hipEventRecord(h->ev_pyramid, s));
        ORBX_HIP(hipStreamWaitEvent(h->side_stream, h->ev_pyramid, 0));
        T.begin(ST_BLUR, h->side_stream);
        hipLaunchKernelGGL(k_blur, dim3(div_up((int)h->strips.size(), 4), frames), dim3(256), 0, h->side_stream,
                           h->d_atlas.p, h->d_blur.p, A, g, h->d_strips.p, (int)h->strips.size(), h->blur_taps);
        T.end(ST_BLUR, h->side_stream);
        ORBX_HIP(hipEventRecord(h->ev_blur, h->side_stream));
        h->blur_forked = true;
    }
    run_emit_and_harris(h, frames, split_harris);
    ORBX_HIP(hipGetLastError());
    h->pyramid_valid = true;
    return ORBX_OK;
}

// Pyramid only (for orbx_compute on its own).
orbx_status run_pyramid(orbx_extractor* h, const uint8_t* d_images, int frames, size_t frame_stride, int w, int hgt,
                        int stride, int ch) {
    join_blur_fork(h);
    launch_pyramid(h, d_images, frames, frame_stride, w, hgt, stride, ch);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

// Blur + rBRIEF for keypoints already on the device (d_kp / d_kp_counts).
orbx_status run_describe_stage(orbx_extractor* h, int frames) {
    const Geom& g = h->geom;
    hipStream_t s = h->stream;
    size_t A = h->atlas_bytes;
    if (h->blur_forked) {
        ORBX_HIP(hipStreamWaitEvent(s, h->ev_blur, 0));
        h->blur_forked = false;
    } else {
        h->timer.begin(ST_BLUR, s);
        hipLaunchKernelGGL(k_blur, dim3(div_up((int)h->strips.size(), 4), frames), dim3(256), 0, s, h->d_atlas.p,
                           h->d_blur.p, A, g, h->d_strips.p, (int)h->strips.size(), h->blur_taps);
        h->timer.end(ST_BLUR, s);
    }
    ORBX_TRY(h->d_steer.ensure((size_t)h->kp_cap * frames));
    h->timer.begin(ST_BRIEF, s);
    hipLaunchKernelGGL(k_steer, dim3(div_up(h->kp_cap, 256), frames), dim3(256), 0, s, g, h->d_kp.p, h->d_kp_counts.p,
                       h->kp_cap, h->d_steer.p);
    {
        const int chunks = div_up(h->kp_cap, 8);
        hipLaunchKernelGGL(k_brief, dim3(8 * chunks * div_up(frames, 8)), dim3(256), 0, s, h->d_atlas.p, h->d_blur.p, A,
                           g.atlas_w, h->d_steer.p, h->d_kp_counts.p, h->kp_cap, h->d_desc.p, h->d_pattern.p, chunks,
                           frames);
    }
    h->timer.end(ST_BRIEF, s);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

void collect_timings(orbx_extractor* h) {
    for (int i = 0; i < ST_COUNT; i++) h->last_ms[i] = h->timer.ms(i);
}

// Downloads the candidates of all frames and runs the host part of cv::ORB::detect per frame.
// out[f] receives the final keypoints of frame f (level-0 coordinates).
orbx_status select_keypoints(orbx_extractor* h, int frames, std::vector<std::vector<orbx_keypoint>>& out,
                             bool regrown = false) {
    const Geom& g = h->geom;
    hipStream_t s = h->stream;
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(64), 0, s, h->d_totals.p, frames, h->cand_cap, h->d_offsets.p);
    hipLaunchKernelGGL(k_pack_cands, dim3(div_up(h->cand_cap, 256), frames), dim3(256), 0, s, h->d_cand.p,
                       h->cand_cap, h->d_totals.p, h->d_offsets.p, h->d_packed.p);
    int* hi = h->h_ints.p;
    int* h_totals = hi;
    int* h_offsets = hi + frames;
    int* h_counts = hi + 2 * frames + 1;
    ORBX_HIP(hipMemcpyAsync(h_totals, h->d_totals.p, sizeof(int) * frames, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(h_offsets, h->d_offsets.p, sizeof(int) * (frames + 1), hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipMemcpyAsync(h_counts, h->d_level_counts_p, sizeof(int) * kMaxLevels * frames, hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));
    int most = 0;
    for (int f = 0; f < frames; f++) most = std::max(most, h_totals[f]);
    if (most > h->cand_cap) {
        if (regrown) return ORBX_CAPACITY;  // cannot happen: the list was sized for the totals just measured
        ORBX_TRY(regrow_candidates(h, frames, most));
        return select_keypoints(h, frames, out, true);
    }
    size_t total = (size_t)h_offsets[frames];
    ORBX_TRY(h->h_packed.ensure(std::max<size_t>(total, 1)));
    if (total)
        ORBX_HIP(hipMemcpyAsync(h->h_packed.p, h->d_packed.p, total * sizeof(Cand), hipMemcpyDeviceToHost, s));
    ORBX_HIP(hipStreamSynchronize(s));

    out.assign(frames, {});
    int quotas[kMaxLevels];
    float scales[kMaxLevels];
    for (int l = 0; l < g.nlevels; l++) {
        quotas[l] = g.L[l].quota;
        scales[l] = g.L[l].scale;
    }
    auto work = [&](int f0, int f1) {
        for (int f = f0; f < f1; f++)
            host::select_keypoints(h->h_packed.p + h_offsets[f], h_totals[f], h_counts + f * kMaxLevels, g.nlevels,
                                   quotas, scales, h->prm.patch_size, out[f]);
    };
    int nthreads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    nthreads = std::min(nthreads, frames);
    if (nthreads <= 1) work(0, frames);
    else {
        std::vector<std::thread> th;
        int per = div_up(frames, nthreads);
        for (int t = 0; t < nthreads; t++) {
            int f0 = t * per, f1 = std::min(frames, f0 + per);
            if (f0 < f1) th.emplace_back(work, f0, f1);
        }
        for (auto& t : th) t.join();
    }
    return ORBX_OK;
}

orbx_status check_image_args(const orbx_extractor* h, const void* image, int w, int hgt, int stride, int ch) {
    if (!h || !image) return ORBX_BAD_ARG;
    if (ch != 1 && ch != 3) return ORBX_BAD_ARG;
    if (w <= 0 || hgt <= 0 || stride < w * ch) return ORBX_BAD_ARG;
    if (w > 16384 || hgt > 16384) return ORBX_UNSUPPORTED;  // tile and strip origins are 16-bit
    return ORBX_OK;
}

}  // namespace

extern "C" {

orbx_status orbx_extractor_create(const orbx_extractor_params* params, int max_width, int max_height, int max_batch,
                                  int device, orbx_extractor** out) {
    if (!params || !out || max_width <= 0 || max_height <= 0 || max_batch <= 0) return ORBX_BAD_ARG;
    if (params->first_level != 0 || params->wta_k != 2 || params->patch_size != 31) return ORBX_UNSUPPORTED;
    if (params->level_pyramid < 1 || params->level_pyramid > kMaxLevels) return ORBX_UNSUPPORTED;
    if (params->n_key_points < 0 || !(params->scale_factor > 1.0f) || params->edge_threshold < 0) return ORBX_BAD_ARG;
    if (max_width > 16384 || max_height > 16384) return ORBX_UNSUPPORTED;
    ORBX_HIP(hipSetDevice(device));
    orbx_extractor* h = new orbx_extractor();
    h->prm = *params;
    h->device = device;
    h->max_w = max_width;
    h->max_h = max_height;
    h->max_batch = max_batch;
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return ORBX_DEVICE;
    }
    h->stream = h->own_stream;
    if (hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_pyramid, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_blur, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming) != hipSuccess) {
        orbx_extractor_destroy(h);
        return ORBX_DEVICE;
    }
    orbx_status st = h->timer.init();
    if (st != ORBX_OK) { orbx_extractor_destroy(h); return st; }

    // cv::ORB computeKeyPoints(): umax table of the circular patch
    {
        int half_patch = params->patch_size / 2;
        int umax[32] = {};
        int v, v0, vmax = cv_floor(half_patch * std::sqrt(2.f) / 2 + 1);
        int vmin = cv_ceil(half_patch * std::sqrt(2.f) / 2);
        for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt((double)half_patch * half_patch - v * v));
        for (v = half_patch, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
        for (int i = 0; i < 16; i++) h->umax.v[i] = umax[i];
    }
    // cv::getGaussianKernel(7, 2, CV_32F) converted to Q8 (createSeparableLinearFilter, 8-bit symmetric smooth)
    {
        const int n = 7;
        const double sigma = 2.0;
        double scale2x = -0.5 / (sigma * sigma);
        float cf[7];
        int k[7];
        double sum = 0;
        for (int i = 0; i < n; i++) {
            double x = i - (n - 1) * 0.5;
            cf[i] = (float)std::exp(scale2x * x * x);
            sum += cf[i];
        }
        sum = 1. / sum;
        for (int i = 0; i < n; i++) {
            cf[i] = (float)(cf[i] * sum);
            k[i] = (int)lrintf(cf[i] * 256.f);
        }
        // the kernel packs the taps for v_dot4_u32_u8: they must be symmetric bytes
        if (k[0] != k[6] || k[1] != k[5] || k[2] != k[4] || k[3] > 255 || k[0] < 0) { orbx_extractor_destroy(h); return ORBX_UNSUPPORTED; }
        h->blur_taps.wa = (unsigned)k[0] | ((unsigned)k[1] << 8) | ((unsigned)k[2] << 16) | ((unsigned)k[3] << 24);
        h->blur_taps.wb = (unsigned)k[2] | ((unsigned)k[1] << 8) | ((unsigned)k[0] << 16);
        for (int i = 0; i < 4; i++) h->blur_taps.k[i] = k[i];
    }
    {
        float scale = 1.f / ((1 << 2) * 7 * 255.f);
        h->harris_scale_sq_sq = scale * scale * scale * scale;
    }
    BriefPattern bp;
    for (int i = 0; i < 1024; i++) bp.p[i] = (int8_t)kBitPattern31[i];
    st = h->d_pattern.ensure(1);
    if (st == ORBX_OK && hipMemcpy(h->d_pattern.p, &bp, sizeof(bp), hipMemcpyHostToDevice) != hipSuccess) st = ORBX_DEVICE;
    if (st == ORBX_OK) st = h->d_in.ensure((size_t)max_width * max_height * 3);
    if (st != ORBX_OK) { orbx_extractor_destroy(h); return st; }
    *out = h;
    return ORBX_OK;
}

void orbx_extractor_destroy(orbx_extractor* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    h->d_tiles.release(); h->d_strips.release(); h->d_coefx.release(); h->d_coefy.release(); h->d_pattern.release();
    h->d_in.release(); h->d_atlas.release(); h->d_score.release(); h->d_blur.release(); h->d_desc.release();
    h->d_mask.release(); h->d_steer.release(); h->d_cand.release(); h->d_packed.release();
    h->d_acc.release();
    h->d_thresholds.release(); h->d_totals.release(); h->d_offsets.release();
    h->d_kp_counts.release(); h->d_kp.release(); h->d_kp_src.release();
    h->h_packed.release(); h->h_ints.release(); h->h_kp.release(); h->h_desc.release();
    h->d_cell_start.release(); h->d_cell_items.release(); h->h_cell_start.release(); h->h_cell_items.release();
    h->d_status.release();
    h->d_select_ws.release();
    h->d_scols.release(); h->d_srows.release(); h->d_ssched.release();
    h->d_tcols.release(); h->d_trows.release(); h->d_tsched.release();
    h->d_labels.release(); h->d_picked_idx.release(); h->d_picked_n.release();
    h->timer.destroy();
    if (h->side_stream) { (void)hipStreamSynchronize(h->side_stream); (void)hipStreamDestroy(h->side_stream); }
    if (h->ev_pyramid) (void)hipEventDestroy(h->ev_pyramid);
    if (h->ev_blur) (void)hipEventDestroy(h->ev_blur);
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

orbx_status orbx_extractor_set_stream(orbx_extractor* h, void* hip_stream) {
    ORBX_ENTER(h);
    if (!h) return ORBX_BAD_ARG;
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return ORBX_OK;
}

orbx_status orbx_extractor_scale_tables(const orbx_extractor* h, float* sf, float* isf, float* ss, float* iss,
                                        float* log_sf) {
    if (!h) return ORBX_BAD_ARG;
    host::scale_tables(h->prm.scale_factor, h->prm.level_pyramid, sf, isf, ss, iss, log_sf);
    return ORBX_OK;
}

orbx_status orbx_extractor_layout(const orbx_extractor* hc, int width, int height, int* atlas_w, int* atlas_h,
                                  int32_t* rects, float* scales) {
    orbx_extractor* h = const_cast<orbx_extractor*>(hc);
    ORBX_ENTER(h);
    if (!h || width <= 0 || height <= 0) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    ORBX_TRY(build_geometry(h, width, height, h->prm.level_pyramid));
    if (atlas_w) *atlas_w = h->geom.atlas_w;
    if (atlas_h) *atlas_h = h->geom.atlas_h;
    for (int l = 0; l < h->geom.nlevels; l++) {
        if (rects) {
            rects[4 * l] = h->geom.L[l].x;
            rects[4 * l + 1] = h->geom.L[l].y;
            rects[4 * l + 2] = h->geom.L[l].w;
            rects[4 * l + 3] = h->geom.L[l].h;
        }
        if (scales) scales[l] = h->geom.L[l].scale;
    }
    return ORBX_OK;
}

orbx_status orbx_extractor_read_plane(orbx_extractor* h, int frame, int what, uint8_t* out, size_t out_bytes) {
    ORBX_ENTER(h);
    if (!h || !out || frame < 0 || frame >= std::max(h->last_frames, 1)) return ORBX_BAD_ARG;
    if (out_bytes < h->atlas_bytes) return ORBX_CAPACITY;
    ORBX_HIP(hipSetDevice(h->device));
    ORBX_TRY(drain_pending(h));
    const uint8_t* src = what == 0 ? h->d_atlas.p : what == 1 ? h->d_score.p : what == 2 ? h->d_blur.p : nullptr;
    if (!src) return ORBX_BAD_ARG;
    if (what == 1) {
        // the detect stage stores scores only where keypoints survive; fill in the whole plane of this frame
        if (!h->pyramid_valid) return ORBX_BAD_ARG;
        ORBX_HIP(hipMemsetAsync(h->d_score.p + (size_t)frame * h->atlas_bytes, 0, h->atlas_bytes, h->stream));
        hipLaunchKernelGGL(k_fast_nms<true>, dim3(div_up(h->geom.total_tiles, 8) * 8, 1), dim3(256), 0, h->stream, h->d_atlas.p,
                           h->d_score.p, h->d_mask.p, h->atlas_bytes, h->geom, h->d_tiles.p,
                           std::min(std::max(h->prm.score_threshold, 0), 255), frame, nullptr, 1);
        ORBX_HIP(hipGetLastError());
    }
    ORBX_HIP(hipStreamSynchronize(h->stream));
    ORBX_HIP(hipMemcpy(out, src + (size_t)frame * h->atlas_bytes, h->atlas_bytes, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

orbx_status orbx_selftest_retain_best_device(int device, const float* keys, int n, int n_points, int32_t* perm_out,
                                             int* n_out) {
    if (!keys || !perm_out || !n_out || n < 0 || n > kSelectMaxCands) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(device));
    DevBuf<float> d_keys;
    DevBuf<int> d_out;
    ORBX_TRY(d_keys.ensure(std::max(n, 1)));
    ORBX_TRY(d_out.ensure((size_t)n + 1));
    ORBX_HIP(hipMemcpy(d_keys.p, keys, sizeof(float) * n, hipMemcpyHostToDevice));
    const size_t shmem = (size_t)std::max(n, 1) * (sizeof(stdsel::Elem) + 2 * sizeof(uint16_t)) + 16;
    ORBX_TRY(ensure_dynamic_lds((const void*)k_selftest_retain_best, shmem));
    hipLaunchKernelGGL(k_selftest_retain_best, dim3(1), dim3(64), shmem, 0, d_keys.p, n, n_points, d_out.p + 1, d_out.p);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipDeviceSynchronize());
    ORBX_HIP(hipMemcpy(n_out, d_out.p, sizeof(int), hipMemcpyDeviceToHost));
    if (*n_out > 0) ORBX_HIP(hipMemcpy(perm_out, d_out.p + 1, sizeof(int) * (size_t)*n_out, hipMemcpyDeviceToHost));
    d_keys.release();
    d_out.release();
    return ORBX_OK;
}

int orbx_extractor_last_timings(const orbx_extractor* h, float* ms, int cap) {
    if (!h || !ms) return 0;
    int n = std::min(cap, (int)ST_COUNT);
    for (int i = 0; i < n; i++) ms[i] = h->last_ms[i];
    return n;
}

orbx_status orbx_detect(orbx_extractor* h, const uint8_t* image, int width, int height, int stride, int channels,
                        orbx_keypoint* keypoints, int capacity, int* count) {
    ORBX_ENTER(h);
    if (!count || capacity < 0 || (capacity > 0 && !keypoints)) return ORBX_BAD_ARG;
    *count = 0;
    ORBX_TRY(check_image_args(h, image, width, height, stride, channels));
    ORBX_HIP(hipSetDevice(h->device));
    ORBX_TRY(drain_pending(h));
    if (h->distribute_mode) {
        // the distribution mode lives in the batch path: a batch of one, keypoints only
        const size_t nbytes = (size_t)stride * height;
        ORBX_TRY(h->d_in.ensure(nbytes));
        ORBX_HIP(hipMemcpyAsync(h->d_in.p, image, nbytes, hipMemcpyHostToDevice, h->stream));
        const int cap = std::max(capacity, h->prm.n_key_points + 4 * kMaxLevels);
        std::vector<orbx_keypoint> tmp((size_t)cap);
        int32_t n = 0;
        ORBX_TRY(orbx_extract_batch(h, h->d_in.p, 1, nbytes, width, height, stride, channels, tmp.data(), cap, &n, nullptr));
        *count = n;
        if (n > capacity) return ORBX_CAPACITY;
        std::copy(tmp.begin(), tmp.begin() + n, keypoints);
        return ORBX_OK;
    }
    ORBX_TRY(build_geometry(h, width, height, h->prm.level_pyramid));
    ORBX_TRY(ensure_buffers(h, 1));
    size_t bytes = (size_t)stride * height;
    ORBX_TRY(h->d_in.ensure(bytes));
    h->cached_image.valid = false;
    ORBX_HIP(hipMemcpyAsync(h->d_in.p, image, bytes, hipMemcpyHostToDevice, h->stream));
    h->last_frames = 1;
    ORBX_TRY(run_detect_stage(h, h->d_in.p, 1, bytes, width, height, stride, channels));
    if (h->image_cache_enabled)
        remember_image(h, image, width, height, stride, channels, sample_rows(image, width, height, stride, channels));
    std::vector<std::vector<orbx_keypoint>> sel;
    ORBX_TRY(select_keypoints(h, 1, sel));
    collect_timings(h);
    *count = (int)sel[0].size();
    if (*count > capacity) return ORBX_CAPACITY;
    std::copy(sel[0].begin(), sel[0].end(), keypoints);
    return ORBX_OK;
}

orbx_status orbx_compute(orbx_extractor* h, const uint8_t* image, int width, int height, int stride, int channels,
                         orbx_keypoint* keypoints, int* count, uint8_t* descriptors) {
    ORBX_ENTER(h);
    if (!count || *count < 0 || (*count > 0 && (!keypoints || !descriptors))) return ORBX_BAD_ARG;
    ORBX_TRY(check_image_args(h, image, width, height, stride, channels));
    int n = *count;
    // cv::ORB detectAndCompute(useProvidedKeypoints): number of levels from the keypoints
    int nlevels = 0;
    bool sorted = true;
    for (int i = 0; i < n; i++) {
        int level = keypoints[i].octave;
        if (level < 0) return ORBX_BAD_ARG;
        if (i > 0 && level < keypoints[i - 1].octave) sorted = false;
        nlevels = std::max(nlevels, level);
    }
    nlevels++;
    if (nlevels > kMaxLevels) return ORBX_UNSUPPORTED;
    std::vector<orbx_keypoint> k(keypoints, keypoints + n);
    host::run_by_image_border(k, width, height, h->prm.edge_threshold);
    if (!sorted) host::regroup_by_level(k, nlevels);
    if (k.empty()) { *count = 0; return ORBX_OK; }

    ORBX_HIP(hipSetDevice(h->device));
    ORBX_TRY(drain_pending(h));
    // a geometry change (more levels than the extractor's own) invalidates the cached pyramid by itself
    ORBX_TRY(build_geometry(h, width, height, std::max(nlevels, h->prm.level_pyramid)));
    ORBX_TRY(ensure_buffers(h, 1));
    size_t bytes = (size_t)stride * height;
    h->last_frames = 1;
    h->timer.clear();
    const uint64_t sample = h->image_cache_enabled ? sample_rows(image, width, height, stride, channels) : 0;
    if (same_image(h, image, width, height, stride, channels, sample)) {
        join_blur_fork(h);  // nothing to upload, nothing to build: the pyramid of this image is in the atlas
    } else {
        ORBX_TRY(h->d_in.ensure(bytes));
        h->cached_image.valid = false;
        ORBX_HIP(hipMemcpyAsync(h->d_in.p, image, bytes, hipMemcpyHostToDevice, h->stream));
        ORBX_TRY(run_pyramid(h, h->d_in.p, 1, bytes, width, height, stride, channels));
        h->pyramid_valid = true;
        if (h->image_cache_enabled) remember_image(h, image, width, height, stride, channels, sample);
    }
    int m = (int)k.size();
    if (m > h->kp_cap) h->kp_cap = m;
    ORBX_TRY(h->d_kp.ensure((size_t)h->kp_cap));
    ORBX_TRY(h->d_desc.ensure((size_t)h->kp_cap * 32));
    ORBX_TRY(h->d_kp_counts.ensure(1));
    ORBX_HIP(hipMemcpyAsync(h->d_kp.p, k.data(), sizeof(orbx_keypoint) * m, hipMemcpyHostToDevice, h->stream));
    ORBX_HIP(hipMemcpyAsync(h->d_kp_counts.p, &m, sizeof(int), hipMemcpyHostToDevice, h->stream));
    ORBX_TRY(run_describe_stage(h, 1));
    ORBX_HIP(hipMemcpyAsync(descriptors, h->d_desc.p, (size_t)m * 32, hipMemcpyDeviceToHost, h->stream));
    ORBX_HIP(hipStreamSynchronize(h->stream));
    collect_timings(h);
    std::copy(k.begin(), k.end(), keypoints);
    *count = m;
    return ORBX_OK;
}

}  // extern "C"

// The opt-in distribution mode: every (frame, level) chooses its keypoints from the candidate list (per-cell
// adaptive threshold, quad-tree); k_select_frame then takes the chosen candidates instead of running the two culls.
static orbx_status launch_distribute(orbx_extractor* h, int n_frames) {
    const Geom& g = h->geom;
    int max_quota = 0, cell_words = 1;
    const int lo = std::max(3, g.edge);
    for (int l = 0; l < g.nlevels; l++) {
        max_quota = std::max(max_quota, g.L[l].quota);
        const int aw = g.L[l].w - 2 * lo, ah = g.L[l].h - 2 * lo;
        if (aw <= 0 || ah <= 0) continue;
        const int csz = std::max(h->distribute_cell, 1);
        const int ncol = std::max(aw / csz, 1), nrow = std::max(ah / csz, 1);
        cell_words = std::max(cell_words, (ncol * nrow + 31) / 32);
    }
    DistributeParams dp;
    dp.high_threshold = std::min(std::max(h->prm.score_threshold, 0), 255);
    dp.cell_size = h->distribute_cell;
    dp.max_nodes = std::max(max_quota + 4, 16);
    dp.cell_words = cell_words;
    dp.out_stride = dp.max_nodes;
    if (dp.max_nodes > 60000) return ORBX_UNSUPPORTED;   // box labels are 16 bits
    const size_t lds = sizeof(int) * ((size_t)13 * dp.max_nodes + dp.cell_words);
    if (lds > 150 * 1024) return ORBX_UNSUPPORTED;
    ORBX_TRY(ensure_dynamic_lds((const void*)k_distribute_level, lds));
    ORBX_TRY(h->d_labels.ensure((size_t)h->cand_cap * n_frames));
    ORBX_TRY(h->d_picked_idx.ensure((size_t)n_frames * kMaxLevels * dp.out_stride));
    ORBX_TRY(h->d_picked_n.ensure((size_t)n_frames * kMaxLevels));
    h->picked_stride = dp.out_stride;
    ORBX_HIP(hipMemsetAsync(h->d_picked_n.p, 0, sizeof(int) * (size_t)n_frames * kMaxLevels, h->stream));
    hipLaunchKernelGGL(k_distribute_level, dim3(g.nlevels, n_frames), dim3(kDistributeThreads), lds, h->stream,
                       h->d_cand.p, h->cand_cap, h->d_level_counts_p, h->d_totals.p, g, dp, h->d_labels.p,
                       h->d_picked_idx.p, h->d_picked_n.p);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

// Frames the device selection left at status 2 because the global form ran out of workspace slices (more frames
// beyond the LDS form than slices) get further passes of the global form, a slice-load at a time; the describe stage is
// then repeated for the batch.  Returns with *left = number of frames still at status 2 (their candidate list itself
// overflowed: the host path enlarges it).  The status and count words are in h->h_ints when this returns.
static orbx_status more_global_passes(orbx_extractor* h, int n_frames, int capacity, int* left) {
    int* h_status = h->h_ints.p;
    int* h_cnt = h->h_ints.p + n_frames;
    auto count_left = [&]() {
        int n = 0;
        for (int f = 0; f < n_frames; f++) n += h_status[f] == 2;
        return n;
    };
    int before = count_left();
    *left = before;
    if (h->sel_ws_slices <= 0) return ORBX_OK;
    int* ws_counter = (int*)(h->d_select_ws.p + h->sel_ws_stride * h->sel_ws_slices);
    while (before > 0) {
        ORBX_HIP(hipMemsetAsync(ws_counter, 0, sizeof(int), h->stream));
        hipLaunchKernelGGL(k_select_frame<true>, dim3(n_frames), dim3(64 * h->geom.nlevels), 0, h->stream, h->d_cand.p,
                           h->cand_cap, h->d_level_counts_p, h->d_totals.p, h->geom, h->sel_fm, h->d_kp.p, capacity,
                           h->d_kp_counts.p, h->d_cell_start.p, h->d_cell_items.p, h->d_status.p, h->cand_cap, 2,
                           h->d_select_ws.p, h->sel_ws_stride, h->sel_ws_slices, ws_counter, h->d_kp_src.p);
        ORBX_HIP(hipMemcpyAsync(h_status, h->d_status.p, sizeof(int) * n_frames, hipMemcpyDeviceToHost, h->stream));
        ORBX_HIP(hipMemcpyAsync(h_cnt, h->d_kp_counts.p, sizeof(int) * n_frames, hipMemcpyDeviceToHost, h->stream));
        ORBX_HIP(hipStreamSynchronize(h->stream));
        const int now = count_left();
        if (now >= before) break;  // no progress: what is left overflowed the candidate list
        before = now;
    }
    *left = before;
    if (before == 0) {
        launch_angle_kept(h, n_frames);   // every frame again: the same angles for those that had them
        ORBX_TRY(run_describe_stage(h, n_frames));
        ORBX_HIP(hipStreamSynchronize(h->stream));
    }
    return ORBX_OK;
}

// orbx_extract_batch from the selection on, by the host path: candidates are downloaded, cv::ORB's two retainBest
// culls run with the real std:: calls, then undistortion, grid and border filter as MonoFrame::extractFeatures does
// them (mono_frame.cpp:85-100), and the keypoints go back to the device for the describe stage.  No limit on
// candidates or keypoint slots.  The detect stage of the batch must be what the device buffers hold.
static orbx_status batch_host_path(orbx_extractor* h, int n_frames, int width, int height, int capacity,
                                   orbx_keypoint* keypoints, int32_t* counts, uint8_t* descriptors) {
    // the host selection reads the orientation from the candidates: a detect stage that left it to the device
    // selection gets the combined kernel now (the same responses again, and the angles)
    if (h->harris_split) launch_harris_angle(h, n_frames);
    std::vector<std::vector<orbx_keypoint>> sel;
    ORBX_TRY(select_keypoints(h, n_frames, sel));
    // MonoFrame::extractFeatures between detect and compute (mono_frame.cpp:85-100): undistort, bucket, filter
    const int ncell = h->has_grid ? h->grid.grid_cols * h->grid.grid_rows : 0;
    if (h->has_grid) {
        ORBX_TRY(h->h_cell_start.ensure((size_t)(ncell + 1) * n_frames));
        ORBX_TRY(h->h_cell_items.ensure((size_t)capacity * n_frames));
    }
    int max_n = 0;
    for (auto& v : sel) max_n = std::max(max_n, (int)v.size());
    if (max_n > capacity) return ORBX_CAPACITY;
    if (h->has_camera || h->has_grid) {
        auto work = [&](int f0, int f1) {
            std::vector<float> xy;
            for (int f = f0; f < f1; f++) {
                auto& k = sel[f];
                if (h->has_camera && h->camera.k1 != 0.0 && !k.empty()) {
                    xy.resize(k.size() * 2);
                    for (size_t i = 0; i < k.size(); i++) { xy[2 * i] = k[i].x; xy[2 * i + 1] = k[i].y; }
                    host::undistort_points(h->camera, xy.data(), (int)k.size(), xy.data());
                    for (size_t i = 0; i < k.size(); i++) { k[i].x = xy[2 * i]; k[i].y = xy[2 * i + 1]; }
                }
                if (h->has_grid) {
                    int cnt = (int)k.size();
                    (void)orbx_assign_grid(&h->grid, k.data(), &cnt, h->h_cell_start.p + (size_t)f * (ncell + 1),
                                           h->h_cell_items.p + (size_t)f * capacity);
                    k.resize(cnt);
                }
                // cv::ORB::compute drops keypoints whose rounded position is within the border (it may, after
                // undistortion); the grid built above then refers to the unfiltered list, as in the reference.
            }
        };
        int nthreads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
        nthreads = std::min(nthreads, n_frames);
        std::vector<std::thread> th;
        int per = div_up(n_frames, nthreads);
        for (int t = 0; t < nthreads; t++) {
            int f0 = t * per, f1 = std::min(n_frames, f0 + per);
            if (f0 < f1) th.emplace_back(work, f0, f1);
        }
        for (auto& t : th) t.join();
    }
    h->kp_cap = capacity;
    ORBX_TRY(h->d_kp.ensure((size_t)capacity * n_frames));
    ORBX_TRY(h->d_desc.ensure((size_t)capacity * n_frames * 32));
    ORBX_TRY(h->d_kp_counts.ensure(n_frames));
    ORBX_TRY(h->h_kp.ensure((size_t)capacity * n_frames));
    int* h_counts = h->h_ints.p;
    for (int f = 0; f < n_frames; f++) {
        // cv::ORB::compute: runByImageBorder on the (possibly moved) keypoints; octaves are already ascending
        const size_t pre = sel[f].size();
        if (h->has_camera) host::run_by_image_border(sel[f], width, height, h->prm.edge_threshold);
        h_counts[f] = (int)sel[f].size();
        std::copy(sel[f].begin(), sel[f].end(), h->h_kp.p + (size_t)f * capacity);
        // grid entries of the dropped keypoints name the slots past the new end: a feature no window accepts
        for (size_t i = sel[f].size(); i < pre; i++)
            h->h_kp.p[(size_t)f * capacity + i] = orbx_keypoint{0.f, 0.f, 0.f, -1.f, 0.f, -128, -1};
    }
    ORBX_HIP(hipMemcpyAsync(h->d_kp.p, h->h_kp.p, sizeof(orbx_keypoint) * (size_t)capacity * n_frames,
                            hipMemcpyHostToDevice, h->stream));
    ORBX_HIP(hipMemcpyAsync(h->d_kp_counts.p, h_counts, sizeof(int) * n_frames, hipMemcpyHostToDevice, h->stream));
    if (h->has_grid) {
        ORBX_TRY(h->d_cell_start.ensure((size_t)(ncell + 1) * n_frames));
        ORBX_TRY(h->d_cell_items.ensure((size_t)capacity * n_frames));
        ORBX_HIP(hipMemcpyAsync(h->d_cell_start.p, h->h_cell_start.p, sizeof(int32_t) * (size_t)(ncell + 1) * n_frames,
                                hipMemcpyHostToDevice, h->stream));
        ORBX_HIP(hipMemcpyAsync(h->d_cell_items.p, h->h_cell_items.p, sizeof(int32_t) * (size_t)capacity * n_frames,
                                hipMemcpyHostToDevice, h->stream));
    }
    ORBX_TRY(run_describe_stage(h, n_frames));
    if (descriptors)
        ORBX_HIP(hipMemcpyAsync(descriptors, h->d_desc.p, (size_t)capacity * n_frames * 32, hipMemcpyDeviceToHost,
                                h->stream));
    ORBX_HIP(hipStreamSynchronize(h->stream));
    collect_timings(h);
    h->last_counts.assign(h_counts, h_counts + n_frames);
    if (counts) std::memcpy(counts, h_counts, sizeof(int) * n_frames);
    if (keypoints) std::memcpy(keypoints, h->h_kp.p, sizeof(orbx_keypoint) * (size_t)capacity * n_frames);
    return ORBX_OK;
}

extern "C" {


orbx_status orbx_extract_batch(orbx_extractor* h, const uint8_t* d_images, int n_frames, size_t frame_stride,
                               int width, int height, int stride, int channels, orbx_keypoint* keypoints,
                               int capacity, int32_t* counts, uint8_t* descriptors) {
    ORBX_ENTER(h);
    ORBX_TRY(check_image_args(h, d_images, width, height, stride, channels));
    if (n_frames <= 0 || capacity <= 0 || frame_stride < (size_t)stride * height) return ORBX_BAD_ARG;
    if (n_frames > h->max_batch) return ORBX_CAPACITY;
    ORBX_HIP(hipSetDevice(h->device));
    ORBX_TRY(drain_pending(h));
    h->batch_dropped = false;  // this batch is now the most recent one
    h->last_counts.clear();
    h->cached_image.valid = false;
    ORBX_TRY(build_geometry(h, width, height, h->prm.level_pyramid));
    ORBX_TRY(ensure_buffers(h, n_frames));
    h->last_frames = n_frames;
    BlurForkGuard fork_guard{h};  // whichever way this call ends, the side stream is joined

    // Device-resident path: selection, undistortion, bucket grid and border filter run in one kernel per frame
    // (k_select_frame), so nothing crosses PCIe between detect and compute.  Its fast form keeps a frame's candidates
    // and keypoint slots in LDS; frames beyond that (more than kSelectMaxCands candidates, or a keypoint capacity
    // that does not fit beside them: 4K frames, thousands of keypoints) take the same kernel on a global workspace.
    // Only a frame with more candidates than the candidate list holds is left to the host path (which enlarges it).
    const int ncell_dev = h->has_grid ? h->grid.grid_cols * h->grid.grid_rows : 0;
    // the LDS form needs its selection elements and `capacity` keypoint slots in one CU's LDS (160 KB)
    const size_t select_lds_full = sizeof(stdsel::Elem) * (size_t)kSelectMaxCands +
                                   std::max((size_t)capacity * (sizeof(orbx_keypoint) + sizeof(int)),
                                            sizeof(uint16_t) * 2 * (size_t)kSelectMaxCands);
    // the grid (cell offsets + item lists) is built in the area of the selection elements, which are dead by then
    const size_t grid_ints = h->has_grid ? (size_t)ncell_dev + 1 + (size_t)capacity : 0;
    const bool lds_form = select_lds_full <= 160 * 1024 - 512 && grid_ints <= (size_t)kSelectMaxCands * 2;
    const bool device_path = h->device_select && h->geom.nlevels <= 16 && grid_ints <= (size_t)h->cand_cap * 2;
    // with the selection on the device the orientation is computed for the kept keypoints only, after it
    ORBX_TRY(run_detect_stage(h, d_images, n_frames, frame_stride, width, height, stride, channels, h->fork_blur,
                              device_path));
    if (h->distribute_mode && !device_path) {
        set_last_error_text("orbx_extract_batch: the distribution mode needs the selection on the device");
        return ORBX_UNSUPPORTED;
    }
    if (device_path) {
        h->kp_cap = capacity;
        ORBX_TRY(h->d_kp_src.ensure((size_t)capacity * n_frames));
        ORBX_TRY(h->d_kp.ensure((size_t)capacity * n_frames));
        ORBX_TRY(h->d_desc.ensure((size_t)capacity * n_frames * 32));
        ORBX_TRY(h->d_kp_counts.ensure(n_frames));
        ORBX_TRY(h->d_status.ensure(n_frames));
        if (h->has_grid) {
            ORBX_TRY(h->d_cell_start.ensure((size_t)(ncell_dev + 1) * n_frames));
            ORBX_TRY(h->d_cell_items.ensure((size_t)capacity * n_frames));
        }
        FrameModel fm;
        std::memset(&fm, 0, sizeof(fm));
        fm.has_camera = h->has_camera;
        fm.has_grid = h->has_grid;
        if (h->has_camera) fm.cam = h->camera;
        if (h->has_grid) fm.grid = h->grid;
        fm.width = width;
        fm.height = height;
        fm.edge = h->prm.edge_threshold;
        fm.patch_size = h->prm.patch_size;
        // LDS of one workgroup: the selection elements of up to max_cands candidates, then the keypoint / cell area
        // (which the stopper lists of the partitions alias).  A workgroup sized for kSelectMaxCands has a CU to
        // itself; most frames have far fewer candidates, so a first launch with half a CU's LDS per workgroup (two
        // frames per CU) takes every frame that fits it and marks the others (status 3) for a second, full-size launch.
        auto select_lds = [&](int max_cands) {
            return sizeof(stdsel::Elem) * (size_t)max_cands +
                   std::max((size_t)capacity * (sizeof(orbx_keypoint) + sizeof(int)), sizeof(uint16_t) * 2 * (size_t)max_cands);
        };
        // the global form's workspace: per slice the selection elements of cand_cap candidates, then the keypoint /
        // cell area or the (32-bit) stopper lists, whichever is larger.  With the LDS form in front only the rare
        // frame needs a slice (a few are kept; a batch with more such frames than slices goes to the host path).
        const int ws_slices = lds_form ? std::min(n_frames, 8) : n_frames;
        const size_t ws_stride = (sizeof(stdsel::Elem) * (size_t)h->cand_cap +
                                  std::max((size_t)capacity * (sizeof(orbx_keypoint) + sizeof(int)),
                                           sizeof(uint32_t) * 2 * (size_t)h->cand_cap) + 255) & ~(size_t)255;
        ORBX_TRY(h->d_select_ws.ensure(ws_stride * ws_slices + sizeof(int) * 64));
        int* ws_counter = (int*)(h->d_select_ws.p + ws_stride * ws_slices);
        ORBX_HIP(hipMemsetAsync(ws_counter, 0, sizeof(int), h->stream));
        h->sel_fm = fm;
        h->sel_ws_stride = ws_stride;
        h->sel_ws_slices = ws_slices;
        h->timer.begin(ST_SELECT, h->stream);
        const int* picked_idx = nullptr;
        const int* picked_n = nullptr;
        if (h->distribute_mode) {
            ORBX_TRY(launch_distribute(h, n_frames));
            picked_idx = h->d_picked_idx.p;
            picked_n = h->d_picked_n.p;
        }
        auto launch_global = [&](int pass) {
            hipLaunchKernelGGL(k_select_frame<true>, dim3(n_frames), dim3(64 * h->geom.nlevels), 0, h->stream,
                               h->d_cand.p, h->cand_cap, h->d_level_counts_p, h->d_totals.p, h->geom, fm, h->d_kp.p,
                               capacity, h->d_kp_counts.p, h->d_cell_start.p, h->d_cell_items.p, h->d_status.p,
                               h->cand_cap, pass, h->d_select_ws.p, ws_stride, ws_slices, ws_counter, h->d_kp_src.p,
                               picked_idx, picked_n, h->picked_stride);
        };
        if (lds_form) {
            const size_t shmem = select_lds(kSelectMaxCands);
            ORBX_TRY(ensure_dynamic_lds((const void*)k_select_frame<false>, shmem));
            const long half_cu = 80 * 1024 - 512;  // half of the CU's 160 KB, less the kernel's static LDS and rounding
            const long kp_area = (long)capacity * (long)(sizeof(orbx_keypoint) + sizeof(int));
            const int small_cands = (int)((half_cu - kp_area) / (long)sizeof(stdsel::Elem)) / 64 * 64;
            const bool two_tier = small_cands >= 2048 && small_cands < kSelectMaxCands && grid_ints <= (size_t)small_cands * 2;
            auto launch_select = [&](int max_cands, int pass) {
                hipLaunchKernelGGL(k_select_frame<false>, dim3(n_frames), dim3(64 * h->geom.nlevels),
                                   select_lds(max_cands), h->stream, h->d_cand.p, h->cand_cap, h->d_level_counts_p,
                                   h->d_totals.p, h->geom, fm, h->d_kp.p, capacity, h->d_kp_counts.p,
                                   h->d_cell_start.p, h->d_cell_items.p, h->d_status.p, max_cands, pass, nullptr, 0, 0,
                                   nullptr, h->d_kp_src.p, picked_idx, picked_n, h->picked_stride);
            };
            if (picked_idx) {
                // the chosen keypoints of a frame are few: one launch of the small form takes every frame
                launch_select(two_tier && small_cands >= capacity + 4 * h->geom.nlevels ? small_cands : kSelectMaxCands, -1);
            } else if (two_tier) {
                launch_select(small_cands, 0);
                launch_select(kSelectMaxCands, 1);
            } else {
                launch_select(kSelectMaxCands, -1);
            }
            if (!picked_idx) launch_global(2);  // frames the LDS form could not hold
        } else {
            launch_global(-1);
        }
        h->timer.end(ST_SELECT, h->stream);
        launch_angle_kept(h, n_frames);
        ORBX_HIP(hipGetLastError());
        ORBX_TRY(run_describe_stage(h, n_frames));
        ORBX_TRY(h->h_ints.ensure((size_t)n_frames * (2 * kMaxLevels + 2) + 1));
        int* h_status = h->h_ints.p;
        int* h_cnt = h->h_ints.p + n_frames;
        ORBX_HIP(hipMemcpyAsync(h_status, h->d_status.p, sizeof(int) * n_frames, hipMemcpyDeviceToHost, h->stream));
        ORBX_HIP(hipMemcpyAsync(h_cnt, h->d_kp_counts.p, sizeof(int) * n_frames, hipMemcpyDeviceToHost, h->stream));
        if (keypoints)
            ORBX_HIP(hipMemcpyAsync(keypoints, h->d_kp.p, sizeof(orbx_keypoint) * (size_t)capacity * n_frames,
                                    hipMemcpyDeviceToHost, h->stream));
        if (descriptors)
            ORBX_HIP(hipMemcpyAsync(descriptors, h->d_desc.p, (size_t)capacity * n_frames * 32, hipMemcpyDeviceToHost,
                                    h->stream));
        if (!keypoints && !descriptors && !counts) {
            // nothing to hand to the host: the call returns with the work queued; orbx_extractor_finish looks at the
            // status words (the device results may be consumed on the same stream before that)
            ORBX_HIP(hipEventRecord(h->ev_done, h->stream));
            h->pending.active = true;
            h->pending.d_images = d_images;
            h->pending.n_frames = n_frames;
            h->pending.frame_stride = frame_stride;
            h->pending.width = width;
            h->pending.height = height;
            h->pending.stride = stride;
            h->pending.channels = channels;
            h->pending.capacity = capacity;
            return ORBX_OK;
        }
        ORBX_HIP(hipStreamSynchronize(h->stream));
        collect_timings(h);
        bool redo = false;
        for (int f = 0; f < n_frames; f++) {
            if (h_status[f] == 1) return ORBX_CAPACITY;
            if (h_status[f] == 2) redo = true;
        }
        if (redo && h->distribute_mode) {
            set_last_error_text("orbx_extract_batch: a frame has more corners at the low threshold than the candidate "
                                "list holds; the distribution mode has no host path to enlarge it");
            return ORBX_UNSUPPORTED;
        }
        if (redo) {
            int left = 0;
            ORBX_TRY(more_global_passes(h, n_frames, capacity, &left));
            redo = left > 0;
            for (int f = 0; f < n_frames; f++)
                if (h_status[f] == 1) return ORBX_CAPACITY;
            if (!redo) {
                if (keypoints)
                    ORBX_HIP(hipMemcpy(keypoints, h->d_kp.p, sizeof(orbx_keypoint) * (size_t)capacity * n_frames,
                                       hipMemcpyDeviceToHost));
                if (descriptors)
                    ORBX_HIP(hipMemcpy(descriptors, h->d_desc.p, (size_t)capacity * n_frames * 32, hipMemcpyDeviceToHost));
            }
        }
        if (!redo) {
            h->last_counts.assign(h_cnt, h_cnt + n_frames);
            if (counts) std::memcpy(counts, h_cnt, sizeof(int) * n_frames);
            return ORBX_OK;
        }
    }
    return batch_host_path(h, n_frames, width, height, capacity, keypoints, counts, descriptors);
}

orbx_status orbx_extractor_finish(orbx_extractor* h, int32_t* counts, int* redone) {
    ORBX_ENTER(h);
    if (!h) return ORBX_BAD_ARG;
    if (redone) *redone = 0;
    if (!h->pending.active) {
        if (h->batch_dropped) {
            set_last_error_text("orbx_extractor_finish: the deferred batch was dropped by a later call on the handle "
                                "before it was finished; there is nothing to report");
            return ORBX_BAD_ARG;
        }
        // the most recent batch was completed inside orbx_extract_batch: only its counts are left to hand over
        if (counts && !h->last_counts.empty()) std::memcpy(counts, h->last_counts.data(), sizeof(int32_t) * h->last_counts.size());
        return ORBX_OK;
    }
    ORBX_HIP(hipSetDevice(h->device));
    const orbx_extractor::Pending a = h->pending;
    h->pending.active = false;
    ORBX_HIP(hipEventSynchronize(h->ev_done));
    collect_timings(h);
    const int* h_status = h->h_ints.p;
    const int* h_cnt = h->h_ints.p + a.n_frames;
    bool redo = false;
    for (int f = 0; f < a.n_frames; f++) {
        if (h_status[f] == 1) return ORBX_CAPACITY;
        if (h_status[f] == 2) redo = true;
    }
    if (!redo) {
        h->last_counts.assign(h_cnt, h_cnt + a.n_frames);
        if (counts) std::memcpy(counts, h_cnt, sizeof(int) * a.n_frames);
        return ORBX_OK;
    }
    if (h->distribute_mode) {
        set_last_error_text("orbx_extractor_finish: a frame has more corners at the low threshold than the candidate "
                            "list holds; the distribution mode has no host path to enlarge it");
        return ORBX_UNSUPPORTED;
    }
    // frames were left over (more of them beyond the LDS form than workspace slices, or a candidate list that
    // overflowed): the batch is finished by further device passes or, for an overflowed list, by the host path, and
    // whatever consumed the device results in the meantime has to run again
    if (redone) *redone = 1;
    BlurForkGuard fork_guard{h};
    int left = 0;
    ORBX_TRY(more_global_passes(h, a.n_frames, a.capacity, &left));
    for (int f = 0; f < a.n_frames; f++)
        if (h_status[f] == 1) return ORBX_CAPACITY;
    if (left == 0) {
        h->last_counts.assign(h_cnt, h_cnt + a.n_frames);
        if (counts) std::memcpy(counts, h_cnt, sizeof(int) * a.n_frames);
        return ORBX_OK;
    }
    return batch_host_path(h, a.n_frames, a.width, a.height, a.capacity, nullptr, counts, nullptr);
}

orbx_status orbx_extractor_set_blur_fork(orbx_extractor* h, int enabled) {
    ORBX_ENTER(h);
    if (!h) return ORBX_BAD_ARG;
    h->fork_blur = enabled != 0;
    return ORBX_OK;
}

orbx_status orbx_extractor_set_frame_model(orbx_extractor* h, const orbx_camera* camera, const orbx_grid* grid) {
    ORBX_ENTER(h);
    if (!h) return ORBX_BAD_ARG;
    if (grid && (grid->grid_cols <= 0 || grid->grid_rows <= 0 || grid->grid_size_x <= 0 || grid->grid_size_y <= 0 ||
                 (long long)grid->grid_cols * grid->grid_rows > (1 << 20)))
        return ORBX_BAD_ARG;
    if (camera && (!(camera->fx != 0.0) || !(camera->fy != 0.0))) return ORBX_BAD_ARG;
    h->has_camera = camera != nullptr;
    h->has_grid = grid != nullptr;
    if (camera) h->camera = *camera;
    if (grid) h->grid = *grid;
    return ORBX_OK;
}

orbx_status orbx_extractor_set_distribution(orbx_extractor* h, int mode, int low_threshold, int cell_size) {
    ORBX_ENTER(h);
    if (mode != 0 && mode != 1) return ORBX_BAD_ARG;
    if (mode == 1 && (low_threshold < 1 || low_threshold > 255 || low_threshold > std::max(h->prm.score_threshold, 1) ||
                      cell_size < 8 || cell_size > 4096))
        return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    ORBX_TRY(drain_pending(h));
    h->distribute_mode = mode;
    if (mode == 1) {
        h->distribute_low = low_threshold;
        h->distribute_cell = cell_size;
    }
    h->pyramid_valid = false;
    h->cached_image.valid = false;
    return ORBX_OK;
}

orbx_status orbx_extractor_set_pyramid_tail(orbx_extractor* h, int base_level, int rows_per_step) {
    ORBX_ENTER(h);
    if (base_level < 0 || base_level >= kMaxLevels || rows_per_step < 0 || rows_per_step > 64) return ORBX_BAD_ARG;
    const int rows = rows_per_step > 0 ? rows_per_step : h->tail_rows;
    if (base_level != h->tail_base || rows != h->tail_rows) {
        h->tail_base = base_level;
        h->tail_rows = rows;
        h->cur_w = h->cur_h = h->cur_levels = -1;   // plan again at the next call
    }
    return ORBX_OK;
}

orbx_status orbx_extractor_set_pyramid_kernel(orbx_extractor* h, int kind, int rows_per_step) {
    ORBX_ENTER(h);
    if (kind < 0 || kind > 2 || rows_per_step < 0 || rows_per_step > 64) return ORBX_BAD_ARG;
    h->pyramid_kernel = kind;
    if (rows_per_step > 0 && rows_per_step != h->stream_rows) {
        h->stream_rows = rows_per_step;
        h->cur_w = h->cur_h = h->cur_levels = -1;   // plan again at the next call
    }
    return ORBX_OK;
}

orbx_status orbx_extractor_set_image_cache(orbx_extractor* h, int enabled) {
    ORBX_ENTER(h);
    if (!h) return ORBX_BAD_ARG;
    h->image_cache_enabled = enabled != 0;
    h->cached_image.valid = false;
    return ORBX_OK;
}

orbx_status orbx_extractor_set_device_select(orbx_extractor* h, int enabled) {
    ORBX_ENTER(h);
    if (!h) return ORBX_BAD_ARG;
    h->device_select = enabled != 0;
    return ORBX_OK;
}

orbx_status orbx_extractor_device_grid(orbx_extractor* h, const int32_t** d_cell_start, const int32_t** d_cell_items) {
    ORBX_ENTER(h);
    if (!h || !h->has_grid) return ORBX_BAD_ARG;
    if (d_cell_start) *d_cell_start = h->d_cell_start.p;
    if (d_cell_items) *d_cell_items = h->d_cell_items.p;
    return ORBX_OK;
}

orbx_status orbx_extractor_device_results(orbx_extractor* h, const orbx_keypoint** d_keypoints,
                                          const uint8_t** d_descriptors, const int32_t** d_counts, int* kp_capacity) {
    ORBX_ENTER(h);
    if (!h) return ORBX_BAD_ARG;
    if (d_keypoints) *d_keypoints = h->d_kp.p;
    if (d_descriptors) *d_descriptors = h->d_desc.p;
    if (d_counts) *d_counts = h->d_kp_counts.p;
    if (kp_capacity) *kp_capacity = h->kp_cap;
    return ORBX_OK;
}

}  // extern "C"
