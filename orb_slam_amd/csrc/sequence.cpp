// Sequence input for the extraction path: the TUM association list of the reference's TUMRGBDCamera
// (reference src/geometry/rgbd_camera.cpp:225-252, include/orb_slam/geometry/rgbd_camera.h:204-235), a PNG decoder
// that returns what cv::imread returns for the two ways the reference calls it, and a feeder that keeps batches of
// frames moving into HBM while the previous batch is being processed (pinned staging, a copy stream of its own).
// Host code only; the decoder needs zlib.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <fstream>
#include <mutex>
#include <sstream>
#include <thread>

#include "common.h"

using namespace orbx;

// ---- association list ------------------------------------------------------------------------------------------
struct orbx_sequence {
    std::string dir;
    std::vector<double> time_stamps;
    std::vector<std::string> rgb_files, depth_files;
};

// ---- PNG -------------------------------------------------------------------------------------------------------
namespace {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct PngHeader {
    int width = 0, height = 0, bit_depth = 0, color_type = 0, interlace = 0;
    int src_channels() const { return color_type == 0 ? 1 : color_type == 2 ? 3 : color_type == 3 ? 1 : color_type == 4 ? 2 : 4; }
};

orbx_status png_header(const uint8_t* f, size_t size, PngHeader& h) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (size < 33 || std::memcmp(f, sig, 8) != 0) return ORBX_BAD_ARG;
    if (be32(f + 8) != 13 || std::memcmp(f + 12, "IHDR", 4) != 0) return ORBX_BAD_ARG;
    h.width = (int)be32(f + 16);
    h.height = (int)be32(f + 20);
    h.bit_depth = f[24];
    h.color_type = f[25];
    h.interlace = f[28];
    if (h.width <= 0 || h.height <= 0 || h.width > 65535 || h.height > 65535) return ORBX_BAD_ARG;
    if ((size_t)h.width * h.height > ((size_t)1 << 27)) return ORBX_UNSUPPORTED;  // 128 Mpixel: bounds every buffer below
    const bool type_ok = h.color_type == 0 || h.color_type == 2 || h.color_type == 3 || h.color_type == 4 || h.color_type == 6;
    if (!type_ok || f[26] != 0 || f[27] != 0) return ORBX_BAD_ARG;
    if (h.interlace != 0) return ORBX_UNSUPPORTED;
    if (h.bit_depth != 8 && !(h.bit_depth == 16 && h.color_type != 3)) return ORBX_UNSUPPORTED;
    return ORBX_OK;
}

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Inflates the IDAT stream and undoes the scanline filters; `raw` receives height rows of width * bpp bytes.
orbx_status png_scanlines(const uint8_t* f, size_t size, const PngHeader& h, std::vector<uint8_t>& raw,
                          uint8_t (*palette)[3], int* palette_n) {
    const int bpp = h.src_channels() * h.bit_depth / 8;
    const size_t row = (size_t)h.width * bpp;
    std::vector<uint8_t> filtered((row + 1) * h.height);
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return ORBX_DEVICE;
    zs.next_out = filtered.data();
    zs.avail_out = (uInt)filtered.size();
    size_t pos = 8;
    bool done = false, seen_end = false;
    *palette_n = 0;
    while (pos + 12 <= size && !seen_end) {
        const uint32_t len = be32(f + pos);
        const uint8_t* type = f + pos + 4;
        if (pos + 12 + (size_t)len > size) break;
        const uint8_t* data = f + pos + 8;
        if (std::memcmp(type, "IDAT", 4) == 0 && !done) {
            zs.next_in = const_cast<uint8_t*>(data);
            zs.avail_in = len;
            const int r = inflate(&zs, Z_NO_FLUSH);
            if (r == Z_STREAM_END) done = true;
            else if (r != Z_OK && r != Z_BUF_ERROR) { inflateEnd(&zs); return ORBX_BAD_ARG; }
        } else if (std::memcmp(type, "PLTE", 4) == 0) {
            *palette_n = std::min<int>(len / 3, 256);
            for (int i = 0; i < *palette_n; i++) std::memcpy(palette[i], data + 3 * i, 3);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            seen_end = true;
        }
        pos += 12 + (size_t)len;
    }
    const bool complete = zs.avail_out == 0;
    inflateEnd(&zs);
    if (!complete) return ORBX_BAD_ARG;
    raw.resize(row * h.height);
    for (int y = 0; y < h.height; y++) {
        const uint8_t* src = filtered.data() + (row + 1) * y;
        const int ft = src[0];
        src++;
        uint8_t* dst = raw.data() + row * y;
        const uint8_t* up = y ? dst - row : nullptr;
        switch (ft) {
        case 0: std::memcpy(dst, src, row); break;
        case 1:
            for (size_t i = 0; i < row; i++) dst[i] = (uint8_t)(src[i] + (i >= (size_t)bpp ? dst[i - bpp] : 0));
            break;
        case 2:
            for (size_t i = 0; i < row; i++) dst[i] = (uint8_t)(src[i] + (up ? up[i] : 0));
            break;
        case 3:
            for (size_t i = 0; i < row; i++)
                dst[i] = (uint8_t)(src[i] + (((i >= (size_t)bpp ? dst[i - bpp] : 0) + (up ? up[i] : 0)) >> 1));
            break;
        case 4:
            for (size_t i = 0; i < row; i++) {
                const int a = i >= (size_t)bpp ? dst[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
                dst[i] = (uint8_t)(src[i] + paeth(a, b, c));
            }
            break;
        default: return ORBX_BAD_ARG;
        }
    }
    return ORBX_OK;
}

// Output geometry of cv::imread for the two flag values the reference uses.
void png_output_format(const PngHeader& h, int mode, int* channels, int* bytes_per_sample) {
    if (mode == ORBX_IMREAD_COLOR) {
        *channels = 3;
        *bytes_per_sample = 1;
    } else {
        *channels = h.color_type == 0 ? 1 : (h.color_type == 2 || h.color_type == 3) ? 3 : h.color_type == 4 ? 2 : 4;
        // cv::imread(IMREAD_UNCHANGED) expands gray+alpha to four channels (gray replicated) and palettes to BGR
        if (h.color_type == 4) *channels = 4;
        *bytes_per_sample = h.bit_depth == 16 ? 2 : 1;
    }
}

orbx_status read_file(const char* path, std::vector<uint8_t>& out) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return ORBX_BAD_ARG;
    const std::streamsize n = f.tellg();
    if (n <= 0) return ORBX_BAD_ARG;
    out.resize((size_t)n);
    f.seekg(0);
    if (!f.read(reinterpret_cast<char*>(out.data()), n)) return ORBX_BAD_ARG;
    return ORBX_OK;
}

}  // namespace

extern "C" {

orbx_status orbx_png_decode(const uint8_t* file, size_t size, int mode, uint8_t* out, size_t out_capacity, int* width,
                            int* height, int* channels, int* bytes_per_sample) {
    if (!file || (mode != ORBX_IMREAD_UNCHANGED && mode != ORBX_IMREAD_COLOR)) return ORBX_BAD_ARG;
    PngHeader h;
    ORBX_TRY(png_header(file, size, h));
    int ch = 0, bps = 0;
    png_output_format(h, mode, &ch, &bps);
    if (width) *width = h.width;
    if (height) *height = h.height;
    if (channels) *channels = ch;
    if (bytes_per_sample) *bytes_per_sample = bps;
    if (!out) return ORBX_OK;
    const size_t need = (size_t)h.width * h.height * ch * bps;
    if (out_capacity < need) return ORBX_CAPACITY;
    std::vector<uint8_t> raw;
    uint8_t palette[256][3];
    int palette_n = 0;
    ORBX_TRY(png_scanlines(file, size, h, raw, palette, &palette_n));
    const int sc = h.src_channels(), sb = h.bit_depth / 8;
    const size_t npx = (size_t)h.width * h.height;
    // a source sample as cv::imread delivers it: 16-bit samples keep both bytes (host order) unless the output is
    // 8-bit, in which case libpng's strip_16 keeps the high byte
    auto sample8 = [&](size_t px, int c) -> uint8_t { return raw[(px * sc + c) * sb]; };
    auto sample16 = [&](size_t px, int c) -> uint16_t {
        const uint8_t* p = &raw[(px * sc + c) * 2];
        return (uint16_t)((p[0] << 8) | p[1]);
    };
    for (size_t px = 0; px < npx; px++) {
        // source colour as (r, g, b, a) channel indices, or a palette entry
        if (bps == 1) {
            uint8_t r, g, b, a = 255;
            if (h.color_type == 3) {
                const int i = raw[px];
                if (i >= palette_n) return ORBX_BAD_ARG;
                r = palette[i][0]; g = palette[i][1]; b = palette[i][2];
            } else if (h.color_type == 0 || h.color_type == 4) {
                r = g = b = sample8(px, 0);
                if (h.color_type == 4) a = sample8(px, 1);
            } else {
                r = sample8(px, 0); g = sample8(px, 1); b = sample8(px, 2);
                if (h.color_type == 6) a = sample8(px, 3);
            }
            uint8_t* d = out + px * ch;
            if (ch == 1) d[0] = r;
            else { d[0] = b; d[1] = g; d[2] = r; if (ch == 4) d[3] = a; }
        } else {
            uint16_t r, g, b, a = 65535;
            if (h.color_type == 0 || h.color_type == 4) {
                r = g = b = sample16(px, 0);
                if (h.color_type == 4) a = sample16(px, 1);
            } else {
                r = sample16(px, 0); g = sample16(px, 1); b = sample16(px, 2);
                if (h.color_type == 6) a = sample16(px, 3);
            }
            uint16_t* d = reinterpret_cast<uint16_t*>(out) + px * ch;
            if (ch == 1) d[0] = r;
            else { d[0] = b; d[1] = g; d[2] = r; if (ch == 4) d[3] = a; }
        }
    }
    return ORBX_OK;
}

orbx_status orbx_png_read(const char* path, int mode, uint8_t* out, size_t out_capacity, int* width, int* height,
                          int* channels, int* bytes_per_sample) {
    if (!path) return ORBX_BAD_ARG;
    std::vector<uint8_t> file;
    ORBX_TRY(read_file(path, file));
    return orbx_png_decode(file.data(), file.size(), mode, out, out_capacity, width, height, channels, bytes_per_sample);
}

orbx_status orbx_depth_to_float(const uint8_t* bgr8, size_t n_values, float depth_scale, float* out) {
    if (!bgr8 || !out) return ORBX_BAD_ARG;
    // cv::Mat::convertTo(CV_32F, alpha) from CV_8U: the product is formed in float
    for (size_t i = 0; i < n_values; i++) out[i] = (float)bgr8[i] * depth_scale + 0.f;
    return ORBX_OK;
}

orbx_status orbx_sequence_open(const char* dataset_dir, orbx_sequence** out) {
    if (!dataset_dir || !out) return ORBX_BAD_ARG;
    std::ifstream assoc_file((std::string(dataset_dir) + "assoc.txt").c_str());
    if (!assoc_file) return ORBX_BAD_ARG;
    orbx_sequence* q = new orbx_sequence();
    q->dir = dataset_dir;
    // the parsing of TUMRGBDCamera::setupCameraStream: `t rgb t depth` per non-empty line, the first stamp kept
    std::string s;
    while (std::getline(assoc_file, s)) {
        if (s.empty()) continue;
        std::stringstream ss;
        ss << s;
        double t;
        std::string rgb, depth;
        ss >> t;
        q->time_stamps.push_back(t);
        ss >> rgb;
        q->rgb_files.push_back(rgb);
        ss >> t;
        ss >> depth;
        q->depth_files.push_back(depth);
    }
    if (q->rgb_files.empty()) {  // the reference asserts on an empty list
        delete q;
        return ORBX_BAD_ARG;
    }
    *out = q;
    return ORBX_OK;
}

void orbx_sequence_close(orbx_sequence* q) { delete q; }

int orbx_sequence_length(const orbx_sequence* q) { return q ? (int)q->rgb_files.size() : 0; }

orbx_status orbx_sequence_entry(const orbx_sequence* q, int index, double* time_stamp, const char** rgb_file,
                                const char** depth_file) {
    if (!q || index < 0 || index >= (int)q->rgb_files.size()) return ORBX_BAD_ARG;
    if (time_stamp) *time_stamp = q->time_stamps[index];
    if (rgb_file) *rgb_file = q->rgb_files[index].c_str();
    if (depth_file) *depth_file = q->depth_files[index].c_str();
    return ORBX_OK;
}

}  // extern "C"

// ---- feeder ----------------------------------------------------------------------------------------------------
// A ring of slots, each a pinned host buffer and a device buffer for one batch of frames.  A producer thread fills
// the slots in order (PNG decode on a pool of workers, or nothing at all when the frames already sit in pinned host
// memory), queues the upload on the feeder's own stream and records the slot's event; orbx_feeder_next hands the
// oldest uploaded batch to the caller and recycles the one handed out before.  With three slots one batch is being
// processed, one is on the bus and one is being decoded.
struct orbx_feeder {
    int device = 0, width = 0, height = 0, channels = 0, batch = 0, n_slots = 0, n_threads = 1;
    size_t frame_bytes = 0;
    hipStream_t copy_stream = nullptr;
    struct Slot {
        uint8_t* pinned = nullptr;
        uint8_t* dev = nullptr;
        hipEvent_t uploaded = nullptr, consumed = nullptr;
        bool wait_consumed = false, was_uploaded = false;
        int frames = 0;
        int state = 0;  // 0 free, 1 uploaded (event recorded), 2 handed to the caller
    };
    std::vector<Slot> slots;
    std::mutex mu;
    std::condition_variable cv;
    std::thread producer;
    bool stop = false, finished = true;
    orbx_status error = ORBX_OK;
    long long next_out = 0, produced = 0;  // batch counters
    int held = -1;
    // the source of the running job
    const orbx_sequence* seq = nullptr;
    const uint8_t* mem = nullptr;
    size_t mem_stride = 0;
    bool mem_registered = false, mem_staged = false;
    void* mem_reg_base = nullptr;
    int first = 0, count = 0;
};

namespace {

void feeder_join(orbx_feeder* f) {
    {
        std::lock_guard<std::mutex> lk(f->mu);
        f->stop = true;
    }
    f->cv.notify_all();
    if (f->producer.joinable()) f->producer.join();
    (void)hipStreamSynchronize(f->copy_stream);
    if (f->mem_registered) {
        (void)hipHostUnregister(f->mem_reg_base);
        f->mem_registered = false;
    }
    f->stop = false;
    f->finished = true;
    f->held = -1;
    f->next_out = f->produced = 0;
    for (auto& s : f->slots) {
        s.state = 0;
        s.wait_consumed = false;
    }
}

// Decodes frames [b0, b0 + n) of the sequence into the slot's pinned buffer with the worker pool.
orbx_status decode_batch(orbx_feeder* f, orbx_feeder::Slot& slot, int b0, int n) {
    std::atomic<int> next(0);
    std::atomic<int> status((int)ORBX_OK);
    auto work = [&]() {
        std::vector<uint8_t> file;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n || status.load() != ORBX_OK) return;
            const std::string path = f->seq->dir + f->seq->rgb_files[b0 + i];
            int w = 0, h = 0, ch = 0, bps = 0;
            orbx_status st = read_file(path.c_str(), file);
            if (st == ORBX_OK)
                st = orbx_png_decode(file.data(), file.size(), ORBX_IMREAD_UNCHANGED, nullptr, 0, &w, &h, &ch, &bps);
            if (st == ORBX_OK && (w != f->width || h != f->height || ch != f->channels || bps != 1)) st = ORBX_BAD_ARG;
            if (st == ORBX_OK)
                st = orbx_png_decode(file.data(), file.size(), ORBX_IMREAD_UNCHANGED,
                                     slot.pinned + (size_t)i * f->frame_bytes, f->frame_bytes, nullptr, nullptr,
                                     nullptr, nullptr);
            if (st != ORBX_OK) status.store((int)st);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < f->n_threads; t++) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    return (orbx_status)status.load();
}

// Copies frames [b0, b0 + n) of a host-memory job into the slot's pinned buffer (used when the caller's memory could
// not be page-locked in place).
void stage_batch(orbx_feeder* f, orbx_feeder::Slot& slot, int b0, int n) {
    std::atomic<int> next(0);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            std::memcpy(slot.pinned + (size_t)i * f->frame_bytes, f->mem + (size_t)(b0 + i) * f->mem_stride, f->frame_bytes);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < f->n_threads; t++) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
}

void producer_main(orbx_feeder* f) {
    (void)hipSetDevice(f->device);
    const int n_batches = (f->count + f->batch - 1) / f->batch;
    for (int b = 0; b < n_batches; b++) {
        orbx_feeder::Slot& slot = f->slots[b % f->n_slots];
        {
            std::unique_lock<std::mutex> lk(f->mu);
            f->cv.wait(lk, [&] { return f->stop || slot.state == 0; });
            if (f->stop) return;
        }
        const int b0 = f->first + b * f->batch;
        const int n = std::min(f->batch, f->first + f->count - b0);
        orbx_status st = ORBX_OK;
        const uint8_t* src = slot.pinned;
        // the slot's last upload must have left the pinned buffer before it is refilled
        const bool staged = f->seq || f->mem_staged;
        if (staged && slot.was_uploaded && hipEventSynchronize(slot.uploaded) != hipSuccess) st = ORBX_DEVICE;
        if (st == ORBX_OK) {
            if (f->seq) st = decode_batch(f, slot, b0, n);
            else if (f->mem_staged) stage_batch(f, slot, b0, n);
            else src = f->mem + (size_t)b0 * f->mem_stride;
        }
        if (st == ORBX_OK) {
            hipError_t e = hipSuccess;
            // ... and the work the caller queued on the slot's device buffer must have finished before the copy
            // engine overwrites it
            if (slot.wait_consumed) e = hipStreamWaitEvent(f->copy_stream, slot.consumed, 0);
            slot.wait_consumed = false;
            if (e != hipSuccess) st = ORBX_DEVICE;
        }
        if (st == ORBX_OK) {
            hipError_t e;
            if (staged || f->mem_stride == f->frame_bytes)
                e = hipMemcpyAsync(slot.dev, src, (size_t)n * f->frame_bytes, hipMemcpyHostToDevice, f->copy_stream);
            else
                e = hipMemcpy2DAsync(slot.dev, f->frame_bytes, src, f->mem_stride, f->frame_bytes, n,
                                     hipMemcpyHostToDevice, f->copy_stream);
            if (e == hipSuccess) e = hipEventRecord(slot.uploaded, f->copy_stream);
            if (e != hipSuccess) st = ORBX_DEVICE;
        }
        {
            std::lock_guard<std::mutex> lk(f->mu);
            if (st != ORBX_OK) {
                f->error = st;
                f->finished = true;
                f->cv.notify_all();
                return;
            }
            slot.frames = n;
            slot.was_uploaded = true;
            slot.state = 1;
            f->produced = b + 1;
        }
        f->cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(f->mu);
        f->finished = true;
    }
    f->cv.notify_all();
}

orbx_status feeder_start(orbx_feeder* f, int first, int count) {
    f->first = first;
    f->count = count;
    f->error = ORBX_OK;
    f->finished = count <= 0;
    if (count > 0) f->producer = std::thread(producer_main, f);
    return ORBX_OK;
}

}  // namespace

extern "C" {

orbx_status orbx_feeder_create(int device, int width, int height, int channels, int batch, int n_slots, int n_threads,
                               orbx_feeder** out) {
    if (!out || width <= 0 || height <= 0 || channels <= 0 || channels > 4 || batch <= 0 || n_slots < 2 ||
        n_slots > 8 || n_threads < 1 || n_threads > 64)
        return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(device));
    orbx_feeder* f = new orbx_feeder();
    f->device = device;
    f->width = width;
    f->height = height;
    f->channels = channels;
    f->batch = batch;
    f->n_slots = n_slots;
    f->n_threads = n_threads;
    f->frame_bytes = (size_t)width * height * channels;
    f->slots.resize(n_slots);
    bool ok = hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking) == hipSuccess;
    for (auto& s : f->slots) {
        ok = ok && hipMalloc((void**)&s.dev, f->frame_bytes * batch) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        orbx_feeder_destroy(f);
        return ORBX_DEVICE;
    }
    *out = f;
    return ORBX_OK;
}

void orbx_feeder_destroy(orbx_feeder* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->copy_stream) feeder_join(f);
    for (auto& s : f->slots) {
        if (s.pinned) (void)hipHostFree(s.pinned);
        if (s.dev) (void)hipFree(s.dev);
        if (s.uploaded) (void)hipEventDestroy(s.uploaded);
        if (s.consumed) (void)hipEventDestroy(s.consumed);
    }
    if (f->copy_stream) (void)hipStreamDestroy(f->copy_stream);
    delete f;
}

orbx_status orbx_feeder_start_sequence(orbx_feeder* f, const orbx_sequence* seq, int first, int count) {
    if (!f || !seq || first < 0 || count < 0 || first + count > (int)seq->rgb_files.size()) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(f->device));
    feeder_join(f);
    for (auto& s : f->slots)
        if (!s.pinned) ORBX_HIP(hipHostMalloc((void**)&s.pinned, f->frame_bytes * f->batch, hipHostMallocDefault));
    f->seq = seq;
    f->mem = nullptr;
    return feeder_start(f, first, count);
}

orbx_status orbx_feeder_start_memory(orbx_feeder* f, const uint8_t* frames, size_t frame_stride, int count,
                                     int already_pinned) {
    if (!f || count < 0 || (count > 0 && !frames) || frame_stride < f->frame_bytes) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(f->device));
    feeder_join(f);
    f->seq = nullptr;
    f->mem = frames;
    f->mem_stride = frame_stride;
    f->mem_staged = false;
    if (count > 0 && !already_pinned) {
        // page-lock the enclosing pages in place; where the runtime refuses, stage through the pinned slots
        const uintptr_t lo = (uintptr_t)frames & ~(uintptr_t)4095;
        const uintptr_t hi = ((uintptr_t)frames + frame_stride * (size_t)count + 4095) & ~(uintptr_t)4095;
        if (hipHostRegister((void*)lo, hi - lo, hipHostRegisterDefault) == hipSuccess) {
            f->mem_registered = true;
            f->mem_reg_base = (void*)lo;
        } else {
            (void)hipGetLastError();
            f->mem_staged = true;
            for (auto& s : f->slots)
                if (!s.pinned) ORBX_HIP(hipHostMalloc((void**)&s.pinned, f->frame_bytes * f->batch, hipHostMallocDefault));
        }
    }
    return feeder_start(f, 0, count);
}

orbx_status orbx_feeder_next(orbx_feeder* f, void* consumer_stream, const uint8_t** d_frames, int* n_frames) {
    if (!f || !d_frames || !n_frames) return ORBX_BAD_ARG;
    *d_frames = nullptr;
    *n_frames = 0;
    int idx = -1;
    ORBX_HIP(hipSetDevice(f->device));
    {
        std::unique_lock<std::mutex> lk(f->mu);
        if (f->held >= 0) {
            // the batch handed out last time is released: everything queued on the caller's stream so far must
            // finish before the slot is overwritten (without a stream the caller has synchronised by itself)
            orbx_feeder::Slot& old = f->slots[f->held];
            if (consumer_stream) {
                ORBX_HIP(hipEventRecord(old.consumed, (hipStream_t)consumer_stream));
                old.wait_consumed = true;
            }
            old.state = 0;
            f->held = -1;
            f->cv.notify_all();
        }
        f->cv.wait(lk, [&] { return f->produced > f->next_out || f->finished || f->error != ORBX_OK; });
        if (f->produced > f->next_out) {
            idx = (int)(f->next_out % f->n_slots);
            f->slots[idx].state = 2;
            f->held = idx;
            f->next_out++;
        } else if (f->error != ORBX_OK) {
            return f->error;
        } else {
            return ORBX_OK;  // end of the job: no frames
        }
    }
    // the caller's stream waits for the upload; nothing blocks on the host
    if (consumer_stream) ORBX_HIP(hipStreamWaitEvent((hipStream_t)consumer_stream, f->slots[idx].uploaded, 0));
    else ORBX_HIP(hipEventSynchronize(f->slots[idx].uploaded));
    *d_frames = f->slots[idx].dev;
    *n_frames = f->slots[idx].frames;
    return ORBX_OK;
}

}  // extern "C"
