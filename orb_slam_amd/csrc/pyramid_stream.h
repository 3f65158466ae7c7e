// The whole pyramid of a frame in ONE pass over the image: a workgroup walks its frame top to bottom in steps of a few
// level-0 rows; every level keeps its most recent rows in an LDS ring, and in the same step each level l >= 1 emits
// the rows whose two source rows of level l-1 earlier steps have completed (cv::resize INTER_LINEAR_EXACT, the same
// arithmetic as the level-by-level kernels).  The image is read once and every pyramid pixel is written once; nothing
// is read back from HBM between levels, and a batch takes one launch instead of one per level.
//
// Which rows each level emits in which step, how deep each ring has to be, and every per-column / per-row
// coefficient are worked out on the host (build_stream_plan), which also replays the schedule against the rings
// (verify_stream_plan) before a kernel ever runs it: the device code only executes tables.  The host part is plain
// C++; emulate_stream_pyramid runs the same tables on the CPU, for tests on a machine without a GPU.
#pragma once

#include <cstdint>
#include <vector>

#include "geometry_host.h"

namespace orbx {

struct StreamCol {        // one aligned dword (four pixels) of a level's padded row, levels >= 1
    unsigned sel;         // byte position of each pixel's left sample within the eight bytes of source dwords d0, d1
    unsigned stepmask;    // 0xff per pixel whose right sample is the next source pixel (else the left one again)
    unsigned wgt[4];      // (256 - w1) | w1 << 16 per pixel: Q8 weights of the left and the right sample
    int j0;               // ring column of source dword d0 (d1, d2 follow)
    unsigned ok;          // bit t: pixel t lies inside the level's padded rectangle
};
static_assert(sizeof(StreamCol) == 32, "StreamCol layout");

struct StreamRow {        // one interior row of a level
    unsigned slots;       // bits 0-7 / 8-15: ring slots of the two source rows (levels >= 1); bits 16-23: this row's own slot
    int w1;               // Q8 weight of the lower source row
    int mirror;           // atlas row of the border row that repeats this row (reflect-101), or -1
    int pad;
};

struct StreamLevel {
    int w, h;             // interior size
    int ax, ay;           // interior origin in the atlas
    int ax_begin;         // atlas x of the first column of the padded row, aligned (16 for level 0, 4 above)
    int n_dwords;         // aligned dwords per padded row
    int pitch, depth;     // ring: dwords per row, rows (depth 0: no level above reads this one)
    int lds_off;          // ring offset in dwords
    int col_off, row_off; // offsets into the column / row tables
    unsigned magic;       // ceil(2^20 / n_dwords): item -> (row, column) without a division
};

struct StreamParams {
    int nlevels, n_steps, border, aw;
    // level 0 (gray conversion): 16-pixel groups as in k_gray_level0
    int nq, q_in0, q_in1, n_in, n_edge, left_edge;
    unsigned magic_in, magic_edge;
    int total_cols;       // columns of all levels >= 1 together (each thread of the workgroup owns up to kStreamColsPerThread)
    int rc_rows;          // most rows any level emits in one step
    int ring_dwords;      // LDS dwords of the rings; the per-step row cache follows them
    StreamLevel L[kMaxLevels];
};

constexpr int kStreamThreads = 256;        // threads of the one-pass kernel's workgroup
constexpr int kStreamColsPerThread = 3;    // columns of levels >= 1 a thread may own
constexpr int kStreamGrayItems = 2;        // 16-pixel groups (and border pixels) of level 0 a thread converts per step

struct StreamPlan {
    bool ok = false;
    StreamParams prm;
    std::vector<StreamCol> cols;
    std::vector<StreamRow> rows;
    std::vector<int16_t> sched;   // [step][level][2]: first row and one past the last row the level emits in the step
    size_t lds_bytes = 0;
};

namespace streamdetail {
inline unsigned magic_for(int n) { return n > 0 ? (unsigned)(((1u << 20) + n - 1) / n) : 0u; }
inline int mirror_row(int y, int h, int ay, int border) {
    if (y >= 1 && y <= border) return ay - y;
    if (y >= h - 1 - border && y <= h - 2) return ay + 2 * (h - 1) - y;
    return -1;
}
}  // namespace streamdetail

// Replays the schedule against the rings: every source row a step reads must be in its slot, and no row written in a
// step may land on a slot the same step still reads.  Also checks every table index the kernel will use.
inline bool verify_stream_plan(const StreamPlan& sp) {
    const StreamParams& P = sp.prm;
    const int nl = P.nlevels;
    if (nl < 1 || nl > kMaxLevels || P.n_steps < 1) return false;
    std::vector<std::vector<int>> held(nl);   // held[l][slot] = row of level l in that ring slot, -1 = none
    for (int l = 0; l < nl; l++) held[l].assign(std::max(P.L[l].depth, 1), -1);
    std::vector<int> done(nl, 0);
    for (int s = 0; s < P.n_steps; s++) {
        // reads of this step, against the state before the step
        for (int l = 1; l < nl; l++) {
            const int y0 = sp.sched[(size_t)(s * nl + l) * 2], y1 = sp.sched[(size_t)(s * nl + l) * 2 + 1];
            if (y0 != done[l] || y1 < y0 || y1 > P.L[l].h) return false;
            for (int y = y0; y < y1; y++) {
                const StreamRow& r = sp.rows[P.L[l].row_off + y];
                const int s0 = r.slots & 255, s1 = (r.slots >> 8) & 255;
                if (s0 >= P.L[l - 1].depth || s1 >= P.L[l - 1].depth) return false;
                // the rows the coefficients name: recover them from what the slots hold; both must be completed rows
                if (held[l - 1][s0] < 0 || held[l - 1][s1] < 0) return false;
                if (held[l - 1][s0] >= done[l - 1] || held[l - 1][s1] >= done[l - 1]) return false;
                if (held[l - 1][s1] != held[l - 1][s0] && held[l - 1][s1] != held[l - 1][s0] + 1) return false;
            }
        }
        // writes of this step: must not overwrite a slot that a read of this step (checked above) uses
        std::vector<std::vector<char>> read_now(nl);
        for (int l = 0; l < nl; l++) read_now[l].assign(held[l].size(), 0);
        for (int l = 1; l < nl; l++)
            for (int y = sp.sched[(size_t)(s * nl + l) * 2]; y < sp.sched[(size_t)(s * nl + l) * 2 + 1]; y++) {
                const StreamRow& r = sp.rows[P.L[l].row_off + y];
                read_now[l - 1][r.slots & 255] = 1;
                read_now[l - 1][(r.slots >> 8) & 255] = 1;
            }
        for (int l = 0; l < nl; l++) {
            const int y0 = sp.sched[(size_t)(s * nl + l) * 2], y1 = sp.sched[(size_t)(s * nl + l) * 2 + 1];
            if (l == 0 && (y0 != done[0] || y1 < y0 || y1 > P.L[0].h)) return false;
            for (int y = y0; y < y1; y++) {
                if (P.L[l].depth == 0) continue;
                const int slot = (sp.rows[P.L[l].row_off + y].slots >> 16) & 255;
                if (slot >= P.L[l].depth || read_now[l][slot]) return false;
                held[l][slot] = y;
            }
            done[l] = y1;
        }
    }
    for (int l = 0; l < nl; l++)
        if (done[l] != P.L[l].h) return false;
    // what a slot holds must be what the coefficients meant: replay once more comparing against the source rows
    // recorded when the plan was built is done by the builder (it derives the slots from the rows); here the column
    // tables: every source dword a lane reads lies inside the ring row
    for (int l = 1; l < nl; l++)
        for (int j = 0; j < P.L[l].n_dwords; j++) {
            const StreamCol& c = sp.cols[P.L[l].col_off + j];
            if (!c.ok) continue;
            if (c.j0 < 0 || c.j0 + 2 >= P.L[l - 1].pitch) return false;
            for (int t = 0; t < 4; t++)
                if (((c.sel >> (8 * t)) & 255u) > 7u) return false;
        }
    size_t end = 0;
    for (int l = 0; l < nl; l++) {
        if (P.L[l].depth > 0 && (size_t)P.L[l].lds_off != end) return false;
        end += (size_t)P.L[l].depth * P.L[l].pitch;
        if (P.L[l].pitch < P.L[l].n_dwords + 2) return false;
    }
    return ((end + 3) & ~(size_t)3) == (size_t)P.ring_dwords;
}

// Builds the plan for one geometry.  rows_per_step: level-0 rows a step converts.  Returns false (sp.ok stays false)
// when the geometry does not suit the streaming form: a level lower than twice the border, a resize whose four-pixel
// source span exceeds two dwords, rings beyond max_lds bytes.
//
// base > 0 plans the TAIL of the pyramid only: plan level 0 is pyramid level `base`, which an earlier kernel has
// already written to the atlas; the pass reads its interior rows from there (as a one-channel image of that size),
// keeps them in the ring and emits the levels above it.  w, hgt are then the size of that level.
inline bool build_stream_plan(const HostGeometry& hg, int w, int hgt, int rows_per_step, size_t max_lds, StreamPlan& sp,
                              int base = 0) {
    using namespace streamdetail;
    sp = StreamPlan();
    const int B = hg.g.border;
    if (base < 0 || base >= hg.g.nlevels) return false;
    const int nl = hg.g.nlevels - base;
    // the plan's own numbering: level l of the plan is pyramid level base + l
    struct GeomView {
        const LevelInfo* L;
        int atlas_w;
    } g{hg.g.L + base, hg.g.atlas_w};
    const ResizePlan* rplan = hg.rplan + base;
    if (nl < 1 || rows_per_step < 1 || g.L[0].w != w || g.L[0].h != hgt) return false;
    if (base > 0 && nl < 2) return false;
    for (int l = 0; l < nl; l++)
        if (g.L[l].h < 2 * B + 2 || g.L[l].w < 8) return false;
    StreamParams& P = sp.prm;
    std::memset(&P, 0, sizeof(P));
    P.nlevels = nl;
    P.border = B;
    P.aw = g.atlas_w;
    // level 0: the 16-pixel groups of k_gray_level0
    {
        const int rx = g.L[0].x;
        const int ax_begin = (rx - B) & ~15;
        const int x_last = rx + w + B;
        P.nq = (x_last - ax_begin + 15) >> 4;
        if (base == 0) {
            P.q_in0 = (rx - ax_begin + 15) >> 4;
            P.q_in1 = (rx + w - ax_begin) >> 4;
            P.n_in = std::max(P.q_in1 - P.q_in0, 0);
            P.n_edge = (P.nq - P.n_in) * 16;
            P.left_edge = P.q_in0 * 16;
        } else {
            // the tail form loads whole groups from the atlas, border bytes included: every group that holds an
            // interior pixel (the levels above sample interior pixels only), no single pixels
            P.q_in0 = (rx - ax_begin) >> 4;
            P.q_in1 = (rx + w - ax_begin + 15) >> 4;
            P.n_in = P.q_in1 - P.q_in0;
            P.n_edge = 0;
            P.left_edge = 0;
            if (ax_begin < 0 || ax_begin + 16 * P.q_in1 > g.atlas_w) return false;
        }
        P.magic_in = magic_for(P.n_in);
        P.magic_edge = magic_for(P.n_edge);
        if (P.n_in == 0) return false;
        StreamLevel& L = P.L[0];
        L.w = w; L.h = hgt; L.ax = rx; L.ay = g.L[0].y;
        L.ax_begin = ax_begin;
        L.n_dwords = P.nq * 4;
        L.pitch = L.n_dwords + 4;
        L.magic = magic_for(L.n_dwords);
    }
    // source rows of every interior row, columns of every level
    std::vector<std::vector<int>> r0(nl), r1(nl), wy(nl);
    for (int l = 1; l < nl; l++) {
        const LevelInfo& D = g.L[l];
        const LevelInfo& S = g.L[l - 1];
        const ResizePlan& rp = rplan[l];
        StreamLevel& L = P.L[l];
        L.w = D.w; L.h = D.h; L.ax = D.x; L.ay = D.y;
        L.ax_begin = rp.ax_begin;
        L.n_dwords = rp.n_dwords;
        L.pitch = L.n_dwords + 2;
        L.magic = magic_for(L.n_dwords);
        L.col_off = (int)sp.cols.size();
        for (int j = 0; j < rp.n_dwords; j++) {
            StreamCol c;
            std::memset(&c, 0, sizeof(c));
            int sx[4], lo = 0x7fffffff, hi = -1;
            bool okp[4];
            for (int t = 0; t < 4; t++) {
                const CoefX& k = hg.coefx[rp.cx_off + 4 * j + t];
                sx[t] = k.src_ax;
                const unsigned f = k.w1f;
                const unsigned w1 = f & 0x1ffu;
                c.wgt[t] = (256u - w1) | (w1 << 16);
                if ((f >> 14) & 1u) c.stepmask |= 0xffu << (8 * t);
                okp[t] = (f >> 15) != 0;
                if (okp[t]) { lo = std::min(lo, sx[t]); hi = std::max(hi, sx[t]); c.ok |= 1u << t; }
            }
            if (c.ok) {
                const int base = lo & ~3;
                if (hi - base >= 8) return false;   // the four left samples must lie within two source dwords
                for (int t = 0; t < 4; t++) c.sel |= (unsigned)(okp[t] ? sx[t] - base : 0) << (8 * t);
                if (base < P.L[l - 1].ax_begin) return false;
                c.j0 = (base - P.L[l - 1].ax_begin) >> 2;
                // the samples are interior pixels of the source level
                if (lo < S.x || hi > S.x + S.w - 1) return false;
            }
            sp.cols.push_back(c);
        }
        r0[l].resize(D.h); r1[l].resize(D.h); wy[l].resize(D.h);
        for (int y = 0; y < D.h; y++) {
            const CoefY& cy = hg.coefy[rp.cy_off + y + B];
            r0[l][y] = cy.row0 - S.y;
            r1[l][y] = cy.row1 - S.y;
            wy[l][y] = cy.w1;
            if (r0[l][y] < 0 || r1[l][y] >= S.h || (r1[l][y] != r0[l][y] && r1[l][y] != r0[l][y] + 1)) return false;
            if (y > 0 && r0[l][y] < r0[l][y - 1]) return false;
        }
    }
    // the schedule, with rings as deep as needed: level 0 takes rows_per_step rows a step; level l emits, up to a
    // cap, every row whose lower source row was completed before the step
    std::vector<int> cap(nl), need(nl, 0), done(nl, 0);
    for (int l = 0; l < nl; l++)
        cap[l] = l == 0 ? rows_per_step : (int)((long long)rows_per_step * g.L[l].h / hgt) + 2;
    // the kernel turns an item number into (row, column) by a multiply and a shift, exact while items * columns < 2^20
    for (int l = 0; l < nl; l++) {
        const long long n = l == 0 ? std::max(P.n_in, P.n_edge) : P.L[l].n_dwords;
        if ((long long)cap[l] * n * n >= (1ll << 20)) return false;
    }
    // every thread owns a fixed set of columns and a fixed number of level-0 items per step
    if ((int)sp.cols.size() > kStreamThreads * kStreamColsPerThread) return false;
    if (cap[0] * P.n_in > kStreamThreads * kStreamGrayItems || cap[0] * P.n_edge > kStreamThreads * kStreamGrayItems)
        return false;
    P.total_cols = (int)sp.cols.size();
    P.rc_rows = 0;
    for (int l = 0; l < nl; l++) P.rc_rows = std::max(P.rc_rows, cap[l]);
    if (nl * P.rc_rows > kStreamThreads) return false;   // one thread fetches one row record per step
    int steps = 0;
    const int max_steps = (hgt + rows_per_step - 1) / rows_per_step + 4 * nl + 8;
    while (done[nl - 1] < g.L[nl - 1].h || done[0] < hgt) {
        if (++steps > max_steps) return false;
        std::vector<int> y1(nl);
        y1[0] = std::min(done[0] + cap[0], hgt);
        for (int l = 1; l < nl; l++) {
            int y = done[l];
            while (y < g.L[l].h && y - done[l] < cap[l] && r1[l][y] < done[l - 1]) y++;
            y1[l] = y;
        }
        for (int l = 0; l < nl; l++) {
            sp.sched.push_back((int16_t)done[l]);
            sp.sched.push_back((int16_t)y1[l]);
            // rows of level l that must stay in the ring while this step writes up to row y1[l] - 1: from the oldest
            // row level l + 1 reads in this step (or will read next, if it emits nothing now)
            if (l + 1 < nl && done[l + 1] < g.L[l + 1].h) need[l] = std::max(need[l], y1[l] - r0[l + 1][done[l + 1]]);
        }
        done = y1;
    }
    P.n_steps = steps;
    size_t off = 0;
    for (int l = 0; l < nl; l++) {
        StreamLevel& L = P.L[l];
        L.depth = l + 1 < nl ? need[l] + 1 : 0;
        if (L.depth > 255) return false;
        L.lds_off = (int)off;
        off += (size_t)L.depth * L.pitch;
    }
    if (off == 0) return false;
    off = (off + 3) & ~(size_t)3;   // the row cache is read as 16-byte records
    P.ring_dwords = (int)off;
    // behind the rings: two buffers of (rows each level emits in a step, one 16-byte record per row) and of the
    // levels' row ranges
    sp.lds_bytes = off * 4 + 2 * ((size_t)nl * P.rc_rows * sizeof(StreamRow) + (size_t)kMaxLevels * 8);
    if (sp.lds_bytes > max_lds) return false;
    for (int l = 0; l < nl; l++) {
        StreamLevel& L = P.L[l];
        L.row_off = (int)sp.rows.size();
        for (int y = 0; y < L.h; y++) {
            StreamRow r;
            r.slots = L.depth > 0 ? (unsigned)(y % L.depth) << 16 : 0u;
            r.w1 = 0;
            if (l > 0) {
                const int d = P.L[l - 1].depth;
                r.slots |= (unsigned)(r0[l][y] % d) | ((unsigned)(r1[l][y] % d) << 8);
                r.w1 = wy[l][y];
            }
            r.mirror = mirror_row(y, L.h, L.ay, B);
            r.pad = 0;
            sp.rows.push_back(r);
        }
    }
    if (!verify_stream_plan(sp)) return false;
    // the slots hold exactly the rows the coefficients name (verify_stream_plan checks presence and adjacency; this
    // pins the identity): replay, tracking the row in every slot
    {
        std::vector<std::vector<int>> held(nl);
        for (int l = 0; l < nl; l++) held[l].assign(std::max(P.L[l].depth, 1), -1);
        for (int s = 0; s < P.n_steps; s++) {
            for (int l = 1; l < nl; l++)
                for (int y = sp.sched[(size_t)(s * nl + l) * 2]; y < sp.sched[(size_t)(s * nl + l) * 2 + 1]; y++) {
                    const StreamRow& r = sp.rows[P.L[l].row_off + y];
                    if (held[l - 1][r.slots & 255] != r0[l][y] || held[l - 1][(r.slots >> 8) & 255] != r1[l][y]) return false;
                }
            for (int l = 0; l < nl; l++)
                if (P.L[l].depth > 0)
                    for (int y = sp.sched[(size_t)(s * nl + l) * 2]; y < sp.sched[(size_t)(s * nl + l) * 2 + 1]; y++)
                        held[l][(sp.rows[P.L[l].row_off + y].slots >> 16) & 255] = y;
        }
    }
    sp.ok = true;
    return true;
}

// The plan executed on the CPU: the same tables, the same order of steps, scalar arithmetic.  image: 8-bit, 1 or 3
// channels (BGR); atlas: atlas_w x atlas_h bytes, only the levels' padded rectangles are written.
// Returns false if any access the tables lead to leaves the atlas (atlas_h rows) or the rings.
// tail: the plan starts at a level that is already in the atlas (image = that level's interior, ch = 1, stride =
// atlas width): its rows are only loaded into the ring, not written again.
inline bool emulate_stream_pyramid(const StreamPlan& sp, const uint8_t* image, int stride, int ch, uint8_t* atlas,
                                   int atlas_h, bool tail = false) {
    const StreamParams& P = sp.prm;
    const int nl = P.nlevels, aw = P.aw, B = P.border;
    std::vector<unsigned> lds((size_t)sp.prm.ring_dwords, 0xdeadbeefu);
    bool inside = true;
    auto put = [&](int row, int ax, uint8_t v) {
        if (row < 0 || row >= atlas_h || ax < 0 || ax >= aw) { inside = false; return; }
        atlas[(size_t)row * aw + ax] = v;
    };
    auto reflect = [](int p, int len) {
        if (len == 1) return 0;
        while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * (len - 1) - p;
        return p;
    };
    auto gray = [](unsigned b, unsigned g, unsigned r) { return (b * 1868u + g * 9617u + r * 4899u + (1u << 13)) >> 14; };
    for (int s = 0; s < P.n_steps; s++) {
        std::vector<unsigned> next = lds;   // writes of a step become visible at its end (the kernel's barrier)
        {
            const StreamLevel& L = P.L[0];
            for (int y = sp.sched[(size_t)(s * nl) * 2]; y < sp.sched[(size_t)(s * nl) * 2 + 1]; y++) {
                const StreamRow& r = sp.rows[L.row_off + y];
                auto ring_put = [&](int ax, uint8_t v) {
                    const size_t byte = ((size_t)L.lds_off + (size_t)((r.slots >> 16) & 255) * L.pitch) * 4 + (ax - L.ax_begin);
                    if (ax < L.ax_begin || (byte >> 2) >= next.size()) { inside = false; return; }
                    unsigned& word = next[byte >> 2];
                    word = (word & ~(0xffu << (8 * (byte & 3)))) | ((unsigned)v << (8 * (byte & 3)));
                };
                if (tail) {
                    // whole 16-byte groups of the level's row as the atlas holds it
                    for (int ax = L.ax_begin + 16 * P.q_in0; ax < L.ax_begin + 16 * P.q_in1; ax++) {
                        const int row = L.ay + y;
                        if (row < 0 || row >= atlas_h || ax < 0 || ax >= aw) { inside = false; continue; }
                        if (L.depth > 0) ring_put(ax, atlas[(size_t)row * aw + ax]);
                    }
                    continue;
                }
                const uint8_t* src = image + (size_t)y * stride;
                for (int ax = L.ax - B; ax < L.ax + L.w + B; ax++) {
                    const uint8_t* px = src + (size_t)reflect(ax - L.ax, L.w) * ch;
                    const uint8_t v = (uint8_t)(ch == 1 ? px[0] : gray(px[0], px[1], px[2]));
                    put(L.ay + y, ax, v);
                    if (r.mirror >= 0) put(r.mirror, ax, v);
                    if (L.depth > 0) ring_put(ax, v);
                }
            }
        }
        for (int l = 1; l < nl; l++) {
            const StreamLevel& L = P.L[l];
            const StreamLevel& S = P.L[l - 1];
            for (int y = sp.sched[(size_t)(s * nl + l) * 2]; y < sp.sched[(size_t)(s * nl + l) * 2 + 1]; y++) {
                const StreamRow& r = sp.rows[L.row_off + y];
                const size_t offA = (size_t)S.lds_off + (size_t)(r.slots & 255) * S.pitch;
                const size_t offB = (size_t)S.lds_off + (size_t)((r.slots >> 8) & 255) * S.pitch;
                const unsigned* rowA = &lds[offA];
                const unsigned* rowB = &lds[offB];
                for (int j = 0; j < L.n_dwords; j++) {
                    const StreamCol& c = sp.cols[L.col_off + j];
                    if (!c.ok) continue;
                    if (c.j0 < 0 || std::max(offA, offB) + c.j0 + 3 > lds.size()) { inside = false; continue; }
                    uint8_t a8[12], b8[12];
                    std::memcpy(a8, rowA + c.j0, 12);
                    std::memcpy(b8, rowB + c.j0, 12);
                    unsigned out = 0;
                    for (int t = 0; t < 4; t++) {
                        const int k = (c.sel >> (8 * t)) & 255, k2 = k + (((c.stepmask >> (8 * t)) & 1u) ? 1 : 0);
                        const unsigned wx0 = c.wgt[t] & 0xffffu, wx1 = c.wgt[t] >> 16;
                        const unsigned hA = wx0 * a8[k] + wx1 * a8[k2], hB = wx0 * b8[k] + wx1 * b8[k2];
                        const unsigned v = (256u - (unsigned)r.w1) * hA + (unsigned)r.w1 * hB;
                        out |= ((v + (1u << 15)) >> 16) << (8 * t);
                    }
                    for (int t = 0; t < 4; t++)
                        if (c.ok & (1u << t)) {
                            const int ax = L.ax_begin + 4 * j + t;
                            put(L.ay + y, ax, (uint8_t)(out >> (8 * t)));
                            if (r.mirror >= 0) put(r.mirror, ax, (uint8_t)(out >> (8 * t)));
                        }
                    if (L.depth > 0) {
                        const size_t at = (size_t)L.lds_off + (size_t)((r.slots >> 16) & 255) * L.pitch + j;
                        if (at >= next.size()) { inside = false; continue; }
                        next[at] = out;
                    }
                }
            }
        }
        lds.swap(next);
    }
    return inside;
}

}  // namespace orbx
