// timg_amd/csrc/block_canvas.hip -- device twin of the pixel work in
// timg::UnicodeBlockCanvas::Send (src/unicode-block-canvas.cc:323-403).
//
// The reference walks cells left to right carrying state (last emitted
// foreground, previous background).  Rows never share state (locals at
// :235-240), so the encode splits into:
//   K1 PickCells   one thread per character cell: FindBestGlyph (:163-227)
//   K2 ScanRows    one wave per text row: who has to emit fg/bg (:270-297),
//                  byte length per cell, exclusive prefix inside the row
//   K3 ScanFrames  one workgroup per frame: row offsets, frame length
//   K4 EmitCells   one thread per cell: write its bytes at its offset
// Frame-diff mode (emit_difference, :244-247, :343-346): a cell whose pixels
// equal the previous frame's is skipped; skipped cells turn into cursor-right
// moves in front of the next emitted cell (:260-263), rows without any emitted
// cell into newlines / a cursor-down move in front of the next emitting row
// (:249-258, :313-315) or after the last one (:397-399).  The previous frame
// stays on the device (it is exactly what the reference's backing store holds).
#include <cstring>

#include "context.h"
#include "pixel_math.h"
#include "wave_ops.h"

namespace timg_amd {
namespace {

enum : uint32_t {
    kBackground = 0,
    kTopLeft,
    kTopRight,
    kBotLeft,
    kBotRight,
    kLeftBar,
    kTopLeftBotRight,
    kLowerBlock,
    kUpperBlock,
};

struct BlockGeom {
    int w, h;            // framebuffer size in pixels
    int quarter, upper, color256;
    int cells;           // character cells per text row
    int rows;            // text rows: (h + 1) / 2
    int row_offset;      // -1 when an odd height shifts everything down (:356-358)
    int indent;          // Send's x in character cells
    const int *indents;  // per frame (a grid row's images side by side), or null: `indent` for all
    int emit_diff;       // compare against the previous frame
    size_t stride, frame_stride;
};

struct CellRec {  // 16 bytes
    uint32_t fg, bg;
    uint32_t meta;  // block | emit_fg << 8 | emit_bg << 9
    uint32_t off;   // byte offset inside the text row
};

struct Lin {
    float r, g, b, a;
};

__device__ __forceinline__ Lin ToLin(uint32_t px) {  // LinearColor(rgba_t)
    const uint32_t r = px & 0xffu, g = (px >> 8) & 0xffu, b = (px >> 16) & 0xffu;
    Lin l;
    l.r = (float)(r * r);
    l.g = (float)(g * g);
    l.b = (float)(b * b);
    l.a = (float)(px >> 24);
    return l;
}

__device__ __forceinline__ uint32_t Repack(const Lin &l) {  // framebuffer.h:150-152
    return GammaByte(l.r) | (GammaByte(l.g) << 8) | (GammaByte(l.b) << 16) |
           (((uint32_t)l.a & 0xffu) << 24);
}

__device__ __forceinline__ float Dist(const Lin &avg, const Lin &m) {
    const float dr = m.r - avg.r, dg = m.g - avg.g, db = m.b - avg.b;
    return dr * dr + dg * dg + db * db;
}

// avd() of src/framebuffer.h:177-194 for 2, 3 and 4 members: running sum from
// zero in argument order, divide by the count, then summed squared distances.
__device__ __forceinline__ float Avd2(Lin *res, const Lin &a, const Lin &b) {
    res->r = ((0.0f + a.r) + b.r) / 2.0f;
    res->g = ((0.0f + a.g) + b.g) / 2.0f;
    res->b = ((0.0f + a.b) + b.b) / 2.0f;
    res->a = ((0.0f + a.a) + b.a) / 2.0f;
    float s = 0.0f;
    s += Dist(*res, a);
    s += Dist(*res, b);
    return s;
}
__device__ __forceinline__ float Avd3(Lin *res, const Lin &a, const Lin &b, const Lin &c) {
    res->r = (((0.0f + a.r) + b.r) + c.r) / 3.0f;
    res->g = (((0.0f + a.g) + b.g) + c.g) / 3.0f;
    res->b = (((0.0f + a.b) + b.b) + c.b) / 3.0f;
    res->a = (((0.0f + a.a) + b.a) + c.a) / 3.0f;
    float s = 0.0f;
    s += Dist(*res, a);
    s += Dist(*res, b);
    s += Dist(*res, c);
    return s;
}
__device__ __forceinline__ float Avd4(Lin *res, const Lin &a, const Lin &b, const Lin &c,
                                      const Lin &d) {
    res->r = ((((0.0f + a.r) + b.r) + c.r) + d.r) / 4.0f;
    res->g = ((((0.0f + a.g) + b.g) + c.g) + d.g) / 4.0f;
    res->b = ((((0.0f + a.b) + b.b) + c.b) + d.b) / 4.0f;
    res->a = ((((0.0f + a.a) + b.a) + c.a) + d.a) / 4.0f;
    float s = 0.0f;
    s += Dist(*res, a);
    s += Dist(*res, b);
    s += Dist(*res, c);
    s += Dist(*res, d);
    return s;
}

__device__ __forceinline__ bool Transparent(uint32_t px) { return (px >> 24) < 0x60u; }

// Pixel (row, x) as the reference's pointer arithmetic sees it: rows outside
// the image are the zeroed empty_line_ (:363-365), and column `w` of an odd-width
// quarter-block frame is whatever follows in memory -- the next row's first
// pixel, or the framebuffer's scratch row, pinned to 0.
__device__ __forceinline__ uint32_t FetchPx(const uint8_t *frame, const BlockGeom &g, int row,
                                            int x) {
    if (row < 0 || row >= g.h) return 0u;
    if (x >= g.w) {
        row += 1;
        x = 0;
        if (row >= g.h) return 0u;
    }
    return *reinterpret_cast<const uint32_t *>(frame + (size_t)row * g.stride + (size_t)x * 4);
}

__global__ void __launch_bounds__(256)
PickCellsKernel(const uint8_t *fb, const uint8_t *prev_fb, BlockGeom g, CellRec *cells) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    const int trow = blockIdx.y;
    const int f    = blockIdx.z;
    if (cell >= g.cells) return;
    const uint8_t *frame = fb + (size_t)f * g.frame_stride;
    const int top_row    = 2 * trow + g.row_offset;
    CellRec rec;
    rec.off = 0;
    if (!g.quarter) {
        const uint32_t top = FetchPx(frame, g, top_row, cell);
        const uint32_t bot = FetchPx(frame, g, top_row + 1, cell);
        if (top == bot || (Transparent(top) && Transparent(bot))) {
            rec.fg   = top;
            rec.bg   = bot;
            rec.meta = kBackground;
        } else if (g.upper) {
            rec.fg   = top;
            rec.bg   = bot;
            rec.meta = kUpperBlock;
        } else {
            rec.fg   = bot;
            rec.bg   = top;
            rec.meta = kLowerBlock;
        }
    } else {
        const int x       = cell * 2;
        const uint32_t p0 = FetchPx(frame, g, top_row, x);
        const uint32_t p1 = FetchPx(frame, g, top_row, x + 1);
        const uint32_t p2 = FetchPx(frame, g, top_row + 1, x);
        const uint32_t p3 = FetchPx(frame, g, top_row + 1, x + 1);
        const Lin tl = ToLin(p0), tr = ToLin(p1), bl = ToLin(p2), br = ToLin(p3);
        const bool t_top = Transparent(p0) && Transparent(p1);
        const bool t_bot = Transparent(p2) && Transparent(p3);
        if (t_top && t_bot) {
            rec.fg   = p2;
            rec.bg   = p0;
            rec.meta = kBackground;
        } else if (t_top) {
            Lin avg;
            Avd2(&avg, bl, br);
            rec.fg   = Repack(avg);
            rec.bg   = p0;
            rec.meta = kLowerBlock;
        } else if (t_bot) {
            Lin avg;
            Avd2(&avg, tl, tr);
            rec.fg   = Repack(avg);
            rec.bg   = p2;
            rec.meta = kUpperBlock;
        } else {
            Lin best_fg = {0, 0, 0, 0}, best_bg = {0, 0, 0, 0};
            uint32_t best_block = kBackground;
            float best          = 1e12f;
            bool done           = false;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (done) continue;
                Lin fg, bg;
                float d;
                uint32_t block = b;
                switch (b) {
                case 0: d = Avd4(&bg, tl, tr, bl, br); fg = bg; break;
                case 1: d = Avd3(&bg, tr, bl, br); fg = tl; break;
                case 2: d = Avd3(&bg, tl, bl, br); fg = tr; break;
                case 3: d = Avd3(&bg, tl, tr, br); fg = bl; break;
                case 4: d = Avd3(&bg, tl, tr, bl); fg = br; break;
                case 5: d = Avd2(&bg, tr, br) + Avd2(&fg, tl, bl); break;
                case 6: d = Avd2(&bg, tr, bl) + Avd2(&fg, tl, br); break;
                default:
                    if (g.upper) {
                        block = kUpperBlock;
                        d     = Avd2(&bg, bl, br) + Avd2(&fg, tl, tr);
                    } else {
                        block = kLowerBlock;
                        d     = Avd2(&bg, tl, tr) + Avd2(&fg, bl, br);
                    }
                    break;
                }
                if (d < best) {
                    best_fg    = fg;
                    best_bg    = bg;
                    best_block = block;
                    if (d < 1.0f)
                        done = true;  // "essentially zero": stop searching (:222)
                    else
                        best = d;
                }
            }
            rec.fg   = Repack(best_fg);
            rec.bg   = Repack(best_bg);
            rec.meta = best_block;
        }
    }
    if (g.emit_diff) {  // EqualToBacking, :129-136
        const uint8_t *prev = prev_fb + (size_t)f * g.frame_stride;
        bool same;
        if (!g.quarter) {
            same = FetchPx(frame, g, top_row, cell) == FetchPx(prev, g, top_row, cell) &&
                   FetchPx(frame, g, top_row + 1, cell) == FetchPx(prev, g, top_row + 1, cell);
        } else {
            const int x = cell * 2;
            same = FetchPx(frame, g, top_row, x) == FetchPx(prev, g, top_row, x) &&
                   FetchPx(frame, g, top_row, x + 1) == FetchPx(prev, g, top_row, x + 1) &&
                   FetchPx(frame, g, top_row + 1, x) == FetchPx(prev, g, top_row + 1, x) &&
                   FetchPx(frame, g, top_row + 1, x + 1) == FetchPx(prev, g, top_row + 1, x + 1);
        }
        if (same) rec.meta |= 0x400u;
    }
    cells[((size_t)f * g.rows + trow) * g.cells + cell] = rec;
}

__device__ __forceinline__ uint32_t DigitsLen(uint32_t v) {  // "ddd;" length
    return v >= 100 ? 4u : (v >= 10 ? 3u : 2u);
}

__device__ __forceinline__ uint32_t Term256(uint32_t px) {  // framebuffer.h:37-52
    const uint32_t r = px & 0xffu, g = (px >> 8) & 0xffu, b = (px >> 16) & 0xffu;
    if (r == g && g == b) return 232u + (r * 23u / 255u);
    auto cube = [](uint32_t v) -> uint32_t {
        return v < 0x5f / 2            ? 0
               : v < (0x5f + 0x87) / 2 ? 1
               : v < (0x87 + 0xaf) / 2 ? 2
               : v < (0xaf + 0xd7) / 2 ? 3
               : v < (0xd7 + 0xff) / 2 ? 4
                                       : 5;
    };
    return 16u + 36u * cube(r) + 6u * cube(g) + cube(b);
}

__device__ __forceinline__ uint32_t ColorLen(uint32_t px, int color256) {
    if (color256) return DigitsLen(Term256(px));
    return DigitsLen(px & 0xffu) + DigitsLen((px >> 8) & 0xffu) + DigitsLen((px >> 16) & 0xffu);
}

__device__ __forceinline__ uint32_t DecLen(uint32_t v) {
    return v >= 10000 ? 5u : v >= 1000 ? 4u : v >= 100 ? 3u : v >= 10 ? 2u : 1u;
}

// meta bits of a CellRec after ScanRows:
//   0-7 block, 8 emit fg, 9 emit bg, 10 skipped (equal to previous frame),
//   11 first emitted cell of its row, 12 last emitted cell, 16-31 cursor-right count
// One wave (64 lanes) per text row.
__global__ void __launch_bounds__(64)
ScanRowsKernel(BlockGeom g, CellRec *cells, uint32_t *row_len) {
    const int lane = threadIdx.x;
    const int trow = blockIdx.x;
    const int f    = blockIdx.y;
    CellRec *row   = cells + ((size_t)f * g.rows + trow) * g.cells;

    // state carried across 64-cell chunks (the reference's per-row locals, :235-240)
    int last_present   = -1;  // index of the last emitted cell
    uint32_t prev_bg   = 0;   // its background (pick.bg of `last`)
    bool have_fg       = false;
    uint32_t last_fg   = 0;   // last_foreground
    uint32_t base_off  = 0;

    for (int c0 = 0; c0 < g.cells; c0 += 64) {
        const int i        = c0 + lane;
        const bool live    = i < g.cells;
        CellRec rec        = live ? row[i] : CellRec{0, 0, 0, 0};
        const bool present = live && !(rec.meta & 0x400u);
        const bool nonbg   = present && (rec.meta & 0xffu) != kBackground;

        // "the nearest present / non-background cell at or below this lane": a prefix maximum of lane + 1 (0: none) on
        // the DPP data path (wave_ops.h) -- as __shfl_up loops these were six dependent ds_bpermute round trips each
        const uint32_t ip = WaveInclusiveMaxU32(present ? (uint32_t)lane + 1u : 0u);
        const uint32_t in = WaveInclusiveMaxU32(nonbg ? (uint32_t)lane + 1u : 0u);
        const int prev_p = (int)WaveShr1(ip) - 1, prev_n = (int)WaveShr1(in) - 1;  // (lane 0: -1)

        // cursor-right in front of an emitted cell: cells skipped since the last
        // emitted one; the row starts with x_skip = indent (:240)
        const int prev_present = prev_p >= 0 ? c0 + prev_p : last_present;
        const uint32_t skip    = present ? (uint32_t)(i - prev_present - 1 + (prev_present < 0 ? (g.indents ? g.indents[f] : g.indent) : 0)) : 0u;

        const uint32_t bg_from_lane = __shfl(rec.bg, prev_p < 0 ? 0 : prev_p);
        const bool left_known       = prev_p >= 0 || last_present >= 0;
        const uint32_t left_bg      = prev_p >= 0 ? bg_from_lane : prev_bg;
        const bool emit_bg          = present && (!left_known || rec.bg != left_bg);

        const uint32_t fg_from_lane = __shfl(rec.fg, prev_n < 0 ? 0 : prev_n);
        const bool fg_known         = prev_n >= 0 || have_fg;
        const uint32_t prev_fg      = prev_n >= 0 ? fg_from_lane : last_fg;
        const bool emit_fg          = nonbg && (!fg_known || rec.fg != prev_fg);

        uint32_t len = 0;
        if (present) {
            if (skip) len += 3 + DecLen(skip);  // ESC [ n C
            if (emit_fg || emit_bg) len += 2;   // ESC [
            if (emit_fg) len += 5 + ColorLen(rec.fg, g.color256);
            if (emit_bg) len += Transparent(rec.bg) ? 3u : 5u + ColorLen(rec.bg, g.color256);
            len += (rec.meta & 0xffu) == kBackground ? 1u : 3u;
        }
        const uint32_t incl = WaveInclusiveAdd(len);
        if (live) {
            rec.off  = base_off + incl - len;
            rec.meta = (rec.meta & 0x4ffu) | (emit_fg ? 0x100u : 0u) | (emit_bg ? 0x200u : 0u) |
                       ((present && prev_present < 0) ? 0x800u : 0u) | ((skip & 0xffffu) << 16);
            row[i] = rec;
        }
        // carry
        base_off += ReadLane(incl, 63);
        const int chunk_p = (int)ReadLane(ip, 63) - 1, chunk_n = (int)ReadLane(in, 63) - 1;  // (wave-uniform)
        const uint32_t cbg = ReadLane(rec.bg, chunk_p < 0 ? 0 : chunk_p);
        const uint32_t cfg = ReadLane(rec.fg, chunk_n < 0 ? 0 : chunk_n);
        if (chunk_p >= 0) {
            last_present = c0 + chunk_p;
            prev_bg      = cbg;
        }
        if (chunk_n >= 0) {
            have_fg = true;
            last_fg = cfg;
        }
    }
    if (lane == 0) {
        if (last_present >= 0) row[last_present].meta |= 0x1000u;  // writes "\033[0m\n" (:313-318)
        row_len[(size_t)f * g.rows + trow] = last_present >= 0 ? base_off + 5 : 0u;
    }
}

// One workgroup per frame: vertical skips, row offsets and the frame length.
// row_len (in/out): byte length of each text row -> its byte offset;
// row_yskip (out): empty rows immediately above each emitting row.
__device__ __forceinline__ char *PutDec(char *p, uint32_t v) {
    char t[10];
    int n = 0;
    do {
        t[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = t[--n];
    return p;
}

__global__ void __launch_bounds__(256)
ScanFramesKernel(BlockGeom g, uint32_t *row_len, uint32_t *row_yskip,
                 unsigned long long *frame_len, char *out, size_t out_cap) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t carry, trailing;
    const int f    = blockIdx.x;
    const int tid  = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    uint32_t *rl   = row_len + (size_t)f * g.rows;
    uint32_t *ys   = row_yskip + (size_t)f * g.rows;
    if (tid == 0) {
        carry      = 0;
        uint32_t k = 0;
        for (int q = g.rows - 1; q >= 0 && rl[q] == 0; --q) ++k;
        trailing = k;  // rows left empty at the bottom
    }
    // y_skip of a row = number of consecutive empty rows right above it
    for (int r = tid; r < g.rows; r += 256) {
        uint32_t k = 0;
        if (rl[r] != 0)
            for (int q = r - 1; q >= 0 && rl[q] == 0; --q) ++k;
        ys[r] = k;
    }
    __syncthreads();
    for (int r0 = 0; r0 < g.rows; r0 += 256) {
        const int r = r0 + tid;
        uint32_t v  = 0;
        if (r < g.rows && rl[r] != 0) {
            const uint32_t k = ys[r];
            // up to four '\n', else ESC [ k B (:249-258)
            v = rl[r] + (k == 0 ? 0u : (k <= 4 ? k : 3u + DecLen(k)));
        }
        const uint32_t incl = WaveInclusiveAdd(v);
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (int k = 0; k < wv; ++k) before += wave_tot[k];
        if (r < g.rows) rl[r] = before + incl - v;
        __syncthreads();
        if (tid == 255) carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t total = carry;
        // nothing emitted at all: zero-size buffer (:390-395); otherwise one
        // cursor-down move over the empty rows at the bottom (:397-399)
        if (total != 0 && trailing != 0) {
            char tmp[16];
            char *p = tmp;
            *p++    = '\033';
            *p++    = '[';
            p       = PutDec(p, trailing);
            *p++    = 'B';
            char *frame_out = out + (size_t)f * out_cap;
            for (int i = 0; i < (int)(p - tmp); ++i)
                if ((size_t)total + i < out_cap) frame_out[total + i] = tmp[i];
            total += (uint32_t)(p - tmp);
        }
        frame_len[f] = total;
    }
}

__device__ __forceinline__ char *PutNum(char *p, uint32_t v) {  // "ddd;"
    if (v >= 100) {
        *p++ = (char)('0' + v / 100);
        v %= 100;
        *p++ = (char)('0' + v / 10);
        *p++ = (char)('0' + v % 10);
    } else if (v >= 10) {
        *p++ = (char)('0' + v / 10);
        *p++ = (char)('0' + v % 10);
    } else {
        *p++ = (char)('0' + v);
    }
    *p++ = ';';
    return p;
}

__device__ __forceinline__ char *PutColor(char *p, uint32_t px, int color256) {
    if (color256) return PutNum(p, Term256(px));
    p = PutNum(p, px & 0xffu);
    p = PutNum(p, (px >> 8) & 0xffu);
    return PutNum(p, (px >> 16) & 0xffu);
}

__constant__ unsigned char kGlyphBytes[9][3] = {
    {' ', 0, 0},        {0xe2, 0x96, 0x98}, {0xe2, 0x96, 0x9d}, {0xe2, 0x96, 0x96},
    {0xe2, 0x96, 0x97}, {0xe2, 0x96, 0x8c}, {0xe2, 0x96, 0x9a}, {0xe2, 0x96, 0x84},
    {0xe2, 0x96, 0x80},
};

__global__ void __launch_bounds__(256)
EmitCellsKernel(BlockGeom g, const CellRec *cells, const uint32_t *row_off,
                const uint32_t *row_yskip, char *out, size_t out_cap) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    const int trow = blockIdx.y;
    const int f    = blockIdx.z;
    if (cell >= g.cells) return;
    const size_t row_index = (size_t)f * g.rows + trow;
    const CellRec rec      = cells[row_index * g.cells + cell];
    if (rec.meta & 0x400u) return;  // unchanged since the previous frame
    const uint32_t yskip   = (rec.meta & 0x800u) ? row_yskip[row_index] : 0u;
    // the vertical skip sits in front of the row's first emitted cell
    const size_t y_len     = yskip == 0 ? 0 : (yskip <= 4 ? yskip : 3 + DecLen(yskip));
    char *frame_out        = out + (size_t)f * out_cap;

    char buf[80];
    char *p = buf;
    if (yskip) {
        if (yskip <= 4) {
            for (uint32_t i = 0; i < yskip; ++i) *p++ = '\n';
        } else {
            *p++ = '\033';
            *p++ = '[';
            p    = PutDec(p, yskip);
            *p++ = 'B';
        }
    }
    const uint32_t skip = rec.meta >> 16;
    if (skip) {  // "\033[<n>C", :260-263
        *p++ = '\033';
        *p++ = '[';
        p    = PutDec(p, skip);
        *p++ = 'C';
    }
    const uint32_t block = rec.meta & 0xffu;
    const bool emit_fg = rec.meta & 0x100u, emit_bg = rec.meta & 0x200u;
    if (emit_fg || emit_bg) {
        *p++ = '\033';
        *p++ = '[';
    }
    if (emit_fg) {
        *p++ = '3'; *p++ = '8'; *p++ = ';'; *p++ = g.color256 ? '5' : '2'; *p++ = ';';
        p = PutColor(p, rec.fg, g.color256);
    }
    if (emit_bg) {
        if (Transparent(rec.bg)) {
            *p++ = '4'; *p++ = '9'; *p++ = ';';
        } else {
            *p++ = '4'; *p++ = '8'; *p++ = ';'; *p++ = g.color256 ? '5' : '2'; *p++ = ';';
            p = PutColor(p, rec.bg, g.color256);
        }
    }
    if (emit_fg || emit_bg) p[-1] = 'm';
    if (block == kBackground) {
        *p++ = ' ';
    } else {
        *p++ = (char)kGlyphBytes[block][0];
        *p++ = (char)kGlyphBytes[block][1];
        *p++ = (char)kGlyphBytes[block][2];
    }
    if (rec.meta & 0x1000u) {  // last emitted cell of the row: "\033[0m\n", :313-318
        *p++ = '\033'; *p++ = '['; *p++ = '0'; *p++ = 'm'; *p++ = '\n';
    }
    // row offsets already include every earlier row's vertical skip; this row's
    // own skip shifts all of its cells but the first one
    const size_t at = (size_t)row_off[row_index] + rec.off + ((rec.meta & 0x800u) ? 0 : 0);
    const int n     = (int)(p - buf);
    (void)y_len;
    for (int i = 0; i < n; ++i)
        if (at + i < out_cap) frame_out[at + i] = buf[i];
}

// Adds a row's own vertical-skip length to the offsets of all of its cells but
// the first emitted one (which carries the skip bytes itself).
__global__ void __launch_bounds__(256)
ShiftRowsKernel(BlockGeom g, CellRec *cells, const uint32_t *row_yskip) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    const int trow = blockIdx.y;
    const int f    = blockIdx.z;
    if (cell >= g.cells) return;
    const size_t row_index = (size_t)f * g.rows + trow;
    const uint32_t k       = row_yskip[row_index];
    if (k == 0) return;
    CellRec *rec = &cells[row_index * g.cells + cell];
    if (rec->meta & 0xc00u) return;  // skipped cell, or the first emitted one
    rec->off += k <= 4 ? k : 3u + DecLen(k);
}

struct BlockScratch {
    CellRec *cells;
    uint32_t *row_len, *row_yskip;
    unsigned long long *frame_len;
};

hipError_t RunBlockEncode(const uint8_t *fb, const uint8_t *prev, const BlockGeom &g, int n_frames,
                          const BlockScratch &sc, char *out, size_t out_cap, hipStream_t st) {
    const dim3 cell_grid((g.cells + 255) / 256, g.rows, n_frames);
    hipLaunchKernelGGL(PickCellsKernel, cell_grid, dim3(256), 0, st, fb, prev, g, sc.cells);
    hipLaunchKernelGGL(ScanRowsKernel, dim3(g.rows, n_frames), dim3(64), 0, st, g, sc.cells,
                       sc.row_len);
    hipLaunchKernelGGL(ScanFramesKernel, dim3(n_frames), dim3(256), 0, st, g, sc.row_len,
                       sc.row_yskip, sc.frame_len, out, out_cap);
    if (g.emit_diff)
        hipLaunchKernelGGL(ShiftRowsKernel, cell_grid, dim3(256), 0, st, g, sc.cells, sc.row_yskip);
    hipLaunchKernelGGL(EmitCellsKernel, cell_grid, dim3(256), 0, st, g, sc.cells, sc.row_len,
                       sc.row_yskip, out, out_cap);
    return hipGetLastError();
}

BlockGeom MakeGeom(int w, int h, int stride, size_t frame_stride, int flags, int x_indent) {
    BlockGeom g;
    g.w            = w;
    g.h            = h;
    g.quarter      = (flags & TIMG_HIP_BLOCK_QUARTER) != 0;
    g.upper        = (flags & TIMG_HIP_BLOCK_UPPER) != 0;
    g.color256     = (flags & TIMG_HIP_BLOCK_COLOR256) != 0;
    g.cells        = g.quarter ? (w + 1) / 2 : w;
    g.rows         = (h + 1) / 2;
    g.row_offset   = ((h & 1) && !g.upper) ? -1 : 0;
    g.indent       = g.quarter ? x_indent / 2 : x_indent;
    g.indents      = nullptr;
    g.emit_diff    = 0;
    g.stride       = (size_t)stride;
    g.frame_stride = frame_stride;
    return g;
}

}  // namespace
}  // namespace timg_amd

using namespace timg_amd;

// Stateful canvas: what UnicodeBlockCanvas keeps between Sends (:66-79 of the
// header): last height / indent and the previous frame (its backing store).
struct timg_hip_block_canvas {
    timg_hip_ctx *ctx = nullptr;
    int flags         = 0;
    int last_height = 0, last_x_indent = 0, last_width = 0;
    TimgBuffer prev;     // previous frame, device
    TimgBuffer cur;      // staging for host frames
    TimgBuffer scratch;  // cells + rows
    TimgBuffer dout;     // device output when the caller's buffer is on the host
};

extern "C" {

size_t timg_hip_block_max_bytes(int w, int h) {
    // RequestBuffers (:405-424): cursor-up + per text row (cursor-right +
    // width * widest cell + end of line)
    const size_t max_pixel = 2 + 5 + 11 + 1 + 5 + 11 + 1 + 3;
    const size_t rows      = (size_t)(h + 1) / 2;
    return 8 + rows * (8 + (size_t)w * max_pixel + 5);
}

static int BlockEncodeImpl(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride,
                           size_t frame_stride, int fb_on_device, int n_frames, int flags, int x_indent,
                           const int *x_indents, char *out, size_t out_cap, int out_on_device,
                           size_t *out_len, void *stream) {
    if (!ctx || !fb || !out || !out_len || w <= 0 || h <= 0 || n_frames <= 0 || x_indent < 0)
        return TIMG_HIP_ERR_ARG;
    if (x_indents)
        for (int i = 0; i < n_frames; ++i)
            if (x_indents[i] < 0) return TIMG_HIP_ERR_ARG;
    if (stride == 0) stride = w * 4;
    if (stride < w * 4 || (stride & 3) || ((uintptr_t)fb & 3))
        return ctx->Fail(TIMG_HIP_ERR_ARG, "bad stride/alignment");
    if (frame_stride == 0) frame_stride = (size_t)stride * h;
    TIMG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->Stream(stream);
    std::lock_guard<std::mutex> lock(ctx->mu);

    BlockGeom g = MakeGeom(w, h, stride, frame_stride, flags, x_indent);
    if (x_indents) {  // per-frame Send x (in cells), uploaded next to the other scratch
        TIMG_HIP_TRY(ctx, ctx->dev[6].Reserve(sizeof(int) * n_frames));
        TIMG_HIP_TRY(ctx, ctx->pin[1].Reserve(sizeof(int) * n_frames));
        int *host = (int *)ctx->pin[1].ptr;
        for (int i = 0; i < n_frames; ++i) host[i] = g.quarter ? x_indents[i] / 2 : x_indents[i];
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(ctx->dev[6].ptr, host, sizeof(int) * n_frames, hipMemcpyHostToDevice, st));
        g.indents = (const int *)ctx->dev[6].ptr;
    }
    const size_t fb_bytes = frame_stride * (size_t)(n_frames - 1) + (size_t)stride * h;
    const uint8_t *dfb    = fb;
    if (!fb_on_device) {
        TIMG_HIP_TRY(ctx, ctx->dev[0].Reserve(fb_bytes));
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(ctx->dev[0].ptr, fb, fb_bytes, hipMemcpyHostToDevice, st));
        dfb = (const uint8_t *)ctx->dev[0].ptr;
    }
    char *dout = out;
    if (!out_on_device) {
        TIMG_HIP_TRY(ctx, ctx->dev[1].Reserve(out_cap * (size_t)n_frames));
        dout = (char *)ctx->dev[1].ptr;
    }
    const size_t n_rows  = (size_t)g.rows * n_frames;
    const size_t n_cells = n_rows * g.cells;
    TIMG_HIP_TRY(ctx, ctx->dev[3].Reserve(n_cells * sizeof(CellRec)));
    TIMG_HIP_TRY(ctx, ctx->dev[4].Reserve(2 * n_rows * sizeof(uint32_t)));
    TIMG_HIP_TRY(ctx, ctx->dev[2].Reserve(sizeof(unsigned long long) * n_frames));
    TIMG_HIP_TRY(ctx, ctx->pin[0].Reserve(sizeof(unsigned long long) * n_frames));
    BlockScratch sc;
    sc.cells      = (CellRec *)ctx->dev[3].ptr;
    sc.row_len    = (uint32_t *)ctx->dev[4].ptr;
    sc.row_yskip  = sc.row_len + n_rows;
    sc.frame_len  = (unsigned long long *)ctx->dev[2].ptr;
    unsigned long long *flen_h = (unsigned long long *)ctx->pin[0].ptr;

    TIMG_HIP_TRY(ctx, RunBlockEncode(dfb, nullptr, g, n_frames, sc, dout, out_cap, st));
    TIMG_HIP_TRY(ctx, hipMemcpyAsync(flen_h, sc.frame_len, sizeof(unsigned long long) * n_frames,
                                     hipMemcpyDeviceToHost, st));
    TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
    size_t worst = 0;
    for (int i = 0; i < n_frames; ++i) {
        out_len[i] = (size_t)flen_h[i];
        if (out_len[i] > worst) worst = out_len[i];
    }
    if (worst > out_cap)
        return ctx->Fail(TIMG_HIP_ERR_SMALL, "frame needs %zu bytes, out_cap is %zu", worst, out_cap);
    if (!out_on_device) return CopyFramesToHost(ctx, out, out_cap, dout, out_len, n_frames, st);
    return TIMG_HIP_OK;
}

int timg_hip_block_encode(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride,
                          size_t frame_stride, int fb_on_device, int n_frames, int flags,
                          int x_indent, char *out, size_t out_cap, int out_on_device,
                          size_t *out_len, void *stream) {
    return BlockEncodeImpl(ctx, fb, w, h, stride, frame_stride, fb_on_device, n_frames, flags, x_indent,
                           nullptr, out, out_cap, out_on_device, out_len, stream);
}

int timg_hip_block_encode_grid(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride,
                               size_t frame_stride, int fb_on_device, int n_frames, int flags,
                               const int *x_indents, char *out, size_t out_cap, int out_on_device,
                               size_t *out_len, void *stream) {
    if (!x_indents) return TIMG_HIP_ERR_ARG;
    return BlockEncodeImpl(ctx, fb, w, h, stride, frame_stride, fb_on_device, n_frames, flags, 0,
                           x_indents, out, out_cap, out_on_device, out_len, stream);
}

int timg_hip_block_canvas_create(timg_hip_ctx *ctx, int flags, timg_hip_block_canvas **out) {
    if (!ctx || !out) return TIMG_HIP_ERR_ARG;
    timg_hip_block_canvas *c = new (std::nothrow) timg_hip_block_canvas();
    if (!c) return TIMG_HIP_ERR_NOMEM;
    c->ctx   = ctx;
    c->flags = flags;
    *out     = c;
    return TIMG_HIP_OK;
}

void timg_hip_block_canvas_destroy(timg_hip_block_canvas *c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    c->prev.Release();
    c->cur.Release();
    c->scratch.Release();
    c->dout.Release();
    delete c;
}

void timg_hip_block_canvas_forget(timg_hip_block_canvas *c) {
    if (c) c->last_height = c->last_width = 0;
}

int timg_hip_block_canvas_send(timg_hip_block_canvas *c, int x, int dy, const uint8_t *fb, int w,
                               int h, int stride, int fb_on_device, char *out, size_t out_cap,
                               size_t *out_len, void *stream) {
    if (!c || !fb || !out || !out_len || w <= 0 || h <= 0 || x < 0) return TIMG_HIP_ERR_ARG;
    timg_hip_ctx *ctx = c->ctx;
    if (stride == 0) stride = w * 4;
    if (stride < w * 4 || (stride & 3) || ((uintptr_t)fb & 3))
        return ctx->Fail(TIMG_HIP_ERR_ARG, "bad stride/alignment");
    TIMG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->Stream(stream);
    BlockGeom g    = MakeGeom(w, h, w * 4, (size_t)w * 4 * h, c->flags, x);
    // :344-346 (plus: the backing store must describe a frame of this width)
    g.emit_diff = (g.indent == c->last_x_indent) && (c->last_height > 0) &&
                  (dy < 0 ? -dy : dy) == c->last_height && c->last_width == w && c->last_height == h;
    const size_t bytes = (size_t)w * 4 * h;
    TIMG_HIP_TRY(ctx, c->cur.Reserve(bytes));
    TIMG_HIP_TRY(ctx, c->prev.Reserve(bytes));
    // tightly packed device copy of the frame (it becomes the next backing store)
    TIMG_HIP_TRY(ctx, hipMemcpy2DAsync(c->cur.ptr, (size_t)w * 4, fb, (size_t)stride, (size_t)w * 4,
                                       (size_t)h, fb_on_device ? hipMemcpyDeviceToDevice
                                                               : hipMemcpyHostToDevice, st));
    const size_t n_rows = (size_t)g.rows, n_cells = n_rows * g.cells;
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_rows = align(n_cells * sizeof(CellRec));
    const size_t o_len  = align(o_rows + 2 * n_rows * sizeof(uint32_t));
    TIMG_HIP_TRY(ctx, c->scratch.Reserve(o_len + 64));
    TIMG_HIP_TRY(ctx, c->dout.Reserve(out_cap));
    BlockScratch sc;
    sc.cells     = (CellRec *)c->scratch.ptr;
    sc.row_len   = (uint32_t *)((char *)c->scratch.ptr + o_rows);
    sc.row_yskip = sc.row_len + n_rows;
    sc.frame_len = (unsigned long long *)((char *)c->scratch.ptr + o_len);
    TIMG_HIP_TRY(ctx, RunBlockEncode((const uint8_t *)c->cur.ptr, (const uint8_t *)c->prev.ptr, g, 1,
                                     sc, (char *)c->dout.ptr, out_cap, st));
    unsigned long long len = 0;
    TIMG_HIP_TRY(ctx, hipMemcpyAsync(&len, sc.frame_len, sizeof(len), hipMemcpyDeviceToHost, st));
    TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
    // the frame just shown is the new backing store
    std::swap(c->cur, c->prev);
    c->last_height   = h;
    c->last_width    = w;
    c->last_x_indent = g.indent;
    *out_len         = (size_t)len;
    if (len > out_cap)
        return ctx->Fail(TIMG_HIP_ERR_SMALL, "frame needs %llu bytes, out_cap is %zu", len, out_cap);
    if (len) TIMG_HIP_TRY(ctx, hipMemcpy(out, c->dout.ptr, (size_t)len, hipMemcpyDeviceToHost));
    return TIMG_HIP_OK;
}

}  // extern "C"
