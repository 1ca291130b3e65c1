/* oracle/block.c -- TEST INFRASTRUCTURE ONLY (see timg_oracle.h).
 *
 * Restates timg::UnicodeBlockCanvas (src/unicode-block-canvas.cc): half- and
 * quarter-block glyph choice, SGR colour elision, frame-diff against a
 * backing store, cursor skips.  Byte-exact target.  The TerminalCanvas cursor
 * prefix for dy<0 (src/terminal-canvas.cc:66-73) is included because Send
 * emits it at the front of the same buffer (unicode-block-canvas.cc:330-332).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_internal.h"
#include "timg_oracle.h"

/* unicode-block-canvas.cc:54-65 */
enum {
    kBackground,
    kTopLeft,
    kTopRight,
    kBotLeft,
    kBotRight,
    kLeftBar,
    kTopLeftBotRight,
    kLowerBlock,
    kUpperBlock
};

/* UTF-8 of U+2598 U+259D U+2596 U+2597 U+258C U+259A U+2584 U+2580
 * (unicode-block-canvas.cc:77-88) */
static const unsigned char kGlyph[9][3] = {
    {' ', 0, 0},          {0xe2, 0x96, 0x98}, {0xe2, 0x96, 0x9d},
    {0xe2, 0x96, 0x96},   {0xe2, 0x96, 0x97}, {0xe2, 0x96, 0x8c},
    {0xe2, 0x96, 0x9a},   {0xe2, 0x96, 0x84}, {0xe2, 0x96, 0x80},
};

struct oracle_block_canvas {
    int quarter, upper, color256;
    uint8_t *backing; /* (w+1)*(h+1) pixels, unicode-block-canvas.cc:428-432 */
    size_t backing_size;
    int last_height, last_x_indent;
};

typedef struct {
    uint8_t fg[4], bg[4];
    int block;
} pick_t;

static int px_eq(const uint8_t *a, const uint8_t *b) {
    return memcmp(a, b, 4) == 0;
}
static int is_transparent(const uint8_t *c) { return c[3] < 0x60; } /* :154 */

/* framebuffer.h:177-194: average into res (starting from zero) and return the
 * summed squared distance of every member to the average. */
static float avd(lin_t *res, const lin_t *v, int n) {
    res->r = res->g = res->b = res->a = 0;
    for (int i = 0; i < n; i++) {
        res->r += v[i].r;
        res->g += v[i].g;
        res->b += v[i].b;
        res->a += v[i].a;
    }
    const float fn = (float)n; /* size_t -> float in `res->r /= n` */
    res->r /= fn;
    res->g /= fn;
    res->b /= fn;
    res->a /= fn;
    float sum = 0;
    for (int i = 0; i < n; i++) {
        float dr = v[i].r - res->r, dg = v[i].g - res->g, db = v[i].b - res->b;
        sum += dr * dr + dg * dg + db * db; /* framebuffer.h:145-148 */
    }
    return sum;
}

/* unicode-block-canvas.cc:163-227 */
static pick_t find_best_glyph(const oracle_block_canvas *c, const uint8_t *top,
                              const uint8_t *bottom) {
    pick_t p;
    if (!c->quarter) {
        if (px_eq(top, bottom) ||
            (is_transparent(top) && is_transparent(bottom))) {
            memcpy(p.fg, top, 4);
            memcpy(p.bg, bottom, 4);
            p.block = kBackground;
            return p;
        }
        if (c->upper) {
            memcpy(p.fg, top, 4);
            memcpy(p.bg, bottom, 4);
            p.block = kUpperBlock;
        } else {
            memcpy(p.fg, bottom, 4);
            memcpy(p.bg, top, 4);
            p.block = kLowerBlock;
        }
        return p;
    }
    const lin_t tl = lin_from_rgba(top), tr = lin_from_rgba(top + 4);
    const lin_t bl = lin_from_rgba(bottom), br = lin_from_rgba(bottom + 4);
    const int t_tl = is_transparent(top), t_tr = is_transparent(top + 4);
    const int t_bl = is_transparent(bottom), t_br = is_transparent(bottom + 4);
    if (t_tl && t_tr && t_bl && t_br) { /* :182-185 */
        memcpy(p.fg, bottom, 4);
        memcpy(p.bg, top, 4);
        p.block = kBackground;
        return p;
    }
    if (t_tl && t_tr) { /* :186-188 */
        lin_t avg, v[2] = {bl, br};
        avd(&avg, v, 2);
        lin_repack(&avg, p.fg);
        memcpy(p.bg, top, 4);
        p.block = kLowerBlock;
        return p;
    }
    if (t_bl && t_br) { /* :189-191 */
        lin_t avg, v[2] = {tl, tr};
        avd(&avg, v, 2);
        lin_repack(&avg, p.fg);
        memcpy(p.bg, bottom, 4);
        p.block = kUpperBlock;
        return p;
    }
    lin_t best_fg = {0, 0, 0, 0}, best_bg = {0, 0, 0, 0};
    int best_block      = kBackground;
    float best_distance = 1e12f;
    for (int b = 0; b < 8; ++b) { /* :198-225 */
        float d;
        lin_t fg, bg;
        const int block = b < 7 ? b : (c->upper ? kUpperBlock : kLowerBlock);
        switch (block) {
        case kBackground: {
            lin_t v[4] = {tl, tr, bl, br};
            d          = avd(&bg, v, 4);
            fg         = bg;
        } break;
        case kTopLeft: {
            lin_t v[3] = {tr, bl, br};
            d          = avd(&bg, v, 3);
            fg         = tl;
        } break;
        case kTopRight: {
            lin_t v[3] = {tl, bl, br};
            d          = avd(&bg, v, 3);
            fg         = tr;
        } break;
        case kBotLeft: {
            lin_t v[3] = {tl, tr, br};
            d          = avd(&bg, v, 3);
            fg         = bl;
        } break;
        case kBotRight: {
            lin_t v[3] = {tl, tr, bl};
            d          = avd(&bg, v, 3);
            fg         = br;
        } break;
        case kLeftBar: {
            lin_t v[2] = {tr, br}, u[2] = {tl, bl};
            d = avd(&bg, v, 2) + avd(&fg, u, 2);
        } break;
        case kTopLeftBotRight: {
            lin_t v[2] = {tr, bl}, u[2] = {tl, br};
            d = avd(&bg, v, 2) + avd(&fg, u, 2);
        } break;
        case kLowerBlock: {
            lin_t v[2] = {tl, tr}, u[2] = {bl, br};
            d = avd(&bg, v, 2) + avd(&fg, u, 2);
        } break;
        default: { /* kUpperBlock */
            lin_t v[2] = {bl, br}, u[2] = {tl, tr};
            d = avd(&bg, v, 2) + avd(&fg, u, 2);
        } break;
        }
        if (d < best_distance) {
            best_fg    = fg;
            best_bg    = bg;
            best_block = block;
            if (d < 1) break;
            best_distance = d;
        }
    }
    lin_repack(&best_fg, p.fg);
    lin_repack(&best_bg, p.bg);
    p.block = best_block;
    return p;
}

/* "ddd;" (unicode-block-canvas.cc:454-491) */
static char *int_semicolon(char *buf, uint8_t v) {
    return buf + sprintf(buf, "%d;", v);
}

static char *write_color(const oracle_block_canvas *c, char *buf,
                         const uint8_t *col) { /* :113-122 */
    if (c->color256) return int_semicolon(buf, term256(col));
    buf = int_semicolon(buf, col[0]);
    buf = int_semicolon(buf, col[1]);
    return int_semicolon(buf, col[2]);
}

/* :231-321.  `prev` walks the backing store (2*N pixels per cell). */
static char *append_double_row(oracle_block_canvas *c, char *pos, int indent,
                               int width, const uint8_t *tline,
                               const uint8_t *bline, int emit_diff, int *y_skip,
                               uint8_t **prev) {
    const int N = c->quarter ? 2 : 1;
    pick_t last;
    memset(&last, 0, sizeof(last));
    uint8_t last_fg[4]  = {0, 0, 0, 0};
    int last_fg_unknown = 1, last_bg_unknown = 1;
    int x_skip        = indent;
    const char *start = pos;
    for (int x = 0; x < width;
         x += N, *prev += 2 * N * 4, tline += N * 4, bline += N * 4) {
        uint8_t *bk = *prev;
        if (emit_diff) { /* :129-136, 244-247 */
            int eq = N == 1 ? (px_eq(tline, bk) && px_eq(bline, bk + 4))
                            : (px_eq(tline, bk) && px_eq(tline + 4, bk + 4) &&
                               px_eq(bline, bk + 8) && px_eq(bline + 4, bk + 12));
            if (eq) {
                ++x_skip;
                continue;
            }
        }
        if (*y_skip) { /* :249-258 */
            if (*y_skip <= 4) {
                memset(pos, '\n', (size_t)*y_skip);
                pos += *y_skip;
            } else
                pos += sprintf(pos, "\033[%dB", *y_skip);
            *y_skip = 0;
        }
        if (x_skip > 0) { /* :260-263 */
            pos += sprintf(pos, "\033[%dC", x_skip);
            x_skip = 0;
        }
        const pick_t pick = find_best_glyph(c, tline, bline);
        int color_emitted = 0;
        if (pick.block != kBackground &&
            (last_fg_unknown || !px_eq(pick.fg, last_fg))) { /* :270-279 */
            memcpy(pos, "\033[", 2);
            pos += 2;
            memcpy(pos, c->color256 ? "38;5;" : "38;2;", 5);
            pos += 5;
            pos           = write_color(c, pos, pick.fg);
            color_emitted = 1;
            memcpy(last_fg, pick.fg, 4);
            last_fg_unknown = 0;
        }
        if (last_bg_unknown || !px_eq(pick.bg, last.bg)) { /* :282-297 */
            if (!color_emitted) {
                memcpy(pos, "\033[", 2);
                pos += 2;
            }
            if (is_transparent(pick.bg)) {
                memcpy(pos, "49;", 3);
                pos += 3;
            } else {
                memcpy(pos, c->color256 ? "48;5;" : "48;2;", 5);
                pos += 5;
                pos = write_color(c, pos, pick.bg);
            }
            color_emitted   = 1;
            last_bg_unknown = 0;
        }
        if (color_emitted) *(pos - 1) = 'm'; /* :299-301 */
        if (pick.block == kBackground)
            *pos++ = ' ';
        else {
            memcpy(pos, kGlyph[pick.block], 3);
            pos += 3;
        }
        last = pick;
        /* StoreBacking, :138-152 */
        if (N == 1) {
            memcpy(bk, tline, 4);
            memcpy(bk + 4, bline, 4);
        } else {
            memcpy(bk, tline, 8);
            memcpy(bk + 8, bline, 8);
        }
    }
    if (pos == start)
        (*y_skip)++;
    else {
        memcpy(pos, "\033[0m\n", 5);
        pos += 5;
    }
    return pos;
}

size_t oracle_block_max_bytes(int w, int h) { /* :405-424 */
    const int max_pixel_size = 2 + 5 + 11 + 1 + 5 + 11 + 1 + 3;
    const int vertical_chars = (h + 1) / 2;
    return (size_t)(5 + 3) +
           (size_t)vertical_chars * ((size_t)(5 + 3) + (size_t)w * max_pixel_size + 5);
}

oracle_block_canvas *oracle_block_canvas_new(int quarter, int upper_block,
                                             int color256) {
    oracle_block_canvas *c = (oracle_block_canvas *)calloc(1, sizeof(*c));
    c->quarter  = quarter != 0;
    c->upper    = upper_block != 0;
    c->color256 = color256 != 0;
    return c;
}

void oracle_block_canvas_free(oracle_block_canvas *c) {
    if (!c) return;
    free(c->backing);
    free(c);
}

/* :323-403 */
long oracle_block_canvas_send(oracle_block_canvas *c, int x, int dy,
                              const uint8_t *fb, int w, int h, char *out,
                              long cap) {
    size_t need = oracle_block_max_bytes(w, h) + 32;
    char *buf   = (char *)malloc(need);
    char *pos   = buf;
    /* terminal-canvas.cc:66-73 via :330; cell_height_for_pixels is
     * (pixels - 1) / 2 on a non-positive argument (unicode-block-canvas.h:42) */
    if (dy < 0) {
        int rows = (dy - 1) / 2;
        if (rows != 0)
            pos += sprintf(pos, rows < 0 ? "\033[%dA" : "\033[%dB", abs(rows));
    }
    if (c->quarter) x /= 2;
    const char *before_image = pos;

    size_t new_backing = (size_t)(w + 1) * (h + 1) * 4; /* :428-432 */
    if (new_backing > c->backing_size) {
        c->backing      = (uint8_t *)realloc(c->backing, new_backing);
        c->backing_size = new_backing;
    }
    uint8_t *prev       = c->backing;
    const int emit_diff = (x == c->last_x_indent) && (c->last_height > 0) &&
                          abs(dy) == c->last_height; /* :344-346 */
    uint8_t *empty_line = (uint8_t *)calloc((size_t)w + 1, 4);
    /* The reference reads one pixel past the last row for odd quarter-block
     * widths (framebuffer scratch row); pinned to transparent black. */
    uint8_t *padded = (uint8_t *)calloc((size_t)w * h + w + 1, 4);
    memcpy(padded, fb, (size_t)w * h * 4);

    const int needs_empty_line = (h % 2 != 0);
    const int row_offset       = (needs_empty_line && !c->upper) ? -1 : 0;
    int y_skip                 = 0;
    for (int y = 0; y < h; y += 2) {
        const int row        = y + row_offset;
        const uint8_t *top   = row < 0 ? empty_line : padded + (size_t)w * row * 4;
        const uint8_t *bot   = (row + 1) >= h
                                   ? empty_line
                                   : padded + (size_t)w * (row + 1) * 4;
        pos = append_double_row(c, pos, x, w, top, bot, emit_diff, &y_skip,
                                &prev);
    }
    c->last_height   = h;
    c->last_x_indent = x;
    long n;
    if (before_image == pos) {
        n = 0; /* zero-size buffer, :390-395 */
    } else {
        if (y_skip) pos += sprintf(pos, "\033[%dB", y_skip);
        n = (long)(pos - buf);
    }
    free(empty_line);
    free(padded);
    if (n > cap) {
        free(buf);
        return -1;
    }
    memcpy(out, buf, (size_t)n);
    free(buf);
    return n;
}

long oracle_block_encode(const uint8_t *fb, int w, int h, int quarter,
                         int upper_block, int color256, int x, char *out,
                         long cap) {
    oracle_block_canvas *c = oracle_block_canvas_new(quarter, upper_block, color256);
    long n = oracle_block_canvas_send(c, x, 0, fb, w, h, out, cap);
    oracle_block_canvas_free(c);
    return n;
}
