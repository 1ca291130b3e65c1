/* oracle/sixel.c -- TEST INFRASTRUCTURE ONLY (see timg_oracle.h).
 *
 * PARITY UNPINNED.  timg's SixelCanvas (src/sixel-canvas.cc:100-155) hands all
 * pixel work to libsixel (sixel_dither_new / sixel_dither_initialize /
 * sixel_encode, :137-145).  libsixel is an external, un-vendored, un-pinned
 * dependency (CMakeLists.txt:44-46) that is absent from /root/reference and
 * from this image, and the reference has no tests or golden vectors for it.
 * This file therefore restates libsixel's published algorithm (1.8.x:
 * src/quant.c computeHistogram / mediancut / lookup_fast / diffuse_fs,
 * src/tosixel.c sixel_encode_header/body/footer) from its documented
 * behaviour, anchored on the call contract at the reference's call site:
 *   256 colours, SIXEL_LARGE_LUM, SIXEL_REP_AVERAGE_COLORS, SIXEL_QUALITY_AUTO
 *   (=> the "low" sampling budget at 256 colours), RGBA8888 input with alpha
 *   dropped, default Floyd-Steinberg diffusion, RGB palette in percent.
 * What can be and is verified independently: the emitted stream is valid
 * sixel (oracle_sixel_decode round trip), uses <=256 colours, and reproduces
 * the input within a Delta-E bound (tests/test_sixel_oracle.py).
 *
 * lookup_mode 0: libsixel's lossy nearest-colour cache -- the first pixel that
 *   lands in a 15-bit (5:5:5) cell decides the palette entry of every later
 *   pixel in that cell, in raster order.
 * lookup_mode 1: the same 15-bit cell granularity, but each cell's entry is
 *   the palette colour nearest to the cell's centre.  Deterministic and
 *   order-free, so the GPU can build the whole table up front; this is the
 *   variant the HIP path implements bit-exactly.
 */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "timg_oracle.h"

/* ---- palette: 15-bit histogram + median cut ----------------------------- */
typedef struct {
    unsigned char c[3];
    unsigned int count;
} hcolor_t;

typedef struct {
    unsigned int ind, colors, sum;
} box_t;

static unsigned int hash555(const unsigned char *p) {
    return ((unsigned int)(p[0] >> 3) << 10) | ((unsigned int)(p[1] >> 3) << 5) |
           (unsigned int)(p[2] >> 3);
}

/* Stable merge sorts: libsixel calls qsort(), whose order among equal keys is
 * unspecified; we pin it to "stable" (what glibc's merge-sort qsort gives).
 * How much hangs on that choice is MEASURED, not assumed: oracle_sixel_set_tie_order() turns the order among equal
 * keys around -- bit 0 for the colours of a box (equal in the split plane), bit 1 for boxes of equal weight -- i.e. the
 * other extreme an unstable qsort could produce; tests/test_sixel_oracle.py bounds what that does to the palette and
 * to the decoded picture.  0 (default) is the pinned order every other test and the device implement. */
static int g_tie_order = 0;
void oracle_sixel_set_tie_order(int mode) { g_tie_order = mode; }
static void sort_colors_by_plane(hcolor_t *a, unsigned int n, int plane,
                                 hcolor_t *tmp) {
    if (n < 2) return;
    unsigned int h = n / 2;
    sort_colors_by_plane(a, h, plane, tmp);
    sort_colors_by_plane(a + h, n - h, plane, tmp);
    unsigned int i = 0, j = h, k = 0;
    while (i < h && j < n)
        tmp[k++] = ((g_tie_order & 1) ? a[j].c[plane] <= a[i].c[plane] : a[j].c[plane] < a[i].c[plane]) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof(hcolor_t));
}

static void sort_boxes_by_sum_desc(box_t *b, unsigned int n) {
    for (unsigned int i = 1; i < n; i++) { /* stable insertion sort */
        box_t v        = b[i];
        unsigned int j = i;
        while (j > 0 && ((g_tie_order & 2) ? b[j - 1].sum <= v.sum : b[j - 1].sum < v.sum)) {
            b[j] = b[j - 1];
            j--;
        }
        b[j] = v;
    }
}

/* returns ncolors (<=256); *origcolors = distinct sampled 15-bit colours */
static int make_palette(const unsigned char *rgb, int w, int h,
                        unsigned char *pal, int *origcolors) {
    const unsigned int reqcolors = 256, depth = 3;
    const unsigned int length    = (unsigned int)w * (unsigned int)h * depth;
    /* SIXEL_QUALITY_AUTO with >8 colours => LOW: max_sample 18383 */
    const unsigned int max_sample = 18383;
    unsigned int step             = length / depth / max_sample * depth;
    if (length < max_sample * depth) step = 6 * depth;
    if (step <= 0) step = depth;

    unsigned short *hist = (unsigned short *)calloc(1 << 15, sizeof(unsigned short));
    unsigned short *refs = (unsigned short *)malloc((1 << 15) * sizeof(unsigned short));
    unsigned int nref    = 0;
    for (unsigned int i = 0; i < length; i += step) {
        unsigned int b = hash555(rgb + i);
        if (hist[b] == 0) refs[nref++] = (unsigned short)b;
        if (hist[b] < 65535) hist[b]++;
    }
    hcolor_t *tab = (hcolor_t *)malloc((size_t)(nref ? nref : 1) * sizeof(hcolor_t));
    for (unsigned int i = 0; i < nref; i++) {
        unsigned int b = refs[i];
        tab[i].count   = hist[b];
        tab[i].c[0]    = (unsigned char)(((b >> 10) & 0x1f) << 3);
        tab[i].c[1]    = (unsigned char)(((b >> 5) & 0x1f) << 3);
        tab[i].c[2]    = (unsigned char)((b & 0x1f) << 3);
    }
    free(hist);
    free(refs);
    *origcolors = (int)nref;
    int ncolors;
    if (nref <= reqcolors) { /* "image already has few enough colours" */
        for (unsigned int i = 0; i < nref; i++) memcpy(pal + i * 3, tab[i].c, 3);
        ncolors = (int)nref;
    } else {
        box_t bv[256];
        hcolor_t *tmp      = (hcolor_t *)malloc((size_t)nref * sizeof(hcolor_t));
        unsigned int boxes = 1, sum = 0;
        for (unsigned int i = 0; i < nref; i++) sum += tab[i].count;
        bv[0].ind    = 0;
        bv[0].colors = nref;
        bv[0].sum    = sum;
        int multicolor = nref > 1;
        while (boxes < reqcolors && multicolor) {
            unsigned int bi;
            for (bi = 0; bi < boxes && bv[bi].colors < 2; ++bi) {}
            if (bi >= boxes) {
                multicolor = 0;
                break;
            }
            unsigned int start = bv[bi].ind, size = bv[bi].colors, sm = bv[bi].sum;
            unsigned char mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0};
            for (unsigned int i = 0; i < size; i++)
                for (int p = 0; p < 3; p++) {
                    unsigned char v = tab[start + i].c[p];
                    if (v < mn[p]) mn[p] = v;
                    if (v > mx[p]) mx[p] = v;
                }
            /* SIXEL_LARGE_LUM: luminosity-weighted spread */
            static const double lum[3] = {0.2989, 0.5866, 0.1145};
            int plane          = 0;
            double best_spread = 0.0;
            for (int p = 0; p < 3; p++) {
                double spread = lum[p] * (mx[p] - mn[p]);
                if (spread > best_spread) {
                    plane       = p;
                    best_spread = spread;
                }
            }
            sort_colors_by_plane(tab + start, size, plane, tmp);
            unsigned int lowersum = tab[start].count, i;
            for (i = 1; i < size - 1 && lowersum < sm / 2; ++i)
                lowersum += tab[start + i].count;
            unsigned int median = i;
            bv[bi].colors       = median;
            bv[bi].sum          = lowersum;
            bv[boxes].ind       = start + median;
            bv[boxes].colors    = size - median;
            bv[boxes].sum       = sm - lowersum;
            ++boxes;
            sort_boxes_by_sum_desc(bv, boxes);
        }
        free(tmp);
        for (unsigned int bi = 0; bi < boxes; bi++) { /* REP_AVERAGE_COLORS */
            for (int p = 0; p < 3; p++) {
                unsigned int s = 0;
                for (unsigned int i = 0; i < bv[bi].colors; i++)
                    s += tab[bv[bi].ind + i].c[p];
                pal[bi * 3 + p] = (unsigned char)(s / bv[bi].colors);
            }
        }
        ncolors = (int)boxes;
    }
    free(tab);
    return ncolors;
}

int oracle_sixel_palette(const uint8_t *rgba, int w, int h, uint8_t *pal_rgb,
                         int *dither_off) {
    unsigned char *rgb = (unsigned char *)malloc((size_t)w * h * 3 + 3);
    for (size_t i = 0; i < (size_t)w * h; i++) memcpy(rgb + i * 3, rgba + i * 4, 3);
    int orig = 0;
    int n    = make_palette(rgb, w, h, pal_rgb, &orig);
    if (dither_off) *dither_off = orig <= n;
    free(rgb);
    return n;
}

/* ---- quantise + diffuse -------------------------------------------------- */
static int nearest(const unsigned char *px, const unsigned char *pal, int ncolors) {
    int best = -1, diff = INT_MAX;
    for (int i = 0; i < ncolors; i++) {
        int r = px[0] - pal[i * 3 + 0], g = px[1] - pal[i * 3 + 1],
            b = px[2] - pal[i * 3 + 2];
        int d = r * r + g * g + b * b; /* complexion == 1 */
        if (d < diff) {
            diff = d;
            best = i;
        }
    }
    return best;
}

static void diffuse_add(unsigned char *data, long pos, int error, int num) {
    int c = data[pos * 3] + error * num / 16;
    if (c < 0) c = 0;
    if (c > 255) c = 255;
    data[pos * 3] = (unsigned char)c;
}

/* trace (may be NULL): the value every pixel had when it was looked up (after the errors of its
 * predecessors had been added), 3 bytes per pixel -- for the per-pixel bound of tests/test_sixel_oracle.py */
static unsigned char *g_trace = NULL;

static void apply_palette(unsigned char *rgb, int w, int h,
                          const unsigned char *pal, int ncolors, int dither,
                          int lookup_mode, unsigned char *index) {
    unsigned short *cache = (unsigned short *)calloc(1 << 15, sizeof(unsigned short));
    if (lookup_mode == 1) {
        for (unsigned int c = 0; c < (1u << 15); c++) {
            unsigned char centre[3] = {
                (unsigned char)((((c >> 10) & 0x1f) << 3) | 4),
                (unsigned char)((((c >> 5) & 0x1f) << 3) | 4),
                (unsigned char)(((c & 0x1f) << 3) | 4)};
            cache[c] = (unsigned short)(nearest(centre, pal, ncolors) + 1);
        }
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            long pos          = (long)y * w + x;
            unsigned char *px = rgb + pos * 3;
            unsigned int hsh  = hash555(px);
            int ci;
            if (cache[hsh])
                ci = cache[hsh] - 1;
            else {
                ci         = nearest(px, pal, ncolors);
                cache[hsh] = (unsigned short)(ci + 1);
            }
            index[pos] = (unsigned char)ci;
            if (g_trace) memcpy(g_trace + pos * 3, px, 3);
            if (!dither) continue;
            for (int n = 0; n < 3; n++) {
                int err = px[n] - pal[ci * 3 + n];
                /* Floyd-Steinberg; note the unguarded left-bottom neighbour:
                 * at x==0 it is the last pixel of the current row. */
                if (x < w - 1 && y < h - 1) {
                    diffuse_add(rgb + n, pos + 1, err, 7);
                    diffuse_add(rgb + n, pos + w - 1, err, 3);
                    diffuse_add(rgb + n, pos + w, err, 5);
                    diffuse_add(rgb + n, pos + w + 1, err, 1);
                }
            }
        }
    }
    free(cache);
}

/* ---- sixel body ---------------------------------------------------------- */
typedef struct {
    char *buf;
    long pos, cap;
    int save_pixel, save_count, active_palette, overflow;
} sout_t;

static void s_putc(sout_t *o, int c) {
    if (o->pos < o->cap)
        o->buf[o->pos++] = (char)c;
    else
        o->overflow = 1;
}
static void s_puts(sout_t *o, const char *s) {
    while (*s) s_putc(o, *s++);
}
static void s_putnum(sout_t *o, int v) {
    char t[16];
    snprintf(t, sizeof t, "%d", v);
    s_puts(o, t);
}

static void put_flash(sout_t *o) {
    while (o->save_count > 255) { /* has_gri_arg_limit */
        s_puts(o, "!255");
        s_putc(o, o->save_pixel);
        o->save_count -= 255;
    }
    if (o->save_count > 3) {
        s_putc(o, '!');
        s_putnum(o, o->save_count);
        s_putc(o, o->save_pixel);
    } else
        for (int n = 0; n < o->save_count; n++) s_putc(o, o->save_pixel);
    o->save_pixel = 0;
    o->save_count = 0;
}

static void put_pixel(sout_t *o, int pix) {
    if (pix < 0 || pix > '?') pix = 0;
    pix += '?';
    if (pix == o->save_pixel)
        o->save_count++;
    else {
        put_flash(o);
        o->save_pixel = pix;
        o->save_count = 1;
    }
}

typedef struct node {
    int pal, sx, mx;
    const unsigned char *map;
    struct node *next;
} node_t;

static void put_node(sout_t *o, int *x, const node_t *np) {
    if (o->active_palette != np->pal) {
        s_putc(o, '#');
        s_putnum(o, np->pal);
        o->active_palette = np->pal;
    }
    for (; *x < np->sx; ++*x) put_pixel(o, 0);
    for (; *x < np->mx; ++*x) put_pixel(o, np->map[*x]);
    put_flash(o);
}

static void encode_body(sout_t *o, const unsigned char *index, int w, int h,
                        const unsigned char *pal, int ncolors) {
    unsigned char *map = (unsigned char *)calloc((size_t)ncolors * w, 1);
    o->active_palette  = -1;
    for (int n = 0; n < ncolors; n++) {
        s_putc(o, '#');
        s_putnum(o, n);
        s_puts(o, ";2;");
        s_putnum(o, (pal[n * 3 + 0] * 100 + 127) / 255);
        s_putc(o, ';');
        s_putnum(o, (pal[n * 3 + 1] * 100 + 127) / 255);
        s_putc(o, ';');
        s_putnum(o, (pal[n * 3 + 2] * 100 + 127) / 255);
    }
    node_t *top = NULL;
    int i       = 0;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int pix = index[(long)y * w + x];
            if (pix < ncolors) map[(long)pix * w + x] |= (unsigned char)(1 << i);
        }
        if (++i < 6 && (y + 1) < h) continue;
        for (int c = 0; c < ncolors; c++) {
            const unsigned char *row = map + (long)c * w;
            for (int sx = 0; sx < w; sx++) {
                if (row[sx] == 0) continue;
                int mx;
                for (mx = sx + 1; mx < w; mx++) {
                    if (row[mx] != 0) continue;
                    int n;
                    for (n = 1; (mx + n) < w; n++)
                        if (row[mx + n] != 0) break;
                    if (n >= 10 || (mx + n) >= w) break;
                    mx = mx + n - 1;
                }
                node_t *np = (node_t *)malloc(sizeof(node_t));
                np->pal    = c;
                np->sx     = sx;
                np->mx     = mx;
                np->map    = row;
                node_t head;
                head.next  = top;
                node_t *tp = &head;
                while (tp->next != NULL) {
                    if (np->sx < tp->next->sx) break;
                    if (np->sx == tp->next->sx && np->mx > tp->next->mx) break;
                    tp = tp->next;
                }
                np->next = tp->next;
                tp->next = np;
                top      = head.next;
                sx       = mx - 1;
            }
        }
        if (y != 5) s_putc(o, '-'); /* DECGNL before every band but the first */
        node_t *np;
        for (int x = 0; (np = top) != NULL;) {
            if (x > np->sx) {
                s_putc(o, '$'); /* DECGCR */
                x = 0;
            }
            put_node(o, &x, np);
            top          = np->next;
            node_t *prev = NULL, *cur = np->next;
            free(np);
            while (cur != NULL) {
                if (cur->sx < x) {
                    prev = cur;
                    cur  = cur->next;
                    continue;
                }
                put_node(o, &x, cur);
                node_t *next = cur->next;
                if (prev)
                    prev->next = next;
                else
                    top = next;
                free(cur);
                cur = next;
            }
        }
        i = 0;
        memset(map, 0, (size_t)ncolors * w);
    }
    free(map);
}

static int round6(int px) { /* src/sixel-canvas.cc:91-94 */
    px += 5;
    return px - px % 6;
}

/* What the two libsixel calls of SixelCanvas::Send produce for an RGBA8888 frame
 * (sixel_dither_initialize + sixel_encode, src/sixel-canvas.cc:137-145): DCS q, raster
 * attributes, palette, bands, ST -- no padding, no cursor strings.  oracle/stub/sixel.h
 * forwards the reference's OWN sixel-canvas.cc to this function (oracle/_ref), which
 * pins the wrapper below -- not this function -- to the reference class. */
long oracle_libsixel_encode(const uint8_t *rgba, int w, int h, int lookup_mode, char *out, long cap) {
    sout_t o;
    memset(&o, 0, sizeof o);
    o.buf = out;
    o.cap = cap;
    unsigned char *rgb = (unsigned char *)malloc((size_t)w * h * 3 + 3);
    for (size_t i = 0; i < (size_t)w * h; i++) memcpy(rgb + i * 3, rgba + i * 4, 3);
    unsigned char pal[256 * 3];
    int orig    = 0;
    int ncolors = make_palette(rgb, w, h, pal, &orig);
    int dither  = !(orig <= ncolors);
    unsigned char *index = (unsigned char *)malloc((size_t)w * h);
    apply_palette(rgb, w, h, pal, ncolors, dither, lookup_mode, index);
    s_puts(&o, "\033Pq");
    s_puts(&o, "\"1;1;");
    s_putnum(&o, w);
    s_putc(&o, ';');
    s_putnum(&o, h);
    encode_body(&o, index, w, h, pal, ncolors);
    s_puts(&o, "\033\\");
    free(index);
    free(rgb);
    return o.overflow ? -1 : o.pos;
}

/* Quantisation alone, with a trace: palette (ncolors * 3 bytes), the index of every pixel and the
 * value it had when it was looked up.  Returns ncolors; *dithered says whether errors were diffused. */
int oracle_sixel_quantize_trace(const uint8_t *rgba, int w, int h, int lookup_mode, uint8_t *pal_rgb,
                                uint8_t *index, uint8_t *looked_up_rgb, int *dithered) {
    unsigned char *rgb = (unsigned char *)malloc((size_t)w * h * 3 + 3);
    for (size_t i = 0; i < (size_t)w * h; i++) memcpy(rgb + i * 3, rgba + i * 4, 3);
    int orig    = 0;
    int ncolors = make_palette(rgb, w, h, pal_rgb, &orig);
    int dither  = !(orig <= ncolors);
    g_trace     = looked_up_rgb;
    apply_palette(rgb, w, h, pal_rgb, ncolors, dither, lookup_mode, index);
    g_trace = NULL;
    if (dithered) *dithered = dither;
    free(rgb);
    return ncolors;
}

long oracle_sixel_encode(const uint8_t *fb, int w, int h, int has_getter,
                         uint32_t bg, uint32_t pattern, int pw, int ph,
                         int broken_cursor, int lookup_mode, char *out,
                         long cap) {
    const int ph6 = round6(h);
    /* src/sixel-canvas.cc:111-120: transparent frame of the padded height,
     * background/pattern blended into rows >= h only, original copied over */
    uint8_t *padded = (uint8_t *)calloc((size_t)w * ph6, 4);
    oracle_alpha_compose(padded, w, ph6, has_getter, bg, pattern, pw, ph, h);
    memcpy(padded, fb, (size_t)w * h * 4);

    sout_t o;
    memset(&o, 0, sizeof o);
    o.buf = out;
    o.cap = cap;
    /* src/sixel-canvas.cc:66-79 */
    s_puts(&o, broken_cursor ? "\033[80l\033[?7730l\033[?8452h"
                             : "\033[80h\033[?7730h\033[?8452l");
    unsigned char *rgb = (unsigned char *)malloc((size_t)w * ph6 * 3 + 3);
    for (size_t i = 0; i < (size_t)w * ph6; i++) memcpy(rgb + i * 3, padded + i * 4, 3);
    unsigned char pal[256 * 3];
    int orig    = 0;
    int ncolors = make_palette(rgb, w, ph6, pal, &orig);
    int dither  = !(orig <= ncolors);
    unsigned char *index = (unsigned char *)malloc((size_t)w * ph6);
    apply_palette(rgb, w, ph6, pal, ncolors, dither, lookup_mode, index);
    s_puts(&o, "\033Pq");
    s_puts(&o, "\"1;1;");
    s_putnum(&o, w);
    s_putc(&o, ';');
    s_putnum(&o, ph6);
    encode_body(&o, index, w, ph6, pal, ncolors);
    s_puts(&o, "\033\\");
    s_puts(&o, broken_cursor ? "\n" : "\r");
    free(index);
    free(rgb);
    free(padded);
    return o.overflow ? -1 : o.pos;
}

/* ---- independent decoder ------------------------------------------------- */
int oracle_sixel_decode(const char *data, long len, uint8_t *rgba_out, int cap_w,
                        int cap_h, int *w_out, int *h_out, int *ncolors_out) {
    long p = 0;
    /* find DCS: ESC P ... q */
    while (p + 1 < len && !(data[p] == '\033' && data[p + 1] == 'P')) p++;
    if (p + 1 >= len) return -1;
    p += 2;
    while (p < len && data[p] != 'q') p++;
    if (p >= len) return -2;
    p++;
    unsigned char pal[1024][3];
    unsigned char defined[1024];
    memset(defined, 0, sizeof defined);
    memset(pal, 0, sizeof pal);
    int x = 0, band = 0, color = 0, maxx = 0, maxy = 0, rw = 0, rh = 0, repeat = 1;
    int nc = 0;
    while (p < len) {
        unsigned char c = (unsigned char)data[p];
        if (c == '\033') break; /* ST */
        if (c == '"') {
            int v[4] = {0, 0, 0, 0}, k = 0;
            p++;
            while (p < len && k < 4) {
                while (p < len && data[p] >= '0' && data[p] <= '9')
                    v[k] = v[k] * 10 + (data[p++] - '0');
                k++;
                if (p < len && data[p] == ';')
                    p++;
                else
                    break;
            }
            rw = v[2];
            rh = v[3];
            continue;
        }
        if (c == '#') {
            int v[5] = {0, 0, 0, 0, 0}, k = 0;
            p++;
            while (p < len && k < 5) {
                while (p < len && data[p] >= '0' && data[p] <= '9')
                    v[k] = v[k] * 10 + (data[p++] - '0');
                k++;
                if (p < len && data[p] == ';')
                    p++;
                else
                    break;
            }
            if (v[0] < 0 || v[0] >= 1024) return -3;
            color = v[0];
            if (k == 5) {
                if (v[1] != 2) return -4; /* only RGB percent */
                for (int i = 0; i < 3; i++) {
                    if (v[2 + i] > 100) return -5;
                    pal[color][i] = (unsigned char)((v[2 + i] * 255 + 50) / 100);
                }
                if (!defined[color]) {
                    defined[color] = 1;
                    nc++;
                }
            }
            continue;
        }
        if (c == '!') {
            int n = 0;
            p++;
            while (p < len && data[p] >= '0' && data[p] <= '9')
                n = n * 10 + (data[p++] - '0');
            repeat = n;
            continue;
        }
        if (c == '$') {
            x = 0;
            p++;
            continue;
        }
        if (c == '-') {
            x = 0;
            band++;
            p++;
            continue;
        }
        if (c >= '?' && c <= '~') {
            int bits = c - '?';
            for (int r = 0; r < repeat; r++, x++) {
                if (x >= cap_w) return -6;
                for (int b = 0; b < 6; b++) {
                    if (!(bits & (1 << b))) continue;
                    int y = band * 6 + b;
                    if (y >= cap_h) return -7;
                    uint8_t *o = rgba_out + ((size_t)y * cap_w + x) * 4;
                    o[0]       = pal[color][0];
                    o[1]       = pal[color][1];
                    o[2]       = pal[color][2];
                    o[3]       = 255;
                    if (y + 1 > maxy) maxy = y + 1;
                }
            }
            if (x > maxx) maxx = x;
            repeat = 1;
            p++;
            continue;
        }
        p++; /* ignore anything else (newlines etc.) */
    }
    *w_out = rw ? rw : maxx;
    *h_out = rh ? rh : maxy;
    if (ncolors_out) *ncolors_out = nc;
    return 0;
}
