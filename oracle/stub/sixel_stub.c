/* oracle/stub/sixel_stub.c -- TEST INFRASTRUCTURE ONLY: see sixel.h in this directory. */
#include "sixel.h"

#include <stdlib.h>

#include "../timg_oracle.h"

struct sixel_output {
    sixel_write_function write;
    void *priv;
};
struct sixel_dither {
    int ncolors;
    int initialised_w, initialised_h;
};

/* lookup rule of the restatement behind sixel_encode: 0 libsixel's own first-hit cache (default), 1 the cell-centre
 * table the device implements.  Set by the test drivers, or -- for a whole timg built over this stub
 * (integration/Makefile), which has no call for it -- by TIMG_STUB_SIXEL_LOOKUP in the environment. */
static int g_lookup_mode = -1;
void timg_stub_sixel_set_lookup_mode(int mode) { g_lookup_mode = mode; }
static int lookup_mode(void) {
    if (g_lookup_mode < 0) {
        const char *e = getenv("TIMG_STUB_SIXEL_LOOKUP");
        g_lookup_mode = (e && e[0] == '1') ? 1 : 0;
    }
    return g_lookup_mode;
}

SIXELSTATUS sixel_output_new(sixel_output_t **output, sixel_write_function fn_write, void *priv,
                             sixel_allocator_t *allocator) {
    (void)allocator;
    *output = (sixel_output_t *)calloc(1, sizeof(**output));
    if (!*output) return SIXEL_FALSE;
    (*output)->write = fn_write;
    (*output)->priv  = priv;
    return SIXEL_OK;
}
void sixel_output_destroy(sixel_output_t *output) { free(output); }

SIXELSTATUS sixel_dither_new(sixel_dither_t **ppdither, int ncolors, sixel_allocator_t *allocator) {
    (void)allocator;
    *ppdither = (sixel_dither_t *)calloc(1, sizeof(**ppdither));
    if (!*ppdither) return SIXEL_FALSE;
    (*ppdither)->ncolors = ncolors;
    return SIXEL_OK;
}
void sixel_dither_destroy(sixel_dither_t *dither) { free(dither); }

/* The restatement derives the palette inside its encode step from the same pixels; the call
 * site passes the same frame to both calls (src/sixel-canvas.cc:139-145), which is checked. */
SIXELSTATUS sixel_dither_initialize(sixel_dither_t *dither, unsigned char *data, int width, int height,
                                    int pixelformat, int method_for_largest, int method_for_rep,
                                    int quality_mode) {
    (void)data;
    if (!dither || dither->ncolors != 256 || pixelformat != SIXEL_PIXELFORMAT_RGBA8888 ||
        method_for_largest != SIXEL_LARGE_LUM || method_for_rep != SIXEL_REP_AVERAGE_COLORS ||
        quality_mode != SIXEL_QUALITY_AUTO)
        return SIXEL_FALSE; /* the restatement only covers the reference's call */
    dither->initialised_w = width;
    dither->initialised_h = height;
    return SIXEL_OK;
}

SIXELSTATUS sixel_encode(unsigned char *pixels, int width, int height, int depth, sixel_dither_t *dither,
                         sixel_output_t *context) {
    (void)depth;
    if (!dither || !context || dither->initialised_w != width || dither->initialised_h != height)
        return SIXEL_FALSE;
    const long cap = 4096 + (long)width * height * 8;
    char *buf      = (char *)malloc((size_t)cap);
    if (!buf) return SIXEL_FALSE;
    const long n = oracle_libsixel_encode(pixels, width, height, lookup_mode(), buf, cap);
    if (n < 0) {
        free(buf);
        return SIXEL_FALSE;
    }
    /* libsixel hands its output over in pieces; so does this */
    for (long at = 0; at < n; at += 16384) {
        const long piece = n - at < 16384 ? n - at : 16384;
        context->write(buf + at, (int)piece, context->priv);
    }
    free(buf);
    return SIXEL_OK;
}
