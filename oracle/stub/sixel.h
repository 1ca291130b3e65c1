/* oracle/stub/sixel.h -- TEST INFRASTRUCTURE ONLY.
 *
 * libsixel is neither vendored by hzeller/timg nor installed in this image, so the
 * reference's src/sixel-canvas.cc cannot be compiled as it is.  This header declares
 * exactly the slice of libsixel's public API that file uses (src/sixel-canvas.cc:17,
 * :134-148) and oracle/stub/sixel_stub.c implements it by forwarding the pixel work to
 * this repository's restatement (oracle_libsixel_encode).  With it the REAL
 * timg::SixelCanvas -- padding to 6-row bands, background for the pad rows only, cursor
 * strings, prefix handling, one future per Send on the encoder pool -- compiles from the
 * sources where they lie and becomes the checker for the wrapper around the encoder
 * (SURVEY 8a-12).  It does NOT pin the encoder itself: libsixel's bytes stay unpinned.
 *
 * Names, argument order and constant values follow libsixel's sixel.h (1.8.x). */
#ifndef TIMG_ORACLE_STUB_SIXEL_H
#define TIMG_ORACLE_STUB_SIXEL_H
#ifdef __cplusplus
extern "C" {
#endif

typedef int SIXELSTATUS;
/* (src/timg-print-version.cc:121 prints it: a build of timg itself against this stub says what it is) */
#define LIBSIXEL_VERSION "none: oracle/stub/sixel.h over the restatement"
#define SIXEL_OK 0x0000
#define SIXEL_FALSE 0x1000

#define SIXEL_PIXELFORMAT_RGBA8888 0x11 /* (SIXEL_FORMATTYPE_COLOR | 0x11) */
#define SIXEL_LARGE_LUM 0x2             /* method for finding the largest dimension */
#define SIXEL_REP_AVERAGE_COLORS 0x2    /* method for choosing a box's colour */
#define SIXEL_QUALITY_AUTO 0x0

typedef struct sixel_allocator sixel_allocator_t;
typedef struct sixel_output sixel_output_t;
typedef struct sixel_dither sixel_dither_t;
typedef int (*sixel_write_function)(char *data, int size, void *priv);

SIXELSTATUS sixel_output_new(sixel_output_t **output, sixel_write_function fn_write, void *priv,
                             sixel_allocator_t *allocator);
void sixel_output_destroy(sixel_output_t *output);
SIXELSTATUS sixel_dither_new(sixel_dither_t **ppdither, int ncolors, sixel_allocator_t *allocator);
void sixel_dither_destroy(sixel_dither_t *dither);
SIXELSTATUS sixel_dither_initialize(sixel_dither_t *dither, unsigned char *data, int width, int height,
                                    int pixelformat, int method_for_largest, int method_for_rep,
                                    int quality_mode);
SIXELSTATUS sixel_encode(unsigned char *pixels, int width, int height, int depth, sixel_dither_t *dither,
                         sixel_output_t *context);

/* not libsixel: which lookup the forwarded encoder uses (0: libsixel-like first-hit cache,
 * 1: nearest to the 15-bit cell's centre -- what the HIP path implements).  Process-wide. */
void timg_stub_sixel_set_lookup_mode(int mode);

#ifdef __cplusplus
}
#endif
#endif
