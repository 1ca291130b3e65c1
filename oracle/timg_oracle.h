/* oracle/timg_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's rendering hot path (hzeller/timg):
 * scaler resample, alpha compose, half/quarter block encode, sixel encode.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product (timg_amd/) never
 * links, loads or calls anything in oracle/.
 *
 * Parity status (see oracle/README.md):
 *   scale / blend / block encode : pinned against the real reference compiled
 *                                  from /root/reference (oracle/_ref) and the
 *                                  golden vectors in tests/golden/.
 *   sixel                        : libsixel is absent and un-vendored
 *                                  => "parity unpinned" (restates the published
 *                                  algorithm; validated by decode round trip).
 */
#ifndef TIMG_ORACLE_H
#define TIMG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* in_fmt: 0 = RGBA (ImageScaler::ColorFmt::kRGBA), 1 = BGRA in memory
 * (kRGB32).  filter: 0 = reference default (Mitchell down / box-trapezoid up /
 * point at 1:1; src/image-scaler.cc:32,85-91), 2 = triangle on both axes.
 * Returns 0 on success. */
int oracle_scale(const uint8_t *src, int sw, int sh, int in_fmt, uint8_t *dst,
                 int dw, int dh, int filter);

/* Introspection for tests: fills the resampling decisions of a (sw,sh)->(dw,dh)
 * plan: info[0]=vertical_first, [1]=h widest, [2]=v is_gather(0 scatter,1 up,
 * 2 down), [3]=v widest, [4]=h filter enum, [5]=v filter enum. */
int oracle_scale_plan_info(int sw, int sh, int dw, int dh, int filter,
                           int info[6]);

/* Framebuffer::AlphaComposeBackground (src/framebuffer.cc:108-150), in place.
 * has_getter==0 models a null bgcolor_query ("-b none").  Colors are packed
 * r | g<<8 | b<<16 | a<<24 (memory order r,g,b,a).  Returns how many times the
 * background getter would have been called (0 or 1). */
int oracle_alpha_compose(uint8_t *fb, int w, int h, int has_getter,
                         uint32_t bg, uint32_t pattern, int pw, int ph,
                         int start_row);

/* UnicodeBlockCanvas (src/unicode-block-canvas.cc).  A canvas object keeps
 * the frame-diff backing store across Sends exactly like the reference. */
typedef struct oracle_block_canvas oracle_block_canvas;
oracle_block_canvas *oracle_block_canvas_new(int quarter, int upper_block,
                                             int color256);
void oracle_block_canvas_free(oracle_block_canvas *c);
/* Appends what Send(x, dy, fb) would hand to the write sequencer (including a
 * pending cursor-up prefix for dy<0) to out; returns bytes appended or -1 if
 * cap is too small.  */
long oracle_block_canvas_send(oracle_block_canvas *c, int x, int dy,
                              const uint8_t *fb, int w, int h, char *out,
                              long cap);
/* One-shot: fresh canvas, Send(x, 0, fb). */
long oracle_block_encode(const uint8_t *fb, int w, int h, int quarter,
                         int upper_block, int color256, int x, char *out,
                         long cap);
/* Worst-case output size the reference allocates (RequestBuffers,
 * src/unicode-block-canvas.cc:405-424). */
size_t oracle_block_max_bytes(int w, int h);

uint8_t oracle_as_256_term_color(uint32_t c); /* src/framebuffer.h:37-52 */

/* --auto-crop / --crop-border (parity unpinned: GraphicsMagick trim(), see oracle/autocrop.c):
 * {x, y, w, h} of what remains, in source coordinates; w = h = 0: nothing but border. */
void oracle_autocrop_bbox(const uint8_t *rgba, int w, int h, int stride, int crop_border, int xywh[4]);

/* ---- sixel (parity unpinned; see oracle/README.md) ------------------------
 * Restates SixelCanvas::Send's call contract (src/sixel-canvas.cc:100-155):
 * pad to a multiple of 6 rows, blend only the pad rows, 256-colour adaptive
 * palette (median cut, LARGE_LUM, REP_AVERAGE_COLORS, QUALITY_AUTO), Floyd-
 * Steinberg diffusion, band/colour RLE.  lookup_mode: 0 = libsixel-like lossy
 * 15-bit lookup cache in raster order, 1 = exact nearest colour per pixel
 * (what the HIP path implements bit-exactly).
 * Writes cursor prefix + DCS ... ST + suffix.  Returns bytes or -1. */
long oracle_sixel_encode(const uint8_t *fb, int w, int h, int has_getter,
                         uint32_t bg, uint32_t pattern, int pw, int ph,
                         int broken_cursor, int lookup_mode, char *out,
                         long cap);
/* The libsixel part alone (sixel_dither_initialize + sixel_encode of an RGBA8888 frame whose
 * height is a multiple of 6): DCS q ... ST.  Used by oracle/stub/sixel.h. */
long oracle_libsixel_encode(const uint8_t *rgba, int w, int h, int lookup_mode, char *out, long cap);
/* Quantisation with a trace (not thread safe): palette, index per pixel, and the value every pixel
 * had when it was looked up (3 bytes per pixel).  Returns ncolors. */
int oracle_sixel_quantize_trace(const uint8_t *rgba, int w, int h, int lookup_mode, uint8_t *pal_rgb,
                                uint8_t *index, uint8_t *looked_up_rgb, int *dithered);
/* Palette only (<=256 entries r,g,b); returns ncolors. dither_off set to 1 if
 * the image had <= 256 distinct 15-bit colours. */
int oracle_sixel_palette(const uint8_t *rgba, int w, int h, uint8_t *pal_rgb,
                         int *dither_off);
/* libsixel sorts with qsort(): the order among equal keys is the C library's.  The restatement pins it (stable,
 * mode 0); mode bits turn the order among equal keys around -- 1: colours equal in the split plane, 2: boxes of equal
 * weight -- so that a test can MEASURE what the pin is worth (process-wide, not thread safe). */
void oracle_sixel_set_tie_order(int mode);
/* Independent sixel decoder: parses DCS q ... ST into RGBA (undrawn pixels
 * stay 0,0,0,0).  Returns 0 on success; w/h are the raster size found. */
int oracle_sixel_decode(const char *data, long len, uint8_t *rgba_out,
                        int cap_w, int cap_h, int *w, int *h, int *ncolors);

/* ---- graphics protocols at --compress=0 (oracle/png.c) --------------------------
 * png::Encode (src/timg-png.cc:91-153) with libdeflate level 0 (stored blocks),
 * EncodeBase64 (src/timg-base64.h), the bytes KittyGraphicsCanvas::Send
 * (src/kitty-canvas.cc:126-221, no tmux) and ITerm2GraphicsCanvas::Send
 * (src/iterm2-canvas.cc:40-75) put behind their cursor prefix.  Pinned against the
 * real reference + libdeflate through oracle/_ref (tests/test_png_oracle.py). */
size_t oracle_png_bytes(int w, int h, int with_alpha); /* exact size of the level-0 PNG */
long oracle_png_encode(const uint8_t *fb, int w, int h, int with_alpha, char *out, long cap);
long oracle_base64(const uint8_t *in, long n, char *out);
size_t oracle_kitty_max_bytes(int w, int h);
long oracle_kitty_encode(const uint8_t *fb, int w, int h, int with_alpha, uint32_t id,
                         char *scratch, char *out, long cap);
long oracle_iterm2_encode(const uint8_t *fb, int w, int h, int with_alpha, char *scratch,
                          char *out, long cap);
/* checksum building blocks (the parallel forms are what a device implementation needs) */
uint32_t oracle_crc32(uint32_t crc, const uint8_t *p, size_t n);
uint32_t oracle_adler32(const uint8_t *p, size_t n);
uint32_t oracle_crc32_multmodp(uint32_t a, uint32_t b);
uint32_t oracle_crc32_xpow_bytes(uint64_t n_bytes);
uint32_t oracle_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);

#ifdef __cplusplus
}
#endif
#endif /* TIMG_ORACLE_H */
