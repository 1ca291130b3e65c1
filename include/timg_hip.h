/* include/timg_hip.h -- C-ABI of libtimg_hip.so, the MI355X (gfx950) twin of
 * hzeller/timg's per-pixel rendering hot path.
 *
 * This is the drop-in boundary: plain C types, opaque handles, caller-owned
 * buffers, no exceptions, no C++ or torch types.  Every entry point returns
 * 0 on success and a negative timg_hip_status on failure (the reference's
 * "return nullptr / false and let the caller fall back" convention,
 * SURVEY.md 8b); timg_hip_last_error() gives the text.  The library fails
 * loudly when no HIP device is usable -- there is no CPU fallback in here.
 *
 * Each function names the reference interface it replaces (paths relative to
 * the hzeller/timg tree).  INTEGRATION.md shows the C++ twins
 * (HipImageScaler, HipUnicodeBlockCanvas, HipSixelCanvas) a maintainer adds on
 * the reference side to bind these.
 *
 * Pixel format everywhere: RGBA8, memory order r,g,b,a (timg::rgba_t,
 * src/framebuffer.h:26-28), rows `stride` bytes apart (stride 0 = 4*width).
 * Colours passed by value are packed r | g<<8 | b<<16 | a<<24.
 * `stream` is a hipStream_t (NULL = the context's own stream); work is
 * enqueued on it and, unless stated otherwise, NOT synchronised: device
 * pointers are valid to consume on the same stream, host pointers force a
 * synchronisation before return.  The context's own stream is NON-BLOCKING: it
 * is not ordered against the NULL stream or any other stream.  Device buffers
 * that another stream fills or clears (a framework's allocator zeroing `dst`,
 * a decoder writing `src`) must be finished -- or that stream passed here as
 * `stream` -- before the call; nothing in this library waits for foreign
 * streams.
 */
#ifndef TIMG_HIP_H
#define TIMG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    TIMG_HIP_OK          = 0,
    TIMG_HIP_ERR_ARG     = -1, /* bad argument */
    TIMG_HIP_ERR_DEVICE  = -2, /* HIP runtime error / no device */
    TIMG_HIP_ERR_NOMEM   = -3,
    TIMG_HIP_ERR_SMALL   = -4, /* caller's output capacity too small */
    TIMG_HIP_ERR_UNSUPP  = -5, /* geometry outside what the kernels cover */
} timg_hip_status;

typedef struct timg_hip_ctx timg_hip_ctx;
typedef struct timg_hip_scaler timg_hip_scaler;

/* ---- context ----------------------------------------------------------- */
int timg_hip_init(int device, timg_hip_ctx **out);
void timg_hip_destroy(timg_hip_ctx *ctx);
const char *timg_hip_last_error(const timg_hip_ctx *ctx); /* ctx may be NULL */
/* Library/ABI version (major<<16 | minor). */
int timg_hip_version(void);

/* Buffers owned by the library (hipMalloc / hipHostMalloc). Convenience for
 * hosts that have no HIP binding of their own (Python tests, the C++ twins). */
int timg_hip_malloc(timg_hip_ctx *ctx, size_t bytes, void **dev_ptr);
int timg_hip_free(timg_hip_ctx *ctx, void *dev_ptr);
int timg_hip_memcpy_h2d(timg_hip_ctx *ctx, void *dst, const void *src, size_t n, void *stream);
int timg_hip_memcpy_d2h(timg_hip_ctx *ctx, void *dst, const void *src, size_t n, void *stream);
int timg_hip_sync(timg_hip_ctx *ctx, void *stream);
int timg_hip_memcpy_d2d(timg_hip_ctx *ctx, void *dst, const void *src, size_t n, void *stream);

/* ---- streams for a partitioned pipeline (round 6; not a reference interface) ---------------------------------------
 * Every entry point takes the caller's stream.  A caller that has no HIP binding of its own -- or wants what plain HIP
 * streams do not give it -- gets streams here:
 *   reserved_cus_per_xcd > 0: the stream's kernels run on all CUs of the device EXCEPT that many of each of its 8 XCDs
 *     (hipExtStreamCreateWithCUMask; rounded down to a multiple of 4 = one CU of each of an XCD's four shader engines:
 *     a mask that takes unequal numbers of CUs from the engines slows a launch down to its smallest engine).  The use:
 *     the scale call of batch k + 1 on such a stream, the sixel chain of batch k on an unrestricted one -- the chain's
 *     latency-bound kernels (histogram, median cut, table, diffusion) then find free CUs while the scale kernel, which
 *     otherwise holds every CU's registers and LDS until its last workgroup retires, runs beside them
 *     (profiles/r6/partitioned_streams.txt: 1.32 -> 1.16 ms per 64-frame step with 12 CUs of every XCD kept free).
 *   high_priority != 0: created with the device's greatest stream priority.
 * The context owns the stream: timg_hip_stream_destroy (after the work on it has finished) or timg_hip_destroy ends it.
 * Ordering BETWEEN streams is the caller's (events of its own binding, or timg_hip_stream_wait_stream below, which
 * makes `waiter`'s later work wait for what `signaller` holds now). */
int timg_hip_stream_create(timg_hip_ctx *ctx, int reserved_cus_per_xcd, int high_priority, void **stream);
int timg_hip_stream_destroy(timg_hip_ctx *ctx, void *stream);
int timg_hip_stream_wait_stream(timg_hip_ctx *ctx, void *waiter, void *signaller);

/* ---- synthetic frames (the measurement plan's inputs) ---------------
 * Not a reference interface: hzeller/timg has no frame generator.  These are the S-noise /
 * S-photo / S-alpha RGBA8 frames the BASELINE configurations are defined on, generated in
 * device memory (n_frames frames frame_stride bytes apart, 0 = packed; frame index
 * first_frame + i enters the hash) so that a 2 GB batch never crosses PCIe.  Every byte is an
 * integer function of (kind, seed, frame, x, y); timg_amd/synth.py: hash_frame is the same
 * function on the host, so parity tests run on the benchmark's own frames.  HipRawRGBASource
 * (timg_amd/twins) serves them to the reference's renderer under the name
 * "synth:<kind>:<w>x<h>:<seed>[:<frame>]". */
#define TIMG_HIP_SYNTH_NOISE 0
#define TIMG_HIP_SYNTH_PHOTO 1
#define TIMG_HIP_SYNTH_ALPHA 2
int timg_hip_synth_frames(timg_hip_ctx *ctx, int kind, int w, int h, uint32_t seed, int first_frame,
                          int n_frames, uint8_t *dst, size_t frame_stride, int dst_on_device, void *stream);

/* ---- scaler: timg::ImageScaler ------------------------------------------
 * Replaces ImageScaler::Create (src/image-scaler.h:33-35, src/image-scaler.cc:
 * 101-116) and ImageScaler::Scale (src/image-scaler.h:39; STB back-end
 * src/image-scaler.cc:83-92).  Creating a scaler builds the resampling plan
 * (stb_image_resize2-compatible coefficient tables) once on the host and
 * uploads it; Scale is then pure device work.
 * Threads: calls that use a context's scratch memory (host-side buffers, the
 * canvas encoders) are serialised by the library, so loader threads may create
 * and use scalers concurrently, as they do in the reference
 * (src/timg.cc:948-968); everything on one context runs on its one stream.  One
 * timg_hip_scaler carries per-launch device state of its own: use a scaler
 * from one thread at a time (the reference creates one per image as well). */
#define TIMG_HIP_FMT_RGBA 0 /* ImageScaler::ColorFmt::kRGBA */
#define TIMG_HIP_FMT_BGRA 1 /* ImageScaler::ColorFmt::kRGB32 (b,g,r,a in memory) */
#define TIMG_HIP_FILTER_STB_DEFAULT 0 /* bit-exact with the STB back-end */
#define TIMG_HIP_FILTER_TRIANGLE    2 /* bilinear weights, same machinery */

int timg_hip_scaler_create(timg_hip_ctx *ctx, int in_w, int in_h, int in_fmt,
                           int out_w, int out_h, int filter,
                           timg_hip_scaler **out);
void timg_hip_scaler_destroy(timg_hip_scaler *s);

/* Background description for the fused Framebuffer::AlphaComposeBackground
 * (src/framebuffer.h:103-106, src/framebuffer.cc:108-150).  enabled==0 models
 * a null bgcolor_query ("-b none"); bg alpha 0 means "do not blend"
 * (framebuffer.cc:120-121); pattern alpha 0 / pw<=0 / ph<=0 / pattern==bg
 * selects the solid fast path (framebuffer.cc:124-132). */
typedef struct {
    int enabled;
    uint32_t bg;      /* what bgcolor_getter() returns */
    uint32_t pattern; /* DisplayOptions::bg_pattern_color */
    int pattern_w, pattern_h;
    int start_row;
} timg_hip_blend;

/* Scale n_frames frames of the scaler's geometry, optionally followed by the
 * fused alpha compose.  src/dst may each live on the device (`*_on_device`)
 * or on the host (staged through pinned memory; then the call synchronises).
 * Frame i starts at src + i*src_frame_stride (0 = in_h*src_stride).
 * any_transparent (optional, host int[n_frames]) receives whether the scaled
 * frame held a pixel with alpha<255 at/after blend.start_row before blending
 * -- the laziness condition of AlphaComposeBackground (framebuffer.cc:113-117);
 * asking for it synchronises.
 *
 * Concurrency: calls on ONE scaler from several host threads are serialised while they are being launched (the
 * scaler's tile bookkeeping), and their kernels may then run side by side on different streams with correct results;
 * they share one table of finished tiles per slot, though, so each may redo tiles of the other -- a caller that
 * wants several calls of one geometry in flight at full speed uses one scaler per caller (the twins' pool does:
 * timg_amd/twins/hip-context.h HipScalerAcquire). */
int timg_hip_scale_blend(timg_hip_ctx *ctx, timg_hip_scaler *s,
                         const uint8_t *src, int src_stride,
                         size_t src_frame_stride, int src_on_device,
                         uint8_t *dst, int dst_stride, size_t dst_frame_stride,
                         int dst_on_device, int n_frames,
                         const timg_hip_blend *blend_or_null,
                         int *any_transparent_or_null, void *stream);

/* Force a kernel family for A/B measurements and parity tests:
 * 0 = auto, 1 = generic gather kernel, 2 = streaming kernel (fails with
 * TIMG_HIP_ERR_UNSUPP when the plan is outside its coverage), 3 / 4 = streaming
 * kernel that skips its opaque / opaque+premultiplied fast passes. */
int timg_hip_scaler_set_kernel(timg_hip_scaler *s, int which);
/* Introspection: info[0]=vertical_first [1]=h_widest [2]=v_is_gather
 * [3]=v_widest [4]=h_filter [5]=v_filter [6]=bit 0 streaming kernel applicable,
 * bit 1 its matrix-core variant serves the plan, bit 2 ... with a fifth live
 * output row (the overflow row), bit 3 a horizontal-first plan served by the kernel
 * that carries two output columns per lane pair [7]=max active output rows per input row. */
int timg_hip_scaler_info(const timg_hip_scaler *s, int info[8]);
/* Algorithmic HBM bytes of one frame (read source once + write result once,
 * SURVEY.md 8d). */
size_t timg_hip_scaler_algorithmic_bytes(const timg_hip_scaler *s);

/* ---- standalone alpha compose -------------------------------------------
 * Framebuffer::AlphaComposeBackground in place on n_frames frames
 * (src/framebuffer.cc:108-150; used for the sixel pad rows by
 * src/sixel-canvas.cc:115-118). */
int timg_hip_alpha_compose(timg_hip_ctx *ctx, uint8_t *fb, int w, int h,
                           int stride, size_t frame_stride, int on_device,
                           int n_frames, const timg_hip_blend *blend,
                           int *any_transparent_or_null, void *stream);

/* ---- auto-crop bounding box ----------------------------------------------
 * The reduction behind --auto-crop / --crop-border (GraphicsMagick img.crop() + img.trim() in
 * src/graphics-magick-source.cc:231-241, applied BEFORE scaling; "parity unpinned", SURVEY.md
 * a6): crop_border pixels are removed on every side first; then, at fuzz 0, the left and top
 * edges move in while pixels equal the top-left corner pixel, the right edge while they equal
 * the top-right corner, the bottom edge while they equal the bottom-left corner (trim()'s
 * published algorithm).  out_xywh is a host int[4] per frame {x, y, w, h} in source
 * coordinates; w = h = 0: nothing but border.  To apply it, create the scaler for w x h and
 * pass src + y * stride + x * 4 with the same stride: the crop costs no copy. */
int timg_hip_autocrop_bbox(timg_hip_ctx *ctx, const uint8_t *src, int w, int h,
                           int stride, size_t frame_stride, int on_device,
                           int n_frames, int crop_border, int *out_xywh,
                           void *stream);

/* ---- block canvas: timg::UnicodeBlockCanvas ------------------------------
 * Replaces the pixel work of UnicodeBlockCanvas::Send
 * (src/unicode-block-canvas.cc:323-403: FindBestGlyph :163-227,
 * AppendDoubleRow :231-321).  Produces, per frame, exactly the bytes Send
 * appends after its cursor prefix for a first Send at indent x (no frame-diff:
 * emit_difference is false in grid mode and for a first frame, :344-346).
 * The prefix itself (TerminalCanvas::AppendPrefixToBuffer,
 * src/terminal-canvas.cc:58-64) stays host-side. */
#define TIMG_HIP_BLOCK_QUARTER   1 /* -p quarter, else -p half */
#define TIMG_HIP_BLOCK_UPPER     2 /* TIMG_USE_UPPER_BLOCK */
#define TIMG_HIP_BLOCK_COLOR256  4 /* --color8 */

/* Worst-case bytes of one frame (RequestBuffers, :405-424). */
size_t timg_hip_block_max_bytes(int w, int h);

/* fb: n_frames frames (device or host).  out: n_frames slots of out_cap bytes
 * each (device or host); out_len: host size_t[n_frames].  x_indent is Send's
 * `x` argument in pixels (:334 halves it for quarter blocks). Synchronises. */
int timg_hip_block_encode(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h,
                          int stride, size_t frame_stride, int fb_on_device,
                          int n_frames, int flags, int x_indent, char *out,
                          size_t out_cap, int out_on_device, size_t *out_len,
                          void *stream);

/* The same for one row of a --grid: frame i is the image of grid column i, Sent at
 * x = x_indents[i] pixels (MultiColumnRenderer, src/renderer.cc:81-189, calls Send once per
 * image with its column's x).  x_indents: host int[n_frames]. */
int timg_hip_block_encode_grid(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h,
                               int stride, size_t frame_stride, int fb_on_device,
                               int n_frames, int flags, const int *x_indents, char *out,
                               size_t out_cap, int out_on_device, size_t *out_len,
                               void *stream);

/* Stateful canvas = one UnicodeBlockCanvas object: keeps what the reference
 * keeps between Sends (last height / indent and the backing store of the
 * previous frame, src/unicode-block-canvas.h:66-79) on the device, so that
 * animations get the frame-difference encoding of Send (:343-346: cells equal
 * to the previous frame are skipped, :244-247, and turn into cursor moves,
 * :249-263, :313-315, :397-399).  `flags` as above. */
typedef struct timg_hip_block_canvas timg_hip_block_canvas;
int timg_hip_block_canvas_create(timg_hip_ctx *ctx, int flags, timg_hip_block_canvas **out);
void timg_hip_block_canvas_destroy(timg_hip_block_canvas *c);
/* One Send(x, dy, fb): writes the bytes Send appends after its cursor prefix
 * into the host buffer out (capacity out_cap >= timg_hip_block_max_bytes) and
 * their count into *out_len (0 = the reference hands an empty buffer to the
 * sequencer, :390-395).  fb may be host or device memory.  Synchronises. */
int timg_hip_block_canvas_send(timg_hip_block_canvas *c, int x, int dy, const uint8_t *fb,
                               int w, int h, int stride, int fb_on_device, char *out,
                               size_t out_cap, size_t *out_len, void *stream);
/* Forgets the previous frame: the next send encodes every cell, as Send does whenever
 * position or size differ from the previous call (:343-346).  For callers that encoded
 * frames in between through timg_hip_block_encode_grid (a grid row in one launch). */
void timg_hip_block_canvas_forget(timg_hip_block_canvas *c);

/* ---- sixel canvas: timg::SixelCanvas -------------------------------------
 * Replaces the two libsixel calls of SixelCanvas::Send
 * (src/sixel-canvas.cc:137-145: sixel_dither_initialize + sixel_encode) and
 * the padding that precedes them (:111-120).  Per frame: pad to a multiple of
 * 6 rows, blend the pad rows only, <=256-colour adaptive palette (15-bit
 * histogram + median cut, LARGE_LUM / REP_AVERAGE_COLORS), exact nearest
 * colour + Floyd-Steinberg diffusion, band/colour RLE.  Output per frame:
 * cursor-mode string (:66-79) + DCS..ST + "\r" or "\n".  Palette choice is
 * within a stated Delta-E of the CPU restatement (libsixel is un-vendored:
 * parity unpinned). */
#define TIMG_HIP_SIXEL_BROKEN_CURSOR 1 /* SixelOptions::known_broken_cursor_placement */
/* (One lookup rule: a 15-bit cell answers with the palette entry nearest to its centre, and the diffusion is
 * pipelined.  libsixel's own first-hit cache -- inherently serial, ~0.3 s per 800x450 frame -- exists as a checker in the
 * TEST-ONLY libtimg_hip_debug.so (timg_hip_debug_sixel_encode_first_hit, csrc/sixel_canvas.hip), not here; a maintainer
 * who needs libsixel's exact cache behaviour keeps the CPU SixelCanvas.  Unknown flag bits are TIMG_HIP_ERR_ARG.) */

size_t timg_hip_sixel_max_bytes(int w, int h); /* 1024 + w*round6(h)*5, :123 */


/* The asynchronous form of timg_hip_sixel_encode (device-resident frames and output only).  SixelCanvas::Send hands
 * the sequencer a FUTURE and returns (src/sixel-canvas.cc:128-154); a host that keeps a device busy wants the same
 * from the library: timg_hip_sixel_encode returns the frames' byte counts, so every call ends with a blocking
 * read-back, and the next batch's first kernel cannot be enqueued before it (measured: ~70 us of idle device per
 * 1.5 ms step).  Here the call only ENQUEUES -- kernels, then a copy of the byte counts into the job's pinned words,
 * then an event, all on `stream` -- and returns; timg_hip_sixel_encode_wait blocks until THAT call's counts have
 * arrived and reports them (or the call's error) exactly as the synchronous form would.  The caller may enqueue
 * further work -- the next batch's scale and encode included, with another job -- before it waits; one job holds one
 * call at a time.  `out` must stay untouched until the wait returns; bytes are those of timg_hip_sixel_encode.
 * Streams: the encoder's scratch memory belongs to the context, one call at a time.  Calls on ONE stream follow each
 * other in stream order; a sixel call (blocking or not) on ANOTHER stream of the same context is ordered behind the
 * call in flight by the library itself (an event wait on the device, nobody blocks) -- correct, not concurrent: callers
 * that want two encodes to overlap use two contexts.  A job belongs to its context (like a scaler): destroy it before
 * the context; destroying a job waits for the call it still holds. */
typedef struct timg_hip_sixel_job timg_hip_sixel_job;
int timg_hip_sixel_job_create(timg_hip_ctx *ctx, int max_frames, timg_hip_sixel_job **out);
void timg_hip_sixel_job_destroy(timg_hip_sixel_job *job);
int timg_hip_sixel_encode_async(timg_hip_ctx *ctx, const uint8_t *fb_device, int w, int h, int stride,
                                size_t frame_stride, int n_frames, int flags,
                                const timg_hip_blend *pad_blend_or_null, char *out_device, size_t out_cap,
                                void *stream, timg_hip_sixel_job *job);
int timg_hip_sixel_encode_wait(timg_hip_sixel_job *job, size_t *out_len /* n_frames of the call */);
/* ---- scale + compose + sixel encode in ONE call (device-resident batches) ------------------
 * What an ImageSource and a SixelCanvas do to a batch of frames back to back -- ImageScaler::Scale +
 * AlphaComposeBackground (src/qoi-image-source.cc:63-74), then SixelCanvas::Send (src/sixel-canvas.cc:
 * 100-155) -- with the batch cut into `pieces` whose chains (scale -> histogram -> median cut ->
 * diffusion -> bands) run on streams the context owns, forked from and joined to `stream` once per
 * piece: the scale of a piece (all CUs, bandwidth) runs beside the per-frame serial stages of the
 * pieces in front of it (one workgroup per frame).  Bytes are those of timg_hip_scale_blend followed
 * by timg_hip_sixel_encode with pad_blend = blend.  src: n_frames source frames in DEVICE memory;
 * scaled: device memory for n_frames packed out_w x out_h frames (left there); out / out_cap /
 * out_on_device / out_len as timg_hip_sixel_encode.  pieces: 0 = library's choice (1 today: measured,
 * profiles/r3/fused_pieces.txt -- the scale kernel's coarse tiles cost what the overlap wins), up to 4.
 * scale_ms (optional): device time of the scale kernels, summed over the pieces (HIP events on the
 * pieces' own streams) -- what a roofline needs when the kernels are not alone on the chip.
 * Returns after synchronising `stream`. */
int timg_hip_scale_sixel_encode(timg_hip_ctx *ctx, timg_hip_scaler *s, const uint8_t *src, int src_stride,
                                size_t src_frame_stride, uint8_t *scaled, int n_frames,
                                const timg_hip_blend *blend, int sixel_flags, char *out, size_t out_cap,
                                int out_on_device, size_t *out_len, int pieces, float *scale_ms,
                                void *stream);
/* Frames up to 4095 pixels wide (columns travel in 12-bit fields); wider ones are refused
 * with TIMG_HIP_ERR_UNSUPP. */

int timg_hip_sixel_encode(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h,
                          int stride, size_t frame_stride, int fb_on_device,
                          int n_frames, int flags, const timg_hip_blend *pad_blend,
                          char *out, size_t out_cap, int out_on_device,
                          size_t *out_len, void *stream);

/* ---- graphics-protocol canvases at --compress=0: timg::KittyGraphicsCanvas,
 * timg::ITerm2GraphicsCanvas, png::Encode ------------------------------------
 * Replace what the encoder closure of KittyGraphicsCanvas::Send
 * (src/kitty-canvas.cc:167-214, no tmux pass-through) and of
 * ITerm2GraphicsCanvas::Send (src/iterm2-canvas.cc:52-71) computes -- png::Encode
 * (src/timg-png.cc:91-153), EncodeBase64 (src/timg-base64.h:28-55) and the escape
 * framing -- for DisplayOptions::compress_pixel_level == 0 (`--compress=0`): the
 * zlib stream then consists of stored blocks and every byte is positional.  Other
 * levels are libdeflate's match finder and stay on the host.
 * flags: TIMG_HIP_GFX_RGB24 = png::ColorEncoding::kRGB_24 (local alpha handling).
 * Frames of a batch share (w, h); frame i goes to out + i * out_cap, out_len[i]
 * bytes.  kitty image ids are the caller's (the reference derives them from
 * time(), src/kitty-canvas.cc:47-52). */
#define TIMG_HIP_GFX_RGB24 1
size_t timg_hip_png_bytes(int w, int h, int flags); /* exact size of the PNG */
size_t timg_hip_gfx_max_bytes(int w, int h);        /* any of the three outputs fits */
int timg_hip_png_encode(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride,
                        size_t frame_stride, int fb_on_device, int n_frames, int flags,
                        char *out, size_t out_cap, int out_on_device, size_t *out_len,
                        void *stream);
int timg_hip_kitty_encode(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride,
                          size_t frame_stride, int fb_on_device, int n_frames, int flags,
                          const uint32_t *image_ids, char *out, size_t out_cap,
                          int out_on_device, size_t *out_len, void *stream);
int timg_hip_iterm2_encode(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride,
                           size_t frame_stride, int fb_on_device, int n_frames, int flags,
                           char *out, size_t out_cap, int out_on_device, size_t *out_len,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TIMG_HIP_H */
