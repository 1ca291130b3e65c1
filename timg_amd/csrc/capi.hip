// timg_amd/csrc/capi.hip -- C-ABI of libtimg_hip.so (include/timg_hip.h):
// context, memory helpers, the ImageScaler twin and the standalone alpha
// compose.  Canvas entry points live in block_canvas.hip / sixel_canvas.hip.
#include <cstring>
#include <functional>
#include <new>

#include "context.h"
#include "cu_mask.h"

using timg_amd::DevBlend;
using timg_amd::DevPlan;
using timg_amd::FrameBatch;

static thread_local std::string g_global_error;

namespace timg_amd {
hipError_t LaunchScaleStream(const timg_hip_scaler *s, const DevBlend &blend,
                             const FrameBatch &batch, hipStream_t stream, int slot);
bool PrepareStreamSchedule(timg_hip_scaler *s, std::string *why_not);
void ReleaseStreamSchedule(timg_hip_scaler *s);
int StreamShapeBits(const timg_hip_scaler *s);
int SixelEncodeImpl(timg_hip_ctx *ctx, const uint8_t *fb, int w, int h, int stride, size_t frame_stride,
                    int fb_on_device, int n_frames, int flags, const timg_hip_blend *pad_blend, char *out,
                    size_t out_cap, int out_on_device, size_t *out_len, void *stream, int pieces_req,
                    const std::function<hipError_t(int, int, int, hipStream_t)> *before_piece, float *hook_ms,
                    timg_hip_sixel_job *job);
}  // namespace timg_amd

DevBlend MakeDevBlend(const timg_hip_blend *b) {
    DevBlend d;
    memset(&d, 0, sizeof(d));
    d.pw = d.ph = 1;
    if (!b) return d;
    d.start_row = b->start_row < 0 ? 0 : b->start_row;
    if (!b->enabled) return d;
    const uint32_t bg = b->bg, pat = b->pattern;
    if ((bg >> 24) == 0) return d;  // "nothing to do", src/framebuffer.cc:120-121
    d.enabled = 1;
    auto lin = [](uint32_t c, float out[3]) {
        for (int i = 0; i < 3; ++i) {
            const uint32_t v = (c >> (8 * i)) & 0xffu;
            out[i]           = (float)(v * v);
        }
    };
    lin(bg, d.bg);
    lin(pat, d.pat);
    // src/framebuffer.cc:124-125
    d.checker = !((pat >> 24) == 0 || pat == bg || b->pattern_w <= 0 || b->pattern_h <= 0);
    if (d.checker) {
        d.pw = b->pattern_w;
        d.ph = b->pattern_h;
    }
    d.pw_magic = d.pw > 1 ? (unsigned)((0x100000000ull + (unsigned)d.pw - 1) / (unsigned)d.pw) : 0u;
    return d;
}

int CopyFramesToHost(timg_hip_ctx *ctx, char *out, size_t out_cap, const char *dout, const size_t *out_len,
                     int n_frames, hipStream_t st) {
    size_t worst = 0, total = 0;
    for (int i = 0; i < n_frames; ++i) {
        worst = out_len[i] > worst ? out_len[i] : worst;
        total += out_len[i];
    }
    if (total == 0) {
        TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
        return TIMG_HIP_OK;
    }
    if (n_frames == 1 || total <= (size_t)128 * 1024) {  // small: the runtime's own staging does as well
        for (int i = 0; i < n_frames; ++i)
            if (out_len[i])
                TIMG_HIP_TRY(ctx, hipMemcpyAsync(out + (size_t)i * out_cap, dout + (size_t)i * out_cap, out_len[i],
                                                 hipMemcpyDeviceToHost, st));
        TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
        return TIMG_HIP_OK;
    }
    // a destination the device can write itself (pinned / registered host memory): one strided copy, no staging
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, out) == hipSuccess && attr.type == hipMemoryTypeHost) {
        TIMG_HIP_TRY(ctx, hipMemcpy2DAsync(out, out_cap, dout, out_cap, worst, (size_t)n_frames, hipMemcpyDeviceToHost, st));
        TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
        return TIMG_HIP_OK;
    }
    (void)hipGetLastError();  // (an ordinary pointer is "invalid value" to the query: not an error of ours)
    // groups of frames whose strided image fits the pinned budget
    const size_t kBudget = (size_t)64 << 20;
    const size_t pitch   = (worst + 255) & ~(size_t)255;
    int group            = (int)(kBudget / pitch);
    group                = group < 1 ? 1 : group > n_frames ? n_frames : group;
    TIMG_HIP_TRY(ctx, ctx->pin[3].Reserve(pitch * (size_t)group));
    char *pin = (char *)ctx->pin[3].ptr;
    for (int f0 = 0; f0 < n_frames; f0 += group) {
        const int m = n_frames - f0 < group ? n_frames - f0 : group;
        size_t w    = 0;
        for (int i = 0; i < m; ++i) w = out_len[f0 + i] > w ? out_len[f0 + i] : w;
        if (w == 0) continue;
        TIMG_HIP_TRY(ctx, hipMemcpy2DAsync(pin, pitch, dout + (size_t)f0 * out_cap, out_cap, w, (size_t)m,
                                           hipMemcpyDeviceToHost, st));
        TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
        for (int i = 0; i < m; ++i)
            if (out_len[f0 + i]) memcpy(out + (size_t)(f0 + i) * out_cap, pin + (size_t)i * pitch, out_len[f0 + i]);
    }
    return TIMG_HIP_OK;
}

extern "C" {

int timg_hip_version(void) { return (1 << 16) | 0; }

const char *timg_hip_last_error(const timg_hip_ctx *ctx) {
    return ctx ? ctx->last_error.c_str() : g_global_error.c_str();
}

int timg_hip_init(int device, timg_hip_ctx **out) {
    if (!out) return TIMG_HIP_ERR_ARG;
    *out      = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_global_error = std::string("no HIP device: ") +
                         (e != hipSuccess ? hipGetErrorString(e) : "count is 0");
        return TIMG_HIP_ERR_DEVICE;
    }
    if (device < 0 || device >= count) {
        g_global_error = "device index out of range";
        return TIMG_HIP_ERR_ARG;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        g_global_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return TIMG_HIP_ERR_DEVICE;
    }
    timg_hip_ctx *ctx = new (std::nothrow) timg_hip_ctx();
    if (!ctx) return TIMG_HIP_ERR_NOMEM;
    ctx->device = device;
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_global_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        delete ctx;
        return TIMG_HIP_ERR_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        ctx->cu_count = prop.multiProcessorCount;
    *out = ctx;
    timg_amd::ArmMallocInjection();
    return TIMG_HIP_OK;
}

void timg_hip_destroy(timg_hip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &b : ctx->dev) b.Release();
    for (auto &b : ctx->pin) b.Release();
    (void)hipStreamDestroy(ctx->stream);
    for (auto st : ctx->side)
        if (st) (void)hipStreamDestroy(st);
    for (auto ev : ctx->join_event)
        if (ev) (void)hipEventDestroy(ev);
    for (auto ev : ctx->hook_event)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
    if (ctx->sixel_done) (void)hipEventDestroy(ctx->sixel_done);
    for (auto st : ctx->owned_streams) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
    if (ctx->order_event) (void)hipEventDestroy(ctx->order_event);
    delete ctx;
}

int timg_hip_malloc(timg_hip_ctx *ctx, size_t bytes, void **dev_ptr) {
    if (!ctx || !dev_ptr) return TIMG_HIP_ERR_ARG;
    TIMG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = timg_amd::DevMalloc(dev_ptr, bytes);
    if (e == hipErrorOutOfMemory) return ctx->Fail(TIMG_HIP_ERR_NOMEM, "hipMalloc(%zu)", bytes);
    if (e != hipSuccess) return ctx->FailHip(e, "hipMalloc");
    return TIMG_HIP_OK;
}

int timg_hip_free(timg_hip_ctx *ctx, void *dev_ptr) {
    if (!ctx) return TIMG_HIP_ERR_ARG;
    TIMG_HIP_TRY(ctx, timg_amd::DevFree(dev_ptr));
    return TIMG_HIP_OK;
}

int timg_hip_memcpy_h2d(timg_hip_ctx *ctx, void *dst, const void *src, size_t n, void *stream) {
    if (!ctx) return TIMG_HIP_ERR_ARG;
    hipStream_t st = ctx->Stream(stream);
    TIMG_HIP_TRY(ctx, hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st));
    TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
    return TIMG_HIP_OK;
}

int timg_hip_memcpy_d2h(timg_hip_ctx *ctx, void *dst, const void *src, size_t n, void *stream) {
    if (!ctx) return TIMG_HIP_ERR_ARG;
    hipStream_t st = ctx->Stream(stream);
    TIMG_HIP_TRY(ctx, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, st));
    TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
    return TIMG_HIP_OK;
}

int timg_hip_memcpy_d2d(timg_hip_ctx *ctx, void *dst, const void *src, size_t n, void *stream) {
    if (!ctx) return TIMG_HIP_ERR_ARG;
    TIMG_HIP_TRY(ctx, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, ctx->Stream(stream)));
    return TIMG_HIP_OK;  // (asynchronous on the stream, like a kernel)
}

int timg_hip_sync(timg_hip_ctx *ctx, void *stream) {
    if (!ctx) return TIMG_HIP_ERR_ARG;
    TIMG_HIP_TRY(ctx, hipStreamSynchronize(ctx->Stream(stream)));
    return TIMG_HIP_OK;
}

// ---- streams for a partitioned pipeline (timg_hip.h; the mask's layout: cu_mask.h) ---------------------------------
int timg_hip_stream_create(timg_hip_ctx *ctx, int reserved_cus_per_xcd, int high_priority, void **stream) {
    if (!ctx || !stream || reserved_cus_per_xcd < 0) return TIMG_HIP_ERR_ARG;
    TIMG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t mask[timg_amd::kCuMaskWords];
    const int reserved = timg_amd::BuildCuMask(ctx->cu_count, reserved_cus_per_xcd, mask);
    if (reserved < 0)
        return ctx->Fail(TIMG_HIP_ERR_UNSUPP, "timg_hip_stream_create: %d CUs of every XCD reserved on a device of %d CUs",
                         reserved_cus_per_xcd, ctx->cu_count);
    hipStream_t st = nullptr;
    if (reserved > 0) {
        TIMG_HIP_TRY(ctx, hipExtStreamCreateWithCUMask(&st, (uint32_t)((ctx->cu_count + 31) / 32), mask));
        // (a masked stream has the default priority: the extension takes no priority argument)
    } else {
        int least = 0, greatest = 0;
        TIMG_HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        TIMG_HIP_TRY(ctx, hipStreamCreateWithPriority(&st, hipStreamNonBlocking, high_priority ? greatest : least));
    }
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->owned_streams.push_back(st);
    *stream = (void *)st;
    return TIMG_HIP_OK;
}

int timg_hip_stream_destroy(timg_hip_ctx *ctx, void *stream) {
    if (!ctx || !stream) return TIMG_HIP_ERR_ARG;
    std::lock_guard<std::mutex> lock(ctx->mu);
    for (size_t i = 0; i < ctx->owned_streams.size(); ++i)
        if ((void *)ctx->owned_streams[i] == stream) {
            ctx->owned_streams.erase(ctx->owned_streams.begin() + (long)i);
            if (ctx->sixel_stream == (hipStream_t)stream && ctx->sixel_in_flight && ctx->sixel_last_job) {
                TIMG_HIP_TRY(ctx, hipEventSynchronize(ctx->sixel_last_job->done));
                ctx->sixel_in_flight = false;
            }
            TIMG_HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
            TIMG_HIP_TRY(ctx, hipStreamDestroy((hipStream_t)stream));
            return TIMG_HIP_OK;
        }
    return ctx->Fail(TIMG_HIP_ERR_ARG, "timg_hip_stream_destroy: not a stream of this context");
}

int timg_hip_stream_wait_stream(timg_hip_ctx *ctx, void *waiter, void *signaller) {
    if (!ctx) return TIMG_HIP_ERR_ARG;
    hipStream_t w = ctx->Stream(waiter), s = ctx->Stream(signaller);
    if (w == s) return TIMG_HIP_OK;
    std::lock_guard<std::mutex> lock(ctx->mu);  // (one event: a wait captures the record it follows, then the event is free again)
    if (!ctx->order_event) TIMG_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->order_event, hipEventDisableTiming));
    TIMG_HIP_TRY(ctx, hipEventRecord(ctx->order_event, s));
    TIMG_HIP_TRY(ctx, hipStreamWaitEvent(w, ctx->order_event, 0));
    return TIMG_HIP_OK;
}

// ---- scaler ------------------------------------------------------------------

int timg_hip_scaler_create(timg_hip_ctx *ctx, int in_w, int in_h, int in_fmt, int out_w,
                           int out_h, int filter, timg_hip_scaler **out) {
    if (!ctx || !out) return TIMG_HIP_ERR_ARG;
    *out = nullptr;
    if (in_fmt != TIMG_HIP_FMT_RGBA && in_fmt != TIMG_HIP_FMT_BGRA)
        return ctx->Fail(TIMG_HIP_ERR_ARG, "unknown input format %d", in_fmt);
    if (filter != TIMG_HIP_FILTER_STB_DEFAULT && filter != TIMG_HIP_FILTER_TRIANGLE)
        return ctx->Fail(TIMG_HIP_ERR_ARG, "unknown filter %d", filter);
    timg_hip_scaler *s = new (std::nothrow) timg_hip_scaler();
    if (!s) return TIMG_HIP_ERR_NOMEM;
    s->ctx = ctx;
    if (!timg_amd::BuildResamplePlan(in_w, in_h, in_fmt, out_w, out_h, filter, &s->plan)) {
        delete s;
        return ctx->Fail(TIMG_HIP_ERR_ARG, "bad geometry %dx%d -> %dx%d", in_w, in_h, out_w,
                         out_h);
    }
    const timg_amd::ResamplePlan &p = s->plan;
    // One device allocation, sections 16-byte aligned.
    auto align = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_ht = 0;
    const size_t o_hc = align(o_ht + p.h_taps.size() * sizeof(int2));
    const size_t o_vr = align(o_hc + p.h_coeff.size() * sizeof(float));
    const size_t o_vi = align(o_vr + p.v_runs.size() * sizeof(int2));
    const size_t o_vc = align(o_vi + p.v_rows.size() * sizeof(int));
    const size_t total = align(o_vc + p.v_coeff.size() * sizeof(float));
    std::vector<char> host(total, 0);
    memcpy(&host[o_ht], p.h_taps.data(), p.h_taps.size() * sizeof(int2));
    memcpy(&host[o_hc], p.h_coeff.data(), p.h_coeff.size() * sizeof(float));
    memcpy(&host[o_vr], p.v_runs.data(), p.v_runs.size() * sizeof(int2));
    memcpy(&host[o_vi], p.v_rows.data(), p.v_rows.size() * sizeof(int));
    memcpy(&host[o_vc], p.v_coeff.data(), p.v_coeff.size() * sizeof(float));
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = timg_amd::DevMalloc(&s->tables, total);
    if (e == hipSuccess) e = hipMemcpy(s->tables, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (s->tables) (void)timg_amd::DevFree(s->tables);
        delete s;
        return ctx->FailHip(e, "uploading resample tables");
    }
    char *base             = (char *)s->tables;
    s->dev.in_w            = in_w;
    s->dev.in_h            = in_h;
    s->dev.out_w           = out_w;
    s->dev.out_h           = out_h;
    s->dev.swap_rb         = in_fmt == TIMG_HIP_FMT_BGRA;
    s->dev.vertical_first  = p.vertical_first;
    s->dev.h_sequential    = p.h_sequential;
    s->dev.h_width         = p.h_width;
    s->dev.h_taps          = (const int2 *)(base + o_ht);
    s->dev.h_coeff         = (const float *)(base + o_hc);
    s->dev.v_runs          = (const int2 *)(base + o_vr);
    s->dev.v_rows          = (const int *)(base + o_vi);
    s->dev.v_coeff         = (const float *)(base + o_vc);
    std::string why;
    s->streaming_ok = timg_amd::PrepareStreamSchedule(s, &why);
    *out            = s;
    return TIMG_HIP_OK;
}

void timg_hip_scaler_destroy(timg_hip_scaler *s) {
    if (!s) return;
    timg_amd::ReleaseStreamSchedule(s);
    if (s->tables) (void)timg_amd::DevFree(s->tables);
    delete s;
}

int timg_hip_scaler_set_kernel(timg_hip_scaler *s, int which) {
    if (!s || which < 0 || which > 4) return TIMG_HIP_ERR_ARG;
    if (which >= 2 && !s->streaming_ok)
        return s->ctx->Fail(TIMG_HIP_ERR_UNSUPP, "streaming kernel does not cover this plan");
    s->forced_kernel = which > 2 ? 2 : which;
    s->stream_cfg[3] = which > 2 ? which - 2 : 0;  // first channel set the streaming kernel tries
    return TIMG_HIP_OK;
}

int timg_hip_scaler_info(const timg_hip_scaler *s, int info[8]) {
    if (!s || !info) return TIMG_HIP_ERR_ARG;
    info[0] = s->plan.vertical_first;
    info[1] = s->plan.h_width;
    info[2] = s->plan.v_is_gather;
    info[3] = s->plan.v_widest;
    info[4] = s->plan.h_filter;
    info[5] = s->plan.v_filter;
    info[6] = s->streaming_ok ? 1 | (timg_amd::StreamShapeBits(s) << 1) : 0;
    info[7] = s->plan.max_active_rows;
    return TIMG_HIP_OK;
}

size_t timg_hip_scaler_algorithmic_bytes(const timg_hip_scaler *s) {
    if (!s) return 0;
    return (size_t)4 * s->plan.in_w * s->plan.in_h + (size_t)4 * s->plan.out_w * s->plan.out_h;
}

int timg_hip_scale_blend(timg_hip_ctx *ctx, timg_hip_scaler *s, const uint8_t *src,
                         int src_stride, size_t src_frame_stride, int src_on_device,
                         uint8_t *dst, int dst_stride, size_t dst_frame_stride,
                         int dst_on_device, int n_frames, const timg_hip_blend *blend,
                         int *any_transparent, void *stream) {
    if (!ctx || !s || !src || !dst || n_frames <= 0) return TIMG_HIP_ERR_ARG;
    if (s->ctx != ctx) return ctx->Fail(TIMG_HIP_ERR_ARG, "scaler belongs to another context");
    const timg_amd::ResamplePlan &p = s->plan;
    if (src_stride == 0) src_stride = p.in_w * 4;
    if (dst_stride == 0) dst_stride = p.out_w * 4;
    if (src_stride < p.in_w * 4 || dst_stride < p.out_w * 4 || (src_stride & 3) ||
        (dst_stride & 3))
        return ctx->Fail(TIMG_HIP_ERR_ARG, "bad stride");
    if (src_frame_stride == 0) src_frame_stride = (size_t)src_stride * p.in_h;
    if (dst_frame_stride == 0) dst_frame_stride = (size_t)dst_stride * p.out_h;
    if (((uintptr_t)src & 3) || ((uintptr_t)dst & 3) || (src_frame_stride & 3) ||
        (dst_frame_stride & 3))
        return ctx->Fail(TIMG_HIP_ERR_ARG, "buffers must be 4-byte aligned");

    TIMG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->Stream(stream);
    std::unique_lock<std::mutex> lock(ctx->mu, std::defer_lock);
    const bool uses_scratch = !src_on_device || !dst_on_device || any_transparent;
    if (uses_scratch) lock.lock();

    FrameBatch batch;
    batch.n_frames          = n_frames;
    batch.src_stride        = (size_t)src_stride;
    batch.src_frame_stride  = src_frame_stride;
    batch.dst_stride        = (size_t)dst_stride;
    batch.dst_frame_stride  = dst_frame_stride;
    batch.transparent_flags = nullptr;

    const size_t src_bytes = src_frame_stride * (size_t)(n_frames - 1) + (size_t)src_stride * p.in_h;
    const size_t dst_bytes = dst_frame_stride * (size_t)(n_frames - 1) + (size_t)dst_stride * p.out_h;
    if (src_on_device) {
        batch.src = src;
    } else {
        TIMG_HIP_TRY(ctx, ctx->dev[0].Reserve(src_bytes));
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(ctx->dev[0].ptr, src, src_bytes, hipMemcpyHostToDevice, st));
        batch.src = (const uint8_t *)ctx->dev[0].ptr;
    }
    if (dst_on_device) {
        batch.dst = dst;
    } else {
        TIMG_HIP_TRY(ctx, ctx->dev[1].Reserve(dst_bytes));
        batch.dst = (uint8_t *)ctx->dev[1].ptr;
    }
    if (any_transparent) {
        TIMG_HIP_TRY(ctx, ctx->dev[2].Reserve(sizeof(int) * n_frames));
        TIMG_HIP_TRY(ctx, hipMemsetAsync(ctx->dev[2].ptr, 0, sizeof(int) * n_frames, st));
        batch.transparent_flags = (int *)ctx->dev[2].ptr;
    }

    const DevBlend db = MakeDevBlend(blend);
    hipError_t e;
    if (p.identity) {
        e = timg_amd::LaunchCopyBlend(s->dev, db, batch, st);
    } else {
        const bool want_stream =
            s->forced_kernel == 2 || (s->forced_kernel == 0 && s->streaming_ok);
        e = want_stream ? timg_amd::LaunchScaleStream(s, db, batch, st, 0)
                        : timg_amd::LaunchScaleGeneric(s->dev, db, batch, st);
    }
    if (e != hipSuccess) return ctx->FailHip(e, "scale kernel launch");

    if (!dst_on_device)
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(dst, batch.dst, dst_bytes, hipMemcpyDeviceToHost, st));
    if (any_transparent)
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(any_transparent, batch.transparent_flags,
                                         sizeof(int) * n_frames, hipMemcpyDeviceToHost, st));
    if (uses_scratch) TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
    return TIMG_HIP_OK;
}

int timg_hip_scale_sixel_encode(timg_hip_ctx *ctx, timg_hip_scaler *s, const uint8_t *src, int src_stride,
                                size_t src_frame_stride, uint8_t *scaled, int n_frames, const timg_hip_blend *blend,
                                int sixel_flags, char *out, size_t out_cap, int out_on_device, size_t *out_len,
                                int pieces, float *scale_ms, void *stream) {
    if (!ctx || !s || !src || !scaled || !out || !out_len || n_frames <= 0) return TIMG_HIP_ERR_ARG;
    if (s->ctx != ctx) return ctx->Fail(TIMG_HIP_ERR_ARG, "scaler belongs to another context");
    const timg_amd::ResamplePlan &p = s->plan;
    if (src_stride == 0) src_stride = p.in_w * 4;
    if (src_stride < p.in_w * 4 || (src_stride & 3) || ((uintptr_t)src & 3) || ((uintptr_t)scaled & 3))
        return ctx->Fail(TIMG_HIP_ERR_ARG, "bad stride/alignment");
    if (src_frame_stride == 0) src_frame_stride = (size_t)src_stride * p.in_h;
    if (src_frame_stride & 3) return ctx->Fail(TIMG_HIP_ERR_ARG, "buffers must be 4-byte aligned");
    if (pieces <= 0) {
        // MEASURED (profiles/r3/fused_pieces.txt, 64x 4K -> 800x450): two pieces 2.21 ms per step against 2.25 as two
        // calls, four pieces 2.47 -- a scale launch beside the serial sixel kernels runs two rounds of its coarse tiles
        // (1024 columns x 45 rows, ~0.22 ms each) on the CUs the diffusion leaves free and loses what the overlap wins.
        // So the library's own choice is one piece; callers (or TIMG_HIP_PIECES) can ask for more.
        pieces = 1;
        if (const char *e = getenv("TIMG_HIP_PIECES")) pieces = atoi(e);  // tuning
    }
    const size_t out_frame = (size_t)p.out_w * p.out_h * 4;
    const DevBlend db      = MakeDevBlend(blend);
    const std::function<hipError_t(int, int, int, hipStream_t)> scale_piece = [&](int piece, int f0, int nfr,
                                                                                  hipStream_t st) -> hipError_t {
        FrameBatch batch;
        batch.n_frames          = nfr;
        batch.src               = src + (size_t)f0 * src_frame_stride;
        batch.src_stride        = (size_t)src_stride;
        batch.src_frame_stride  = src_frame_stride;
        batch.dst               = scaled + (size_t)f0 * out_frame;
        batch.dst_stride        = (size_t)p.out_w * 4;
        batch.dst_frame_stride  = out_frame;
        batch.transparent_flags = nullptr;
        if (p.identity) return timg_amd::LaunchCopyBlend(s->dev, db, batch, st);
        const bool want_stream = s->forced_kernel == 2 || (s->forced_kernel == 0 && s->streaming_ok);
        return want_stream ? timg_amd::LaunchScaleStream(s, db, batch, st, piece)
                           : timg_amd::LaunchScaleGeneric(s->dev, db, batch, st);
    };
    return timg_amd::SixelEncodeImpl(ctx, scaled, p.out_w, p.out_h, 0, 0, 1, n_frames, sixel_flags, blend, out, out_cap,
                                     out_on_device, out_len, stream, pieces, &scale_piece, scale_ms, nullptr);
}

int timg_hip_alpha_compose(timg_hip_ctx *ctx, uint8_t *fb, int w, int h, int stride,
                           size_t frame_stride, int on_device, int n_frames,
                           const timg_hip_blend *blend, int *any_transparent, void *stream) {
    if (!ctx || !fb || w <= 0 || h <= 0 || n_frames <= 0 || !blend) return TIMG_HIP_ERR_ARG;
    if (stride == 0) stride = w * 4;
    if (stride < w * 4 || (stride & 3) || ((uintptr_t)fb & 3))
        return ctx->Fail(TIMG_HIP_ERR_ARG, "bad stride/alignment");
    if (frame_stride == 0) frame_stride = (size_t)stride * h;
    TIMG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->Stream(stream);
    std::unique_lock<std::mutex> lock(ctx->mu, std::defer_lock);
    const bool uses_scratch = !on_device || any_transparent;
    if (uses_scratch) lock.lock();
    const size_t bytes = frame_stride * (size_t)(n_frames - 1) + (size_t)stride * h;
    uint8_t *dfb       = fb;
    if (!on_device) {
        TIMG_HIP_TRY(ctx, ctx->dev[0].Reserve(bytes));
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(ctx->dev[0].ptr, fb, bytes, hipMemcpyHostToDevice, st));
        dfb = (uint8_t *)ctx->dev[0].ptr;
    }
    int *flags = nullptr;
    if (any_transparent) {
        TIMG_HIP_TRY(ctx, ctx->dev[2].Reserve(sizeof(int) * n_frames));
        TIMG_HIP_TRY(ctx, hipMemsetAsync(ctx->dev[2].ptr, 0, sizeof(int) * n_frames, st));
        flags = (int *)ctx->dev[2].ptr;
    }
    const DevBlend db = MakeDevBlend(blend);
    hipError_t e = timg_amd::LaunchAlphaCompose(dfb, w, h, (size_t)stride, frame_stride, n_frames,
                                                db, flags, st);
    if (e != hipSuccess) return ctx->FailHip(e, "alpha compose launch");
    if (!on_device)
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(fb, dfb, bytes, hipMemcpyDeviceToHost, st));
    if (any_transparent)
        TIMG_HIP_TRY(ctx, hipMemcpyAsync(any_transparent, flags, sizeof(int) * n_frames,
                                         hipMemcpyDeviceToHost, st));
    if (uses_scratch) TIMG_HIP_TRY(ctx, hipStreamSynchronize(st));
    return TIMG_HIP_OK;
}

}  // extern "C"
