#include "hip-image-scaler.h"

#include <cstdio>
#include <cstring>

#include "hip-context.h"

namespace timg {

static uint32_t PackColor(rgba_t c) {
    uint32_t v;
    memcpy(&v, &c, 4);
    return v;
}

std::unique_ptr<ImageScaler> HipImageScaler::Create(int in_width, int in_height,
                                                    ColorFmt in_color_format,
                                                    int out_width, int out_height) {
    if (!SharedHipContext()) return nullptr;
    // (a context per loader thread: the upload of this image runs beside the kernels and uploads of the others)
    timg_hip_ctx *ctx = LoaderHipContext();
    const int fmt = in_color_format == ColorFmt::kRGBA ? TIMG_HIP_FMT_RGBA : TIMG_HIP_FMT_BGRA;
    // (recycled by geometry: the images of a grid share their plan, hip-context.h)
    timg_hip_scaler *s = HipScalerAcquire(ctx, in_width, in_height, fmt, out_width, out_height, HipScalerFilter());
    if (!s) return nullptr;
    if (HipTwinTrace()) fprintf(stderr, "HipImageScaler: %dx%d -> %dx%d\n", in_width, in_height, out_width, out_height);
    return std::unique_ptr<ImageScaler>(new HipImageScaler(
        ctx, s, in_width, in_height, in_color_format, out_width, out_height));
}

HipImageScaler::~HipImageScaler() { HipScalerRelease(scaler_); }

// The device failed (or failed earlier: HipDegraded) after this scaler had been created: the reference's own scaler for
// the same geometry does the work -- ImageScaler::Create builds it, because HipImageScaler::Create now returns nullptr.
// Same bytes when the timg build scales with stb (src/image-scaler.cc:75-98), the build's own filter otherwise.
void HipImageScaler::ScaleOnCpu(Framebuffer &in, Framebuffer *out, const char *what) {
    HipDegrade(ctx_, what);
    HipCountFrames(kHipTwinScaler, false);
    std::unique_ptr<ImageScaler> cpu = ImageScaler::Create(in_w_, in_h_, fmt_, out_w_, out_h_);
    if (!cpu) HipFatal(ctx_, what);  // (no CPU back-end either: the reference's loaders would have failed the load)
    cpu->Scale(in, out);
}

void HipImageScaler::Scale(Framebuffer &in, Framebuffer *out) {
    if (in.width() != in_w_ || in.height() != in_h_ || out->width() != out_w_ || out->height() != out_h_)
        HipFatal(ctx_, "HipImageScaler::Scale: geometry");
    if (HipDegraded() ||
        HipCall(ctx_, [&]() {
            return timg_hip_scale_blend(ctx_, scaler_, (const uint8_t *)in.begin(), 0, 0, 0, (uint8_t *)out->begin(), 0, 0, 0,
                                        1, nullptr, nullptr, nullptr);
        }) != TIMG_HIP_OK)
        ScaleOnCpu(in, out, "HipImageScaler::Scale");
    else
        HipCountFrames(kHipTwinScaler, true);
}

void HipImageScaler::ScaleAndCompose(Framebuffer &in, Framebuffer *out,
                                     const Framebuffer::bgcolor_query &get_bg,
                                     rgba_t pattern, int pattern_width, int pattern_height) {
    // Two phases ON THE DEVICE, one trip over PCIe each way: the frame is uploaded and scaled into device memory with
    // the "is any pixel not opaque" flag (the laziness condition of src/framebuffer.cc:113-117: the getter may block
    // on a terminal query, src/timg.cc:924-928, so it must not be called for nothing); only when the flag is up is
    // the getter consulted and the scaled frame composed where it lies; then ONE download.  (Until round 5 the scaled
    // frame went to the host, and a frame with a transparent pixel was uploaded and downloaded a second time by
    // timg_hip_alpha_compose.)
    if (in.width() != in_w_ || in.height() != in_h_ || out->width() != out_w_ || out->height() != out_h_)
        HipFatal(ctx_, "HipImageScaler::ScaleAndCompose: geometry");
    auto on_cpu = [&](const char *what) {  // (src/qoi-image-source.cc:63-74: the pair of calls this one replaces)
        ScaleOnCpu(in, out, what);
        out->AlphaComposeBackground(get_bg, pattern, pattern_width, pattern_height);
    };
    if (HipDegraded()) return on_cpu("HipImageScaler::ScaleAndCompose");
    const size_t bytes = (size_t)out_w_ * out_h_ * 4;
    uint8_t *scaled    = (uint8_t *)HipPoolMalloc(ctx_, bytes);
    if (!scaled) return on_cpu("HipImageScaler::ScaleAndCompose: device memory");
    int transparent = 0;
    if (HipCall(ctx_, [&]() {
            return timg_hip_scale_blend(ctx_, scaler_, (const uint8_t *)in.begin(), 0, 0, 0, scaled, 0, 0, 1, 1, nullptr,
                                        &transparent, nullptr);
        }) != TIMG_HIP_OK) {
        HipPoolFree(ctx_, scaled);
        return on_cpu("HipImageScaler::ScaleAndCompose");
    }
    if (get_bg && transparent) {  // src/framebuffer.cc:111,117: otherwise the getter is not consulted
        timg_hip_blend b;
        b.enabled   = 1;
        b.bg        = PackColor(get_bg());
        b.pattern   = PackColor(pattern);
        b.pattern_w = pattern_width;
        b.pattern_h = pattern_height;
        b.start_row = 0;
        if (HipCall(ctx_, [&]() {
                return timg_hip_alpha_compose(ctx_, scaled, out_w_, out_h_, 0, 0, 1, 1, &b, nullptr, nullptr);
            }) != TIMG_HIP_OK) {
            HipPoolFree(ctx_, scaled);
            return on_cpu("timg_hip_alpha_compose");  // (the getter is asked a second time: it caches, src/timg.cc:924-928)
        }
    }
    if (timg_hip_memcpy_d2h(ctx_, out->begin(), scaled, bytes, nullptr) != TIMG_HIP_OK) {
        HipPoolFree(ctx_, scaled);
        return on_cpu("HipImageScaler::ScaleAndCompose: download");
    }
    HipPoolFree(ctx_, scaled);
    HipCountFrames(kHipTwinScaler, true);
}

}  // namespace timg
