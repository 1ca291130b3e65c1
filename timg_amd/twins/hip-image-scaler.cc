#include "hip-image-scaler.h"

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
    return std::unique_ptr<ImageScaler>(new HipImageScaler(
        ctx, s, in_width, in_height, in_color_format, out_width, out_height));
}

HipImageScaler::~HipImageScaler() { HipScalerRelease(scaler_); }

void HipImageScaler::Scale(Framebuffer &in, Framebuffer *out) {
    if (in.width() != in_w_ || in.height() != in_h_ || out->width() != out_w_ ||
        out->height() != out_h_ ||
        HipCall(ctx_, [&]() {
            return timg_hip_scale_blend(ctx_, scaler_, (const uint8_t *)in.begin(), 0, 0, 0, (uint8_t *)out->begin(), 0, 0, 0,
                                        1, nullptr, nullptr, nullptr);
        }) != TIMG_HIP_OK)
        HipFatal(ctx_, "HipImageScaler::Scale");
}

void HipImageScaler::ScaleAndCompose(Framebuffer &in, Framebuffer *out,
                                     const Framebuffer::bgcolor_query &get_bg,
                                     rgba_t pattern, int pattern_width, int pattern_height) {
    int transparent = 0;
    if (HipCall(ctx_, [&]() {
            return timg_hip_scale_blend(ctx_, scaler_, (const uint8_t *)in.begin(), 0, 0, 0, (uint8_t *)out->begin(), 0, 0, 0,
                                        1, nullptr, &transparent, nullptr);
        }) != TIMG_HIP_OK)
        HipFatal(ctx_, "HipImageScaler::ScaleAndCompose");
    if (!get_bg || !transparent) return;  // src/framebuffer.cc:111,117: getter not consulted
    timg_hip_blend b;
    b.enabled   = 1;
    b.bg        = PackColor(get_bg());
    b.pattern   = PackColor(pattern);
    b.pattern_w = pattern_width;
    b.pattern_h = pattern_height;
    b.start_row = 0;
    if (HipCall(ctx_, [&]() {
            return timg_hip_alpha_compose(ctx_, (uint8_t *)out->begin(), out_w_, out_h_, 0, 0, 0, 1, &b, nullptr, nullptr);
        }) != TIMG_HIP_OK)
        HipFatal(ctx_, "timg_hip_alpha_compose");
}

}  // namespace timg
