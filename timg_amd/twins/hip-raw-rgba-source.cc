#include "hip-raw-rgba-source.h"

#include <cstdio>
#include <cstring>
#include <vector>

#include "hip-context.h"
#include "hip-device-frames.h"

namespace timg {

static uint32_t PackColor(rgba_t c) {
    uint32_t v;
    memcpy(&v, &c, 4);
    return v;
}

HipRawRGBASource::~HipRawRGBASource() {
    if (image_) UnregisterDeviceFrame(image_.get());
    if (device_image_) (void)timg_hip_free(ctx_, device_image_);
}

ImageSource *HipRawRGBASource::TryCreate(const std::string &filename, const DisplayOptions &options,
                                         int frame_offset, int frame_count) {
    std::unique_ptr<HipRawRGBASource> s(new HipRawRGBASource(filename));
    return s->LoadAndScale(options, frame_offset, frame_count) ? s.release() : nullptr;
}

std::string HipRawRGBASource::FormatTitle(const std::string &format_string) const {
    return FormatFromParameters(format_string, filename_, orig_width_, orig_height_, "hip-rgba");
}

bool HipRawRGBASource::LoadAndScale(const DisplayOptions &opts, int, int) {
    options_ = opts;
    ctx_     = SharedHipContext();
    if (!ctx_) return false;  // no device: the next loader of the chain gets the file

    // -- the source frame, in device memory
    uint8_t *src = nullptr;
    int w = 0, h = 0;
    char kind_name[16];
    unsigned seed = 0, frame = 0;
    if (sscanf(filename().c_str(), "synth:%15[a-z]:%dx%d:%u:%u", kind_name, &w, &h, &seed, &frame) >= 4) {
        const int kind = !strcmp(kind_name, "noise")   ? TIMG_HIP_SYNTH_NOISE
                         : !strcmp(kind_name, "photo") ? TIMG_HIP_SYNTH_PHOTO
                         : !strcmp(kind_name, "alpha") ? TIMG_HIP_SYNTH_ALPHA
                                                       : -1;
        if (kind < 0 || w <= 0 || h <= 0) return false;
        if (timg_hip_malloc(ctx_, (size_t)w * h * 4, (void **)&src) != TIMG_HIP_OK) return false;
        if (timg_hip_synth_frames(ctx_, kind, w, h, seed, (int)frame, 1, src, 0, 1, nullptr) != TIMG_HIP_OK) {
            (void)timg_hip_free(ctx_, src);
            return false;
        }
    } else {
        const size_t len = filename().size();
        if (len < 5 || filename().compare(len - 5, 5, ".rgba") != 0) return false;
        FILE *f = fopen(filename().c_str(), "rb");
        if (!f) return false;
        unsigned char head[16];
        uint32_t fw = 0, fh = 0;
        bool ok = fread(head, 1, 16, f) == 16 && memcmp(head, "TIMGRGBA", 8) == 0;
        if (ok) {
            fw = head[8] | head[9] << 8 | head[10] << 16 | (uint32_t)head[11] << 24;
            fh = head[12] | head[13] << 8 | head[14] << 16 | (uint32_t)head[15] << 24;
            ok = fw > 0 && fh > 0 && fw <= 32768 && fh <= 32768;
        }
        std::vector<uint8_t> pixels;
        if (ok) {
            pixels.resize((size_t)fw * fh * 4);
            ok = fread(pixels.data(), 1, pixels.size(), f) == pixels.size();
        }
        fclose(f);
        if (!ok) return false;
        w = (int)fw;
        h = (int)fh;
        if (timg_hip_malloc(ctx_, pixels.size(), (void **)&src) != TIMG_HIP_OK) return false;
        if (timg_hip_memcpy_h2d(ctx_, src, pixels.data(), pixels.size(), nullptr) != TIMG_HIP_OK) {
            (void)timg_hip_free(ctx_, src);
            return false;
        }
    }
    orig_width_  = w;
    orig_height_ = h;

    // -- geometry: the reference's own rule (src/image-source.cc:47-153)
    int target_width, target_height;
    CalcScaleToFitDisplay(w, h, opts, false, &target_width, &target_height);

    // -- scale and compose on the device, as src/qoi-image-source.cc:63-74 does on the host
    timg_hip_scaler *scaler = nullptr;
    bool ok = timg_hip_scaler_create(ctx_, w, h, TIMG_HIP_FMT_RGBA, target_width, target_height,
                                     TIMG_HIP_FILTER_STB_DEFAULT, &scaler) == TIMG_HIP_OK &&
              timg_hip_malloc(ctx_, (size_t)target_width * target_height * 4, (void **)&device_image_) == TIMG_HIP_OK;
    int transparent = 0;
    ok = ok && timg_hip_scale_blend(ctx_, scaler, src, 0, 0, 1, device_image_, 0, 0, 1, 1, nullptr, &transparent,
                                    nullptr) == TIMG_HIP_OK;
    // the background getter is only consulted when a pixel needs it (src/framebuffer.cc:113-121)
    if (ok && transparent && opts.bgcolor_getter) {
        timg_hip_blend b;
        b.enabled   = 1;
        b.bg        = PackColor(opts.bgcolor_getter());
        b.pattern   = PackColor(opts.bg_pattern_color);
        b.pattern_w = opts.pattern_size * opts.cell_x_px;
        b.pattern_h = opts.pattern_size * opts.cell_y_px / 2;
        b.start_row = 0;
        ok = timg_hip_alpha_compose(ctx_, device_image_, target_width, target_height, 0, 0, 1, 1, &b, nullptr,
                                    nullptr) == TIMG_HIP_OK;
    }
    if (ok) ok = timg_hip_sync(ctx_, nullptr) == TIMG_HIP_OK;
    if (scaler) timg_hip_scaler_destroy(scaler);
    (void)timg_hip_free(ctx_, src);
    if (!ok) return false;
    image_.reset(new timg::Framebuffer(target_width, target_height));
    RegisterDeviceFrame(image_.get(), device_image_);
    return true;
}

void HipRawRGBASource::SendFrames(const Duration &, int, const volatile sig_atomic_t &,
                                  const Renderer::WriteFramebufferFun &sink) {
    if (!host_filled_ && HostPixelsNeeded()) {  // a canvas that reads the Framebuffer itself
        if (timg_hip_memcpy_d2h(ctx_, (void *)image_->begin(), device_image_,
                                (size_t)image_->width() * image_->height() * 4, nullptr) != TIMG_HIP_OK)
            HipFatal(ctx_, "HipRawRGBASource");
        host_filled_ = true;
    }
    const int indent = options_.center_horizontally ? (options_.width - image_->width()) / 2 : 0;
    sink(indent, 0, *image_, SeqType::FrameImmediate, {});
}

}  // namespace timg
