#include "hip-raw-rgba-source.h"

#include <algorithm>
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
    if (device_frames_) {
        // (a canvas may have copied a frame of this source a moment ago, on the copy context's stream: the block goes back
        // to the pool -- to the next loader, on a stream of its own -- only when those copies have landed)
        HipFrameCopiesDone();
        HipPoolFree(ctx_, device_frames_);
    }
}

ImageSource *HipRawRGBASource::TryCreate(const std::string &filename, const DisplayOptions &options,
                                         int frame_offset, int frame_count) {
    std::unique_ptr<HipRawRGBASource> s(new HipRawRGBASource(filename));
    return s->LoadAndScale(options, frame_offset, frame_count) ? s.release() : nullptr;
}

std::string HipRawRGBASource::FormatTitle(const std::string &format_string) const {
    return FormatFromParameters(format_string, filename_, orig_width_, orig_height_, "hip-rgba");
}

namespace {
// Where the frames come from: the generator or a file; both deliver a run of frames into device memory.
struct FrameFeed {
    int w = 0, h = 0, frames = 1;
    // synth
    int kind = -1;
    unsigned seed = 0, first = 0;
    // file
    FILE *file = nullptr;
    ~FrameFeed() {
        if (file) fclose(file);
    }
    bool Open(const std::string &name) {
        char kind_name[16];
        unsigned count = 1;
        const int got = sscanf(name.c_str(), "synth:%15[a-z]:%dx%d:%u:%u:%u", kind_name, &w, &h, &seed, &first, &count);
        if (got >= 4) {
            kind = !strcmp(kind_name, "noise")   ? TIMG_HIP_SYNTH_NOISE
                   : !strcmp(kind_name, "photo") ? TIMG_HIP_SYNTH_PHOTO
                   : !strcmp(kind_name, "alpha") ? TIMG_HIP_SYNTH_ALPHA
                                                 : -1;
            frames = got >= 6 ? (int)count : 1;
            return kind >= 0 && w > 0 && h > 0 && frames > 0 && w <= 32768 && h <= 32768;
        }
        const size_t len = name.size();
        if (len < 5 || name.compare(len - 5, 5, ".rgba") != 0) return false;
        file = fopen(name.c_str(), "rb");
        if (!file) return false;
        unsigned char head[16];
        if (fread(head, 1, 16, file) != 16 || memcmp(head, "TIMGRGBA", 8) != 0) return false;
        const uint32_t fw = head[8] | head[9] << 8 | head[10] << 16 | (uint32_t)head[11] << 24;
        const uint32_t fh = head[12] | head[13] << 8 | head[14] << 16 | (uint32_t)head[15] << 24;
        if (fw == 0 || fh == 0 || fw > 32768 || fh > 32768) return false;
        w = (int)fw;
        h = (int)fh;
        if (fseek(file, 0, SEEK_END) != 0) return false;
        const long size = ftell(file);
        const size_t one = (size_t)w * h * 4;
        if (size < 16 || (size_t)(size - 16) < one) return false;
        frames = (int)std::min<size_t>((size_t)(size - 16) / one, 1u << 20);
        return true;
    }
    // frames [f0, f0 + n) of the source into dst (device memory, packed)
    bool Deliver(timg_hip_ctx *ctx, int f0, int n, uint8_t *dst, std::vector<uint8_t> *staging) {
        if (kind >= 0)
            return timg_hip_synth_frames(ctx, kind, w, h, seed, (int)first + f0, n, dst, 0, 1, nullptr) == TIMG_HIP_OK;
        const size_t one = (size_t)w * h * 4;
        staging->resize(one * n);
        if (fseek(file, (long)(16 + one * f0), SEEK_SET) != 0) return false;
        if (fread(staging->data(), 1, staging->size(), file) != staging->size()) return false;
        return timg_hip_memcpy_h2d(ctx, dst, staging->data(), staging->size(), nullptr) == TIMG_HIP_OK;
    }
};
}  // namespace

bool HipRawRGBASource::LoadAndScale(const DisplayOptions &opts, int frame_offset, int frame_count) {
    options_ = opts;
    FrameFeed feed;
    if (!feed.Open(filename())) return false;
    ctx_ = SharedHipContext();
    if (!ctx_) return false;  // no device: the next loader of the chain gets the file
    // The frames are delivered, cropped, scaled and composed on the LOADER thread's own context (own stream, own
    // scratch: LoaderHipContext): timg creates its sources on a pool of loader threads (src/timg.cc:948-968), and on
    // the one shared context their calls queued behind each other's -- 64 sources of a 4K frame were loaded 0.13 ms
    // apart, 8.3 of the 11.3 ms the whole 8x8 grid took (profiles/r6/twin_timeline.txt); a frame alone fills 40 of the
    // chip's 1 024 workgroup slots, eight loaders' frames run side by side.  The call below ends with a sync of
    // that context, so whoever reads the frames later -- a canvas on another context's stream -- finds them complete.
    timg_hip_ctx *const load_ctx = [&]() {
        timg_hip_ctx *c = LoaderHipContext();
        return c ? c : ctx_;
    }();

    orig_width_       = feed.w;
    orig_height_      = feed.h;
    frames_in_source_ = feed.frames;
    // which frames are shown (src/video-source.cc:309-311, src/stb-image-source.cc:163-165)
    const int f0 = std::max(0, std::min(frame_offset, feed.frames - 1));
    int n        = feed.frames - f0;
    if (frame_count > 0) n = std::min(n, frame_count);
    if (n < 1) return false;
    const bool is_animation = feed.frames > 1;

    const size_t src_frame = (size_t)feed.w * feed.h * 4;
    const int chunk        = std::min(n, kChunkFrames);
    uint8_t *src           = (uint8_t *)HipPoolMalloc(ctx_, src_frame * chunk);
    if (!src) return false;
    std::vector<uint8_t> staging;
    timg_hip_scaler *scaler = nullptr;
    bool ok                 = true;
    // the window of the source that is shown: everything, or -- still images only, like
    // src/graphics-magick-source.cc:231-241 -- what --crop-border / --auto-crop leave of it
    int box[4] = {0, 0, feed.w, feed.h};
    int target_width = 0, target_height = 0;
    std::vector<int> transparent(chunk);
    bool any_transparent = false;
    for (int done = 0; ok && done < n; done += chunk) {
        const int m = std::min(chunk, n - done);
        ok          = feed.Deliver(load_ctx, f0 + done, m, src, &staging);
        if (ok && done == 0) {
            if (!is_animation && (opts.crop_border > 0 || opts.auto_crop)) {
                if (opts.auto_crop) {
                    ok = HipCall(load_ctx, [&]() {
                             return timg_hip_autocrop_bbox(load_ctx, src, feed.w, feed.h, 0, 0, 1, 1, std::max(0, opts.crop_border),
                                                           box, nullptr);
                         }) == TIMG_HIP_OK;
                    if (ok && (box[2] <= 0 || box[3] <= 0)) {  // nothing but border: GraphicsMagick's trim() keeps a pixel
                        box[0] = box[1] = 0;
                        box[2] = box[3] = 1;
                    }
                } else {
                    const int c = opts.crop_border;
                    box[2]      = std::max(1, feed.w - 2 * c);
                    box[3]      = std::max(1, feed.h - 2 * c);
                    box[0]      = std::min(c, feed.w - box[2]);
                    box[1]      = std::min(c, feed.h - box[3]);
                }
            }
            // -- geometry: the reference's own rule (src/image-source.cc:47-153), on the cropped size
            CalcScaleToFitDisplay(box[2], box[3], opts, false, &target_width, &target_height);
            frame_bytes_ = (size_t)target_width * target_height * 4;
            if (ok) {
                scaler = HipScalerAcquire(load_ctx, box[2], box[3], TIMG_HIP_FMT_RGBA, target_width, target_height, HipScalerFilter());
                device_frames_ = (uint8_t *)HipPoolMalloc(ctx_, frame_bytes_ * n);
                ok             = scaler != nullptr && device_frames_ != nullptr;
            }
        }
        // -- scale on the device, as src/qoi-image-source.cc:63-68 does on the host: one launch per chunk
        const uint8_t *window = src + (size_t)box[1] * feed.w * 4 + (size_t)box[0] * 4;
        ok = ok && HipCall(load_ctx, [&]() {
                       return timg_hip_scale_blend(load_ctx, scaler, window, feed.w * 4, src_frame, 1,
                                                   device_frames_ + frame_bytes_ * done, 0, 0, 1, m, nullptr, transparent.data(),
                                                   nullptr);
                   }) == TIMG_HIP_OK;
        for (int i = 0; ok && i < m; ++i) any_transparent = any_transparent || transparent[i] != 0;
    }
    // the background getter is only consulted when a pixel needs it (src/framebuffer.cc:113-121); frames
    // without such a pixel come out of the compose untouched, exactly as the reference leaves them
    if (ok && any_transparent && opts.bgcolor_getter) {
        timg_hip_blend b;
        b.enabled   = 1;
        b.bg        = PackColor(opts.bgcolor_getter());
        b.pattern   = PackColor(opts.bg_pattern_color);
        b.pattern_w = opts.pattern_size * opts.cell_x_px;
        b.pattern_h = opts.pattern_size * opts.cell_y_px / 2;
        b.start_row = 0;
        ok = HipCall(load_ctx, [&]() {
                 return timg_hip_alpha_compose(load_ctx, device_frames_, target_width, target_height, 0, 0, 1, n, &b, nullptr, nullptr);
             }) == TIMG_HIP_OK;
    }
    if (ok) ok = timg_hip_sync(load_ctx, nullptr) == TIMG_HIP_OK;
    HipScalerRelease(scaler);
    HipPoolFree(ctx_, src);
    if (!ok) return false;
    n_frames_ = n;
    image_.reset(new timg::Framebuffer(target_width, target_height));
    RegisterDeviceFrame(image_.get(), device_frames_);
    return true;
}

void HipRawRGBASource::SendFrames(const Duration &duration, int loops, const volatile sig_atomic_t &interrupt_received,
                                  const Renderer::WriteFramebufferFun &sink) {
    // the loop of src/stb-image-source.cc:172-205 (frames of this source carry no delay)
    // (by the SOURCE's frame count, like LoadAndScale above and src/stb-image-source.cc:175 `frames_.size() > 1`:
    // --frames=1 of a multi-frame source is still presented as an animation of one frame)
    const bool is_animation = frames_in_source_ > 1;
    if (!is_animation) loops = 1;
    const bool loop_forever = loops < 0;  // (kNotInitialized is negative too)
    const int indent        = options_.center_horizontally ? (options_.width - image_->width()) / 2 : 0;
    int last_height         = -1;
    bool is_first           = true;
    const timg::Duration time_from_first_frame;  // (no delays: stays at zero)
    // (a still image is sent whatever `duration` says, like src/qoi-image-source.cc:85-90)
    for (int k = 0; (loop_forever || k < loops) && !interrupt_received && (!is_animation || time_from_first_frame < duration);
         ++k) {
        for (int f = 0; f < n_frames_ && !interrupt_received; ++f) {
            const uint8_t *device = device_frames_ + frame_bytes_ * f;
            RegisterDeviceFrame(image_.get(), device);  // (the one Framebuffer stands for frame f now)
            if (HostPixelsNeeded() && host_holds_ != f) {  // a canvas that reads the Framebuffer itself
                if (timg_hip_memcpy_d2h(ctx_, (void *)image_->begin(), device, frame_bytes_, nullptr) != TIMG_HIP_OK)
                    HipFatal(ctx_, "HipRawRGBASource");
                host_holds_ = f;
            }
            const int dy = is_animation && last_height > 0 ? -last_height : 0;
            SeqType seq  = SeqType::FrameImmediate;
            if (is_animation) seq = is_first ? SeqType::StartOfAnimation : SeqType::AnimationFrame;
            sink(indent, dy, *image_, seq, std::min(time_from_first_frame, duration));
            last_height = image_->height();
            is_first    = false;
        }
        if (loop_forever && n_frames_ == 1) break;
    }
}

}  // namespace timg
