// tests/twins/host-frames-source.h -- TEST INFRASTRUCTURE: the reference's CPU path for frames that are
// already decoded, as a timg::ImageSource.  It does what the reference's own loaders do once they hold the
// pixels -- src/qoi-image-source.cc:42-77 / src/stb-image-source.cc:44-61: copy into a Framebuffer,
// ImageScaler::Create + Scale (the reference's scaler), Framebuffer::AlphaComposeBackground -- and sends
// frames with the loop of src/stb-image-source.cc:172-205.  twin_check compares HipRawRGBASource's
// multi-frame streams against it; twin_bench times it as the CPU side of the drop-in measurement.
// Only reference classes are used here.
#ifndef TESTS_TWINS_HOST_FRAMES_SOURCE_H
#define TESTS_TWINS_HOST_FRAMES_SOURCE_H

#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "display-options.h"
#include "framebuffer.h"
#include "image-scaler.h"
#include "hip-image-scaler.h"
#include "image-source.h"

namespace timg {

class HostFramesSource final : public ImageSource {
public:
    // frames: `n` RGBA8 frames of w x h back to back (not owned; must outlive LoadAndScale)
    HostFramesSource(const std::string &name, const uint8_t *frames, int n, int w, int h, bool hip_scaler = false)
        : ImageSource(name), src_(frames), n_src_(n), w_(w), h_(h), hip_scaler_(hip_scaler) {}

    // Frames that already ARE Framebuffers (not owned; one per source frame, uncropped): what a decoder leaves behind.
    // Without them LoadAndScale copies every frame out of `frames` first -- fine for the parity drivers, but in a
    // timed run that copy (33 MB per 4K frame, sixteen loaders at once) is host memory traffic a decoder does not add.
    void UseDecodedFrames(const std::vector<timg::Framebuffer *> *decoded) { decoded_ = decoded; }

    // crop window (x, y, w, h) applied before scaling (what GraphicsMagick's crop()/trim() would leave);
    // default: the whole frame
    void SetCrop(int x, int y, int w, int h) {
        cx_ = x; cy_ = y; cw_ = w; ch_ = h;
    }

    bool LoadAndScale(const DisplayOptions &opts, int frame_offset, int frame_count) final {
        options_     = opts;
        const int f0 = std::max(0, std::min(frame_offset, n_src_ - 1));
        int n        = n_src_ - f0;
        if (frame_count > 0) n = std::min(n, frame_count);
        const int sw = cw_ > 0 ? cw_ : w_, sh = ch_ > 0 ? ch_ : h_;
        int tw, th;
        CalcScaleToFitDisplay(sw, sh, opts, false, &tw, &th);
        for (int i = 0; i < n; ++i) {
            std::unique_ptr<timg::Framebuffer> copy;
            timg::Framebuffer *in_p = nullptr;
            if (decoded_ && cw_ <= 0 && ch_ <= 0) {
                in_p = (*decoded_)[f0 + i];
            } else {
                copy.reset(new timg::Framebuffer(sw, sh));  // (the copy the reference's loaders make)
                const uint8_t *frame = src_ + (size_t)(f0 + i) * w_ * h_ * 4;
                for (int y = 0; y < sh; ++y)
                    memcpy((uint8_t *)copy->begin() + (size_t)y * sw * 4, frame + ((size_t)(cy_ + y) * w_ + cx_) * 4, (size_t)sw * 4);
                in_p = copy.get();
            }
            timg::Framebuffer &in = *in_p;
            // hip_scaler_: what a timg built with the twin factory does with a decoded file -- the frame is in HOST
            // memory, HipImageScaler::Create stands where ImageScaler::Create stood (src/stb-image-source.cc:44-61,
            // src/qoi-image-source.cc:42-77): upload, scale and compose on the device, the result back in a Framebuffer
            std::unique_ptr<timg::Framebuffer> out(new timg::Framebuffer(tw, th));
            // (nullptr from the twin's factory -- no device, or the back-end switched off after a failure -- means the next
            // back-end, as in the patched ImageScaler::Create: integration/timg-hip.patch)
            std::unique_ptr<ImageScaler> hip_made;
            if (hip_scaler_) hip_made = HipImageScaler::Create(sw, sh, ImageScaler::ColorFmt::kRGBA, tw, th);
            // (a run that asked for the device scaler and did not get one must not pass as "identical": only a back-end
            // that was switched off after a failure -- the degrade test -- may fall through to the reference's scaler)
            if (hip_scaler_ && !hip_made && !HipDegraded()) return false;
            if (hip_made) {
                auto &scaler = hip_made;
                static_cast<HipImageScaler *>(scaler.get())->ScaleAndCompose(
                    in, out.get(), options_.bgcolor_getter, options_.bg_pattern_color,
                    options_.pattern_size * options_.cell_x_px, options_.pattern_size * options_.cell_y_px / 2);
            } else {
                auto scaler = ImageScaler::Create(sw, sh, ImageScaler::ColorFmt::kRGBA, tw, th);
                if (!scaler) return false;
                scaler->Scale(in, out.get());
                out->AlphaComposeBackground(options_.bgcolor_getter, options_.bg_pattern_color,
                                            options_.pattern_size * options_.cell_x_px,
                                            options_.pattern_size * options_.cell_y_px / 2);
            }
            frames_.push_back(std::move(out));
        }
        return !frames_.empty();
    }

    void SendFrames(const Duration &duration, int loops, const volatile sig_atomic_t &interrupt_received,
                    const Renderer::WriteFramebufferFun &sink) final {
        int last_height         = -1;
        // (by the frames of the SOURCE, as src/stb-image-source.cc:175 decides -- its frames_ holds the whole file, the
        // frame limit only bounds the inner loop: --frames=1 of an animation is an animation of one frame, looped)
        const bool is_animation = n_src_ > 1;
        if (!is_animation) loops = 1;
        const bool loop_forever = loops < 0;
        const timg::Duration time_from_first_frame;
        bool is_first = true;
        const int indent = options_.center_horizontally ? (options_.width - frames_[0]->width()) / 2 : 0;
        for (int k = 0; (loop_forever || k < loops) && !interrupt_received && time_from_first_frame < duration; ++k) {
            for (size_t f = 0; f < frames_.size() && !interrupt_received; ++f) {
                const int dy = is_animation && last_height > 0 ? -last_height : 0;
                SeqType seq  = SeqType::FrameImmediate;
                if (is_animation) seq = is_first ? SeqType::StartOfAnimation : SeqType::AnimationFrame;
                sink(indent, dy, *frames_[f], seq, std::min(time_from_first_frame, duration));
                last_height = frames_[f]->height();
                is_first    = false;
            }
        }
    }

    std::string FormatTitle(const std::string &fmt) const final {
        return FormatFromParameters(fmt, filename_, w_, h_, "host-rgba");
    }

private:
    const uint8_t *const src_;
    const int n_src_, w_, h_;
    const bool hip_scaler_;
    const std::vector<timg::Framebuffer *> *decoded_ = nullptr;
    int cx_ = 0, cy_ = 0, cw_ = 0, ch_ = 0;
    DisplayOptions options_;
    std::vector<std::unique_ptr<timg::Framebuffer>> frames_;
};

}  // namespace timg
#endif
