// timg_amd/twins/hip-raw-rgba-source.h -- a device-resident timg::ImageSource (SURVEY.md §8f-1;
// src/image-source.h:30-101).  Frames are created or uploaded in device memory, scaled and
// composed there (timg_hip_scale_blend with src_on_device = dst_on_device = 1, a batch of frames
// per launch) and handed to the renderer as a plain Framebuffer whose device copy is registered
// in hip-device-frames.h: the Hip canvases encode straight from device memory, any other canvas
// gets the pixels copied back first.  The two host-side copies of the stb / qoi callers
// (src/stb-image-source.cc:48-50, src/qoi-image-source.cc:50-52) do not exist on this path.
//
// Names it answers to (everything else: LoadAndScale returns false, the next loader is tried):
//   synth:<noise|photo|alpha>:<W>x<H>:<seed>[:<first frame>[:<frames>]]
//                   the measurement plan's synthetic frames (timg_hip_synth_frames), never on the
//                   host; <frames> > 1 is a video stream (BASELINE config 4)
//   <file>.rgba     "TIMGRGBA", uint32 width, uint32 height (little endian), then one or more
//                   frames of RGBA8 rows back to back (more than one: an animation)
//
// Like the reference's loaders it honours frame_offset / frame_count (src/video-source.cc:
// 309-311: skip `frame_offset` frames, show at most `frame_count` when that is > 0), loops
// (src/stb-image-source.cc:172-205) and -- for still images only, as
// src/graphics-magick-source.cc:231-241 -- DisplayOptions::crop_border / auto_crop: the bounding
// box comes from timg_hip_autocrop_bbox on the full-resolution frame, the scaler is created for
// the box and reads the window in place (pointer + stride, no copy).
#ifndef TIMG_AMD_TWINS_HIP_RAW_RGBA_SOURCE_H
#define TIMG_AMD_TWINS_HIP_RAW_RGBA_SOURCE_H

#include <cstdint>
#include <memory>
#include <string>

#include "display-options.h"
#include "framebuffer.h"
#include "image-source.h"
#include "timg_hip.h"

namespace timg {

class HipRawRGBASource final : public ImageSource {
public:
    explicit HipRawRGBASource(const std::string &filename) : ImageSource(filename) {}
    ~HipRawRGBASource() override;

    // What ImageSource::Create (src/image-source.cc:155-221) puts in front of its chain:
    //   if (ImageSource *s = HipRawRGBASource::TryCreate(filename, options, frame_offset, frame_count)) return s;
    static ImageSource *TryCreate(const std::string &filename, const DisplayOptions &options, int frame_offset,
                                  int frame_count);

    bool LoadAndScale(const DisplayOptions &options, int frame_offset, int frame_count) final;
    void SendFrames(const Duration &duration, int loops, const volatile sig_atomic_t &interrupt_received,
                    const Renderer::WriteFramebufferFun &sink) final;
    std::string FormatTitle(const std::string &format_string) const final;
    bool IsAnimationBeforeFrameLimit() const final { return frames_in_source_ > 1; }

    // Frames scaled per device launch (source frames of a long stream are generated / uploaded, scaled and
    // released chunk by chunk; the scaled frames all stay resident).
    static constexpr int kChunkFrames = 64;

private:
    DisplayOptions options_;
    timg_hip_ctx *ctx_ = nullptr;
    int orig_width_ = 0, orig_height_ = 0;
    int frames_in_source_ = 1;                  // before frame_offset / frame_count
    int n_frames_         = 0;                  // frames SendFrames shows
    size_t frame_bytes_   = 0;                  // one scaled frame
    uint8_t *device_frames_ = nullptr;          // n_frames_ scaled, composed frames back to back
    std::unique_ptr<timg::Framebuffer> image_;  // the host side: filled only for canvases that need it
    int host_holds_ = -1;                       // index of the frame whose pixels image_ holds
};

}  // namespace timg
#endif
