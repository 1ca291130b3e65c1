// timg_amd/twins/hip-raw-rgba-source.h -- a device-resident timg::ImageSource (SURVEY.md §8f-1;
// src/image-source.h:30-101).  Frames are created or uploaded in device memory, scaled and
// composed there (timg_hip_scale_blend with src_on_device = dst_on_device = 1) and handed to
// the renderer as a plain Framebuffer whose device copy is registered in hip-device-frames.h:
// the Hip canvases encode straight from device memory, any other canvas gets the pixels
// copied back first.  The two host-side copies of the stb / qoi callers
// (src/stb-image-source.cc:48-50, src/qoi-image-source.cc:50-52) do not exist on this path.
//
// Names it answers to (everything else: LoadAndScale returns false, the next loader is tried):
//   synth:<noise|photo|alpha>:<W>x<H>:<seed>[:<frame>]   the measurement plan's synthetic frames
//                                                        (timg_hip_synth_frames), never on the host
//   <file>.rgba     "TIMGRGBA", uint32 width, uint32 height (little endian), then RGBA8 rows
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

private:
    DisplayOptions options_;
    timg_hip_ctx *ctx_ = nullptr;
    int orig_width_ = 0, orig_height_ = 0;
    uint8_t *device_image_ = nullptr;           // the scaled, composed frame
    std::unique_ptr<timg::Framebuffer> image_;  // its host side: filled only for canvases that need it
    bool host_filled_ = false;
};

}  // namespace timg
#endif
