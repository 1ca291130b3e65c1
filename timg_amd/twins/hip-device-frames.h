// timg_amd/twins/hip-device-frames.h -- side channel for frames that live in device memory
// (SURVEY.md §8f-1).  timg hands frames from an ImageSource to a canvas as `const Framebuffer &`
// (src/renderer.h:38-40); a device-resident source (hip-raw-rgba-source.h) registers the device
// copy of such a framebuffer here and the Hip canvases look it up in Send: pixels then never
// leave the GPU between the scaler and the encoder.  Everyone else sees a plain Framebuffer.
#ifndef TIMG_AMD_TWINS_HIP_DEVICE_FRAMES_H
#define TIMG_AMD_TWINS_HIP_DEVICE_FRAMES_H

#include <cstdint>

#include "framebuffer.h"

namespace timg {

// The device copy (RGBA8, packed rows) of fb, valid until it is unregistered.
void RegisterDeviceFrame(const Framebuffer *fb, const uint8_t *device_pixels);
void UnregisterDeviceFrame(const Framebuffer *fb);
const uint8_t *DevicePixels(const Framebuffer &fb);  // nullptr: the pixels only exist on the host

// A source may leave the HOST pixels of a registered framebuffer unfilled while every canvas
// alive reads device frames (the Hip canvases announce themselves); timg has one canvas per run.
void DeviceFrameConsumerCreated();
void DeviceFrameConsumerDestroyed();
bool HostPixelsNeeded();

}  // namespace timg
#endif
