#include "hip-device-frames.h"

#include <atomic>
#include <map>
#include <mutex>

namespace timg {

namespace {
std::mutex mu;
std::map<const Framebuffer *, const uint8_t *> frames;
std::atomic<int> consumers{0};
}  // namespace

void RegisterDeviceFrame(const Framebuffer *fb, const uint8_t *device_pixels) {
    std::lock_guard<std::mutex> l(mu);
    frames[fb] = device_pixels;
}

void UnregisterDeviceFrame(const Framebuffer *fb) {
    std::lock_guard<std::mutex> l(mu);
    frames.erase(fb);
}

const uint8_t *DevicePixels(const Framebuffer &fb) {
    std::lock_guard<std::mutex> l(mu);
    auto it = frames.find(&fb);
    return it == frames.end() ? nullptr : it->second;
}

void DeviceFrameConsumerCreated() { ++consumers; }
void DeviceFrameConsumerDestroyed() { --consumers; }
bool HostPixelsNeeded() { return consumers.load() <= 0; }

}  // namespace timg
