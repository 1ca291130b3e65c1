// timg_amd/csrc/scale_stream.hip -- streaming scale kernels (placeholder: the
// generic kernel covers every plan until the streaming variants land).
#include "context.h"

namespace timg_amd {

bool PrepareStreamSchedule(timg_hip_scaler *s, std::string *why_not) {
    (void)s;
    if (why_not) *why_not = "not built";
    return false;
}

void ReleaseStreamSchedule(timg_hip_scaler *s) { (void)s; }

hipError_t LaunchScaleStream(const timg_hip_scaler *s, const DevBlend &blend,
                             const FrameBatch &batch, hipStream_t stream) {
    (void)s; (void)blend; (void)batch; (void)stream;
    return hipErrorNotSupported;
}

}  // namespace timg_amd
