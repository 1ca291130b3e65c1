// timg_amd/csrc/dev_alloc.hip -- see dev_alloc.h
#include "dev_alloc.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace timg_amd {
namespace {

constexpr unsigned char kPoison = 0xA5;

struct Guarded {
    void *va = nullptr;       // reserved range: [unmapped granule][mapped ...][unmapped granule]
    size_t va_bytes = 0;
    size_t map_bytes = 0;     // mapped part, starts at va + gran
    size_t gran = 0;
    size_t bytes = 0;         // what the caller asked for (rounded up to 4)
    hipMemGenericAllocationHandle_t handle{};
};

std::mutex g_mu;
std::map<void *, Guarded> g_live;

int ReadMode() {
    const char *e = getenv("TIMG_HIP_GUARD");
    if (!e || !*e) return 0;
    if (!strcmp(e, "start")) return 1;
    if (!strcmp(e, "end16")) return 2;
    if (!strcmp(e, "end4")) return 3;
    fprintf(stderr, "timg_hip: TIMG_HIP_GUARD=%s is not one of start|end16|end4\n", e);
    abort();
}

hipError_t GuardMalloc(void **out, size_t bytes, int mode) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop{};
    prop.type          = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id   = dev;
    size_t gran        = 0;
    if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
    if (gran == 0) return hipErrorInvalidValue;
    Guarded g;
    g.gran      = gran;
    g.bytes     = (bytes + 3) & ~(size_t)3;
    if (g.bytes == 0) g.bytes = 4;
    g.map_bytes = (g.bytes + gran - 1) / gran * gran;
    g.va_bytes  = g.map_bytes + 2 * gran;
    if ((e = hipMemAddressReserve(&g.va, g.va_bytes, gran, nullptr, 0)) != hipSuccess) return e;
    if ((e = hipMemCreate(&g.handle, g.map_bytes, &prop, 0)) != hipSuccess) {
        (void)hipMemAddressFree(g.va, g.va_bytes);
        return e;
    }
    char *mapped = (char *)g.va + gran;
    if ((e = hipMemMap(mapped, g.map_bytes, 0, g.handle, 0)) != hipSuccess) {
        (void)hipMemRelease(g.handle);
        (void)hipMemAddressFree(g.va, g.va_bytes);
        return e;
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags    = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(mapped, g.map_bytes, &acc, 1)) != hipSuccess ||
        (e = hipMemset(mapped, kPoison, g.map_bytes)) != hipSuccess ||
        (e = hipDeviceSynchronize()) != hipSuccess) {  // (hipMemset returns before the fill ran)
        (void)hipMemUnmap(mapped, g.map_bytes);
        (void)hipMemRelease(g.handle);
        (void)hipMemAddressFree(g.va, g.va_bytes);
        return e;
    }
    char *p = mapped;
    if (mode == 2) p = mapped + ((g.map_bytes - g.bytes) & ~(size_t)15);
    if (mode == 3) p = mapped + (g.map_bytes - g.bytes);
    std::lock_guard<std::mutex> l(g_mu);
    g_live[p] = g;
    *out      = p;
    return hipSuccess;
}

bool GuardFree(void *p, hipError_t *err) {
    Guarded g;
    {
        std::lock_guard<std::mutex> l(g_mu);
        auto it = g_live.find(p);
        if (it == g_live.end()) return false;
        g = it->second;
        g_live.erase(it);
    }
    (void)hipDeviceSynchronize();
    char *mapped      = (char *)g.va + g.gran;
    const size_t head = (size_t)((char *)p - mapped), tail = g.map_bytes - head - g.bytes;
    std::vector<unsigned char> slack(head + tail);
    bool dirty = false;
    hipError_t ce = hipSuccess;
    if (head) ce = hipMemcpy(slack.data(), mapped, head, hipMemcpyDeviceToHost);
    if (tail && ce == hipSuccess)
        ce = hipMemcpy(slack.data() + head, (char *)p + g.bytes, tail, hipMemcpyDeviceToHost);
    if (ce != hipSuccess) {
        fprintf(stderr, "timg_hip GUARD: reading back the slack of a %zu-byte buffer (head %zu, tail %zu) failed: %s\n",
                g.bytes, head, tail, hipGetErrorString(ce));
        abort();
    }
    for (size_t i = 0; i < slack.size() && !dirty; ++i) {
        if (slack[i] != kPoison) {
            const long off = i < head ? (long)i - (long)head : (long)(i - head) + (long)g.bytes;
            fprintf(stderr, "timg_hip GUARD: byte at offset %ld of a %zu-byte device buffer was overwritten (0x%02x)\n",
                    off, g.bytes, slack[i]);
            dirty = true;
        }
    }
    if (dirty) abort();
    // The virtual range is NOT given back: (1) a use after free then faults as well, (2) with ROCm 7.0's
    // runtime a range that is freed and handed out again by the next hipMemAddressReserve keeps stale
    // translations -- copies into the new mapping land elsewhere: 4 KiB holes of zeros, bytes past the end
    // (scratch/r3_guard_stress.py; 0 errors in 300 round trips once ranges are never reused).  Virtual
    // address space is plentiful for a test run.
    hipError_t e = hipMemUnmap(mapped, g.map_bytes);
    if (e == hipSuccess) e = hipMemRelease(g.handle);
    *err = e;
    return true;
}

}  // namespace

int GuardMode() {
    static const int mode = ReadMode();
    return mode;
}

// Fault injection for the callers' out-of-memory paths (tests/test_twins.py, twin_check under TIMG_HIP_FAIL_MALLOC):
// TIMG_HIP_FAIL_MALLOC=<k> makes the k-th device allocation AFTER the process's first timg_hip_init has returned,
// counted from 1, fail ONCE with hipErrorOutOfMemory -- what a device that is full at that moment answers; the
// allocation after it goes through.  Read once; unset or 0: never.
static std::mutex g_inject_mu;
static long g_inject_at = 0, g_inject_count = 0;
static bool g_inject_armed = false;
void ArmMallocInjection() {
    std::lock_guard<std::mutex> l(g_inject_mu);
    if (g_inject_armed) return;
    g_inject_armed = true;
    const char *e = getenv("TIMG_HIP_FAIL_MALLOC");
    g_inject_at   = e && *e ? atol(e) : 0L;
}
static bool InjectedFailure() {
    std::lock_guard<std::mutex> l(g_inject_mu);
    return g_inject_armed && g_inject_at > 0 && ++g_inject_count == g_inject_at;
}

hipError_t DevMalloc(void **ptr, size_t bytes) {
    if (InjectedFailure()) return hipErrorOutOfMemory;
    const int mode = GuardMode();
    if (mode == 0) return hipMalloc(ptr, bytes ? bytes : 1);
    return GuardMalloc(ptr, bytes, mode);
}

hipError_t DevFree(void *ptr) {
    if (!ptr) return hipSuccess;
    if (GuardMode() != 0) {
        hipError_t e = hipSuccess;
        if (GuardFree(ptr, &e)) return e;
    }
    return hipFree(ptr);
}

}  // namespace timg_amd
