#include "hip-context.h"

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <list>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace timg {

bool HipTwinsEnabled() {
    const char *v = getenv("TIMG_HIP");
    return !(v && v[0] == '0');
}

bool HipTwinTrace() {
    static const bool on = getenv("TIMG_HIP_TWIN_TRACE") != nullptr;
    return on;
}

namespace {
std::atomic<bool> g_degraded{false};
}

bool HipDegraded() { return g_degraded.load(); }

void HipDegrade(timg_hip_ctx *ctx, const char *what) {
    if (!g_degraded.exchange(true))
        fprintf(stderr, "timg: HIP back-end failed in %s: %s -- continuing on the CPU\n", what, timg_hip_last_error(ctx));
}

void HipDegradeUnless(int rc, timg_hip_ctx *ctx, const char *what) {
    if (rc == TIMG_HIP_ERR_UNSUPP) {
        if (HipTwinTrace()) fprintf(stderr, "timg_hip twins: %s refused this frame (%s): the reference's class encodes it\n", what, timg_hip_last_error(ctx));
        return;
    }
    HipDegrade(ctx, what);
}

namespace {
std::atomic<unsigned long> g_frames_device[kHipTwinKinds], g_frames_cpu[kHipTwinKinds];
void PrintTwinStats() {
    const HipTwinCounts c = HipTwinStats();
    fprintf(stderr,
            "timg_hip twins: frames on the device: scaler %lu block %lu sixel %lu graphics %lu; on the CPU: scaler %lu block %lu sixel %lu "
            "graphics %lu; degraded %d\n",
            c.device[0], c.device[1], c.device[2], c.device[3], c.cpu[0], c.cpu[1], c.cpu[2], c.cpu[3], (int)HipDegraded());
}
}  // namespace

void HipCountFrames(HipTwinKind kind, bool on_device, size_t frames) {
    (on_device ? g_frames_device : g_frames_cpu)[kind].fetch_add((unsigned long)frames, std::memory_order_relaxed);
}

HipTwinCounts HipTwinStats() {
    HipTwinCounts c;
    for (int k = 0; k < kHipTwinKinds; ++k) {
        c.device[k] = g_frames_device[k].load();
        c.cpu[k]    = g_frames_cpu[k].load();
    }
    return c;
}

bool HipFailInjected() {
    static const long fail_at = []() {
        const char *e = getenv("TIMG_HIP_FAIL_CALL");
        return e ? atol(e) : 0L;
    }();
    static std::atomic<long> calls{0};
    return fail_at > 0 && calls.fetch_add(1) + 1 == fail_at;
}

static timg_hip_ctx *SharedHipContextEvenIfDegraded() {
    static std::once_flag once;
    static timg_hip_ctx *ctx = nullptr;
    std::call_once(once, []() {
        if (!HipTwinsEnabled()) return;
        const char *d = getenv("TIMG_HIP_DEVICE");
        if (timg_hip_init(d ? atoi(d) : 0, &ctx) != TIMG_HIP_OK) ctx = nullptr;
        if (HipTwinTrace()) {
            fprintf(stderr, "timg_hip twins: device context %s%s\n", ctx ? "created" : "unavailable: ",
                    ctx ? "" : timg_hip_last_error(nullptr));
            atexit(PrintTwinStats);
        }
    });
    return ctx;
}

timg_hip_ctx *SharedHipContext() {
    timg_hip_ctx *ctx = SharedHipContextEvenIfDegraded();
    return g_degraded.load() ? nullptr : ctx;
}

timg_hip_ctx *ExtraHipContext(int k) {
    static std::mutex mu;
    static std::vector<timg_hip_ctx *> extra;
    if (k < 1 || !SharedHipContext()) return nullptr;
    std::lock_guard<std::mutex> l(mu);
    while ((int)extra.size() < k) {
        const char *d     = getenv("TIMG_HIP_DEVICE");
        timg_hip_ctx *ctx = nullptr;
        if (timg_hip_init(d ? atoi(d) : 0, &ctx) != TIMG_HIP_OK) return nullptr;
        extra.push_back(ctx);
    }
    return extra[k - 1];
}

namespace {
std::atomic<timg_hip_ctx *> g_copy_ctx{nullptr};
}

timg_hip_ctx *CopyHipContext() {
    // (objects that exist keep working on their contexts after a degrade: not SharedHipContext(), which hides it then)
    timg_hip_ctx *shared = SharedHipContextEvenIfDegraded();
    if (!shared) return nullptr;
    static std::once_flag once;
    std::call_once(once, []() {
        const char *d   = getenv("TIMG_HIP_DEVICE");
        timg_hip_ctx *c = nullptr;
        if (timg_hip_init(d ? atoi(d) : 0, &c) == TIMG_HIP_OK) g_copy_ctx.store(c);
    });
    timg_hip_ctx *c = g_copy_ctx.load();
    return c ? c : shared;
}

void HipFrameCopiesDone() {
    timg_hip_ctx *c = g_copy_ctx.load();
    if (c) (void)timg_hip_sync(c, nullptr);
}

timg_hip_ctx *LoaderHipContext() {
    // Loader threads (src/timg.cc:948-968 runs 3/4 of the cores as loaders) scale frames that live in HOST memory: a
    // call uploads the frame, scales and downloads the result under its context's lock -- on ONE context every loader
    // waits for the others' uploads (64 frames of 4K: 1.02 ms a frame, 8.1 Gpx/s at sixteen loaders).  Each thread is
    // given one of up to kLoaderContexts contexts of its own stream and staging memory instead, so that one image's
    // upload runs beside another's kernels.  They are the loaders' OWN contexts, created on demand, one per loader
    // thread that actually shows up (a single-image run creates one, not the five encoder contexts in front of it and
    // itself: ADVICE r4) -- the encoders' contexts (ExtraHipContext) are not touched.
    //
    // A thread holds its slot for its lifetime and hands it back when it ends: a process that runs ONE loader at a time
    // -- a single image; a caller that starts a fresh thread pool per presentation, as tests/twins/twin_bench.cc does per
    // run -- keeps meeting the same, warm context (numbered by thread birth, every new pool of one thread paid a new
    // context, its first launch and a scaler plan of its own: one 4K frame 4.3 ms instead of 0.9).
    constexpr int kLoaderContexts = 8;
    static std::mutex mu;
    static timg_hip_ctx *loaders[kLoaderContexts] = {};
    static int users[kLoaderContexts]             = {};
    struct Slot {
        int index;
        Slot() {
            std::lock_guard<std::mutex> l(mu);
            index = 0;
            for (int i = 1; i < kLoaderContexts; ++i)  // (the least used slot, the lowest of them)
                if (users[i] < users[index]) index = i;
            ++users[index];
        }
        ~Slot() {
            std::lock_guard<std::mutex> l(mu);
            --users[index];
        }
    };
    thread_local Slot slot;
    const int mine       = slot.index;
    timg_hip_ctx *shared = SharedHipContext();
    if (!shared) return nullptr;
    std::lock_guard<std::mutex> l(mu);
    if (!loaders[mine]) {
        const char *d = getenv("TIMG_HIP_DEVICE");
        if (timg_hip_init(d ? atoi(d) : 0, &loaders[mine]) != TIMG_HIP_OK) loaders[mine] = nullptr;
    }
    return loaders[mine] ? loaders[mine] : shared;
}

int HipScalerFilter() {
    static const int filter = []() {
        const char *v = getenv("TIMG_HIP_FILTER");
        if (v && !strcmp(v, "bilinear")) return TIMG_HIP_FILTER_TRIANGLE;
        if (v && !strcmp(v, "stb")) return TIMG_HIP_FILTER_STB_DEFAULT;
#if defined(WITH_TIMG_SWS_RESIZE)  // (first, as ImageScaler::Create prefers it when a build defines both: src/image-scaler.cc:24-35)
        return TIMG_HIP_FILTER_TRIANGLE;
#else
        return TIMG_HIP_FILTER_STB_DEFAULT;
#endif
    }();
    return filter;
}

namespace {
// (all contexts of the twins live on ONE device -- TIMG_HIP_DEVICE -- so blocks need no device in their key)
constexpr size_t kPoolBytes = (size_t)4 << 30;
std::mutex g_pool_mu;
std::map<size_t, std::vector<void *>> g_pool_idle;  // size -> blocks
std::map<void *, size_t> g_pool_size;               // every block handed out or idle
size_t g_pool_cached = 0;

// Idle scalers, most recently released first.  A scaler owns device tables (resample plan, two schedule variants with
// their alpha tables: up to tens of megabytes for large geometries), so what stays cached is bounded in TOTAL, not per
// geometry: a slide show or a directory of differently sized images keeps the kMaxIdleScalers it used last and
// destroys the oldest (round 3's pool kept 16 per geometry for the life of the process: the advisor's finding).
typedef std::tuple<timg_hip_ctx *, int, int, int, int, int, int> ScalerKey;  // (a scaler belongs to its context)
std::mutex g_scaler_mu;
std::list<std::pair<ScalerKey, timg_hip_scaler *>> g_scaler_idle;
std::map<timg_hip_scaler *, ScalerKey> g_scaler_key;  // every scaler handed out or idle
constexpr size_t kMaxIdleScalers = 24;
}  // namespace

size_t HipPoolTrim(timg_hip_ctx *ctx) {
    std::vector<void *> drop;
    std::vector<timg_hip_scaler *> scalers;
    {
        std::lock_guard<std::mutex> l(g_pool_mu);
        for (auto &kv : g_pool_idle) {
            for (void *q : kv.second) {
                drop.push_back(q);
                g_pool_size.erase(q);
            }
            kv.second.clear();
        }
        g_pool_cached = 0;
    }
    {
        std::lock_guard<std::mutex> l(g_scaler_mu);
        for (auto &e : g_scaler_idle) {
            scalers.push_back(e.second);
            g_scaler_key.erase(e.second);
        }
        g_scaler_idle.clear();
    }
    for (void *q : drop) (void)timg_hip_free(ctx, q);
    for (timg_hip_scaler *s : scalers) timg_hip_scaler_destroy(s);
    return drop.size() + scalers.size();
}

void *HipPoolMalloc(timg_hip_ctx *ctx, size_t bytes) {
    if (bytes == 0) bytes = 4;
    {
        std::lock_guard<std::mutex> l(g_pool_mu);
        auto it = g_pool_idle.find(bytes);
        if (it != g_pool_idle.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            g_pool_cached -= bytes;
            return p;
        }
    }
    void *p = nullptr;
    if (timg_hip_malloc(ctx, bytes, &p) != TIMG_HIP_OK) {
        // out of memory: everything cached (blocks whose sizes may never recur, idle scalers) goes back, one more try
        HipPoolTrim(ctx);
        if (timg_hip_malloc(ctx, bytes, &p) != TIMG_HIP_OK) return nullptr;
    }
    std::lock_guard<std::mutex> l(g_pool_mu);
    g_pool_size[p] = bytes;
    return p;
}

void HipPoolFree(timg_hip_ctx *ctx, void *ptr) {
    if (!ptr) return;
    {
        std::lock_guard<std::mutex> l(g_pool_mu);
        auto it = g_pool_size.find(ptr);
        if (it != g_pool_size.end() && g_pool_cached + it->second <= kPoolBytes) {
            g_pool_idle[it->second].push_back(ptr);
            g_pool_cached += it->second;
            return;
        }
        if (it != g_pool_size.end()) g_pool_size.erase(it);
    }
    // (what is still enqueued on the context's stream may use the block: the free waits for the device)
    (void)timg_hip_free(ctx, ptr);
}

timg_hip_scaler *HipScalerAcquire(timg_hip_ctx *ctx, int in_w, int in_h, int in_fmt, int out_w, int out_h, int filter) {
    const ScalerKey key(ctx, in_w, in_h, in_fmt, out_w, out_h, filter);
    {
        std::lock_guard<std::mutex> l(g_scaler_mu);
        for (auto it = g_scaler_idle.begin(); it != g_scaler_idle.end(); ++it)
            if (it->first == key) {
                timg_hip_scaler *s = it->second;
                g_scaler_idle.erase(it);
                return s;
            }
    }
    timg_hip_scaler *s = nullptr;
    int rc = timg_hip_scaler_create(ctx, in_w, in_h, in_fmt, out_w, out_h, filter, &s);
    if (rc == TIMG_HIP_ERR_NOMEM) {  // (its tables did not fit: give back what is cached, once)
        HipPoolTrim(ctx);
        rc = timg_hip_scaler_create(ctx, in_w, in_h, in_fmt, out_w, out_h, filter, &s);
    }
    if (rc != TIMG_HIP_OK) return nullptr;
    std::lock_guard<std::mutex> l(g_scaler_mu);
    g_scaler_key[s] = key;
    return s;
}

void HipScalerRelease(timg_hip_scaler *s) {
    if (!s) return;
    timg_hip_scaler *victim = nullptr;
    {
        std::lock_guard<std::mutex> l(g_scaler_mu);
        auto it = g_scaler_key.find(s);
        if (it == g_scaler_key.end()) {
            victim = s;  // (not one of the pool's)
        } else {
            g_scaler_idle.emplace_front(it->second, s);
            if (g_scaler_idle.size() > kMaxIdleScalers) {  // the one released longest ago goes
                victim = g_scaler_idle.back().second;
                g_scaler_idle.pop_back();
                g_scaler_key.erase(victim);
            }
        }
    }
    if (victim) timg_hip_scaler_destroy(victim);
}

size_t HipIdleScalers() {
    std::lock_guard<std::mutex> l(g_scaler_mu);
    return g_scaler_idle.size();
}

void HipFatal(timg_hip_ctx *ctx, const char *what) {
    fprintf(stderr, "timg: HIP back-end failed in %s: %s\n", what, timg_hip_last_error(ctx));
    abort();
}

}  // namespace timg
