// timg_amd/csrc/comm.hip -- libtimg_hip_comm.so: ordered gather of encoded frames to rank 0 over
// RCCL (include/timg_hip_comm.h).  One process per GPU; the only exchange step of the path.
//
// RCCL is bound at run time, never linked: a process that already maps an RCCL (a host program with
// its own copy, e.g. the one PyTorch bundles) must be served by THAT one -- two RCCLs in one process
// each bring their own kernels, proxies and HIP state.  rccl.h is used for its types only.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/timg_hip_comm.h"

namespace {

struct RcclApi {
    void *handle = nullptr;
    std::string path, how;
    int version = 0;
    ncclResult_t (*GetVersion)(int *)                                                                  = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *)                                                        = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int)                                 = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t)                                                            = nullptr;
    const char *(*GetErrorString)(ncclResult_t)                                                        = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t)   = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)           = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)                 = nullptr;
    ncclResult_t (*GroupStart)()                                                                       = nullptr;
    ncclResult_t (*GroupEnd)()                                                                         = nullptr;
};

std::once_flag g_bind_once;
RcclApi g_rccl;
std::string g_bind_error;

template <class F>
bool Sym(void *h, const char *name, F *out) {
    *out = reinterpret_cast<F>(dlsym(h, name));
    return *out != nullptr;
}

void BindRccl() {
    // 1. the RCCL this process already maps (RTLD_NOLOAD: look, do not load)
    static const char *kNames[] = {"librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : kNames) {
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) {
            g_rccl.how = "already mapped by the process";
            break;
        }
    }
    // 2. the file the deployment names (TIMG_HIP_RCCL_LIB=/path/to/librccl.so): a process that maps no RCCL yet can be
    //    pointed at the copy its neighbours already use -- ROCm's own librccl.so.1 is 570 MB, and the first process to
    //    map it on a box whose image is still paging in waits minutes for it (tests/test_twins.py names PyTorch's)
    if (!h) {
        const char *named = getenv("TIMG_HIP_RCCL_LIB");
        if (named && *named && (h = dlopen(named, RTLD_NOW | RTLD_LOCAL))) g_rccl.how = std::string("loaded as ") + named;
    }
    // 3. none: the loader's search path, then ROCm's default prefix
    if (!h) {
        static const char *kLoad[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : kLoad) {
            if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) {
                g_rccl.how = std::string("loaded as ") + n;
                break;
            }
        }
    }
    if (!h) {
        const char *e = dlerror();
        g_bind_error  = std::string("no librccl: ") + (e ? e : "dlopen failed");
        return;
    }
    RcclApi &a = g_rccl;
    const bool ok = Sym(h, "ncclGetVersion", &a.GetVersion) && Sym(h, "ncclGetUniqueId", &a.GetUniqueId) &&
                    Sym(h, "ncclCommInitRank", &a.CommInitRank) && Sym(h, "ncclCommDestroy", &a.CommDestroy) &&
                    Sym(h, "ncclGetErrorString", &a.GetErrorString) && Sym(h, "ncclAllGather", &a.AllGather) &&
                    Sym(h, "ncclSend", &a.Send) && Sym(h, "ncclRecv", &a.Recv) &&
                    Sym(h, "ncclGroupStart", &a.GroupStart) && Sym(h, "ncclGroupEnd", &a.GroupEnd);
    if (!ok) {
        g_bind_error = "librccl lacks a needed entry point";
        dlclose(h);
        return;
    }
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(a.GetVersion), &info) && info.dli_fname) a.path = info.dli_fname;
    (void)a.GetVersion(&a.version);
    a.handle = h;
}

const RcclApi *Rccl() {
    std::call_once(g_bind_once, BindRccl);
    return g_rccl.handle ? &g_rccl : nullptr;
}

std::mutex g_err_mu;
std::string g_last_error;
void SetGlobalError(const std::string &m) {
    std::lock_guard<std::mutex> l(g_err_mu);
    g_last_error = m;
}

}  // namespace

struct timg_hip_comm {
    const RcclApi *api = nullptr;
    ncclComm_t nccl    = nullptr;
    int device = 0, world = 1, rank = 0;
    hipStream_t stream  = nullptr;
    uint64_t *len_dev   = nullptr;  // [world * cap_words] all-gather destination
    uint64_t *len_mine  = nullptr;  // [cap_words]
    size_t cap_words    = 0;
    std::string last_error;
    int Fail(int code, const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        last_error = buf;
        return code;
    }
    // words of this rank -> all[world * n] on the host of every rank
    int AllGatherWords(const uint64_t *mine, size_t n, uint64_t *all, hipStream_t st);
};

#define COMM_HIP(c, expr)                                                                           \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) return (c)->Fail(TIMG_HIP_COMM_ERR, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define COMM_NCCL(c, expr)                                                                                   \
    do {                                                                                                     \
        ncclResult_t _r = (expr);                                                                            \
        if (_r != ncclSuccess) return (c)->Fail(TIMG_HIP_COMM_ERR, "%s: %s", #expr, (c)->api->GetErrorString(_r)); \
    } while (0)

int timg_hip_comm::AllGatherWords(const uint64_t *mine, size_t n, uint64_t *all, hipStream_t st) {
    if (n > cap_words) {
        if (len_dev) (void)hipFree(len_dev);
        if (len_mine) (void)hipFree(len_mine);
        len_dev = len_mine = nullptr;
        cap_words          = 0;
        COMM_HIP(this, hipMalloc((void **)&len_dev, sizeof(uint64_t) * (size_t)world * n));
        COMM_HIP(this, hipMalloc((void **)&len_mine, sizeof(uint64_t) * n));
        cap_words = n;
    }
    // (`mine` is pageable host memory of the caller: a synchronous copy, the runtime stages it)
    COMM_HIP(this, hipStreamSynchronize(st));
    COMM_HIP(this, hipMemcpy(len_mine, mine, sizeof(uint64_t) * n, hipMemcpyHostToDevice));
    COMM_NCCL(this, api->AllGather(len_mine, len_dev, n, ncclUint64, nccl, st));
    COMM_HIP(this, hipStreamSynchronize(st));
    COMM_HIP(this, hipMemcpy(all, len_dev, sizeof(uint64_t) * (size_t)world * n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" {

int timg_hip_comm_rccl_info(char *path, size_t path_cap, int *version) {
    const RcclApi *a = Rccl();
    if (!a) {
        SetGlobalError(g_bind_error);
        return TIMG_HIP_COMM_ERR_NO_RCCL;
    }
    if (path && path_cap) snprintf(path, path_cap, "%s (%s)", a->path.c_str(), a->how.c_str());
    if (version) *version = a->version;
    return 0;
}

int timg_hip_comm_unique_id(uint8_t *id) {
    static_assert(sizeof(ncclUniqueId) == TIMG_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id) return TIMG_HIP_COMM_ERR;
    const RcclApi *a = Rccl();
    if (!a) {
        SetGlobalError(g_bind_error);
        return TIMG_HIP_COMM_ERR_NO_RCCL;
    }
    ncclUniqueId u;
    ncclResult_t r = a->GetUniqueId(&u);
    if (r != ncclSuccess) {
        SetGlobalError(std::string("ncclGetUniqueId: ") + a->GetErrorString(r));
        return TIMG_HIP_COMM_ERR;
    }
    memcpy(id, &u, sizeof(u));
    return 0;
}

int timg_hip_comm_create(int device, int world, int rank, const uint8_t *id, timg_hip_comm **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return TIMG_HIP_COMM_ERR;
    *out             = nullptr;
    const RcclApi *a = Rccl();
    if (!a) {
        SetGlobalError(g_bind_error);
        return TIMG_HIP_COMM_ERR_NO_RCCL;
    }
    timg_hip_comm *c = new timg_hip_comm();
    c->api    = a;
    c->device = device;
    c->world  = world;
    c->rank   = rank;
    auto fail = [&](const std::string &m) {
        SetGlobalError(m);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return TIMG_HIP_COMM_ERR;
    };
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail(std::string("hipSetDevice: ") + hipGetErrorString(e));
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        c->stream = nullptr;
        return fail(std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r = a->CommInitRank(&c->nccl, world, u, rank);
    if (r != ncclSuccess) return fail(std::string("ncclCommInitRank: ") + a->GetErrorString(r));
    *out = c;
    return 0;
}

void timg_hip_comm_destroy(timg_hip_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->nccl) (void)c->api->CommDestroy(c->nccl);
    if (c->len_dev) (void)hipFree(c->len_dev);
    if (c->len_mine) (void)hipFree(c->len_mine);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *timg_hip_comm_last_error(const timg_hip_comm *c) {
    if (c) return c->last_error.c_str();
    static thread_local std::string copy;
    std::lock_guard<std::mutex> l(g_err_mu);
    copy = g_last_error;
    return copy.c_str();
}

int timg_hip_gather_lengths(timg_hip_comm *c, const uint64_t *lengths, int n_frames, int n_frames_max,
                            uint64_t *all_lengths, void *stream) {
    if (!c) return TIMG_HIP_COMM_ERR;
    if (n_frames < 0 || n_frames_max < 1 || n_frames > n_frames_max || (n_frames && !lengths) || !all_lengths)
        return c->Fail(TIMG_HIP_COMM_ERR, "bad argument");
    COMM_HIP(c, hipSetDevice(c->device));
    hipStream_t st = (stream && stream != TIMG_HIP_COMM_PAYLOAD_READY) ? (hipStream_t)stream : c->stream;
    std::vector<uint64_t> mine((size_t)n_frames_max, 0);
    for (int i = 0; i < n_frames; ++i) mine[i] = lengths[i];
    return c->AllGatherWords(mine.data(), mine.size(), all_lengths, st);
}

int timg_hip_gather_payload(timg_hip_comm *c, int root, const uint8_t *payload, const uint64_t *all_lengths,
                            int n_frames_max, uint8_t *recv, size_t recv_cap, size_t *recv_bytes, void *stream) {
    if (!c) return TIMG_HIP_COMM_ERR;
    if (root < 0 || root >= c->world || n_frames_max < 1 || !all_lengths)
        return c->Fail(TIMG_HIP_COMM_ERR, "bad argument");
    if (c->rank == root && !recv_bytes) return c->Fail(TIMG_HIP_COMM_ERR, "the root needs recv_bytes");
    std::vector<size_t> total((size_t)c->world, 0), offset((size_t)c->world, 0);
    size_t sum = 0;
    for (int r = 0; r < c->world; ++r) {
        for (int i = 0; i < n_frames_max; ++i) total[r] += (size_t)all_lengths[(size_t)r * n_frames_max + i];
        offset[r] = sum;
        sum += total[r];
    }
    const size_t my_bytes = total[c->rank];
    if (my_bytes && !payload) return c->Fail(TIMG_HIP_COMM_ERR, "payload is NULL");
    COMM_HIP(c, hipSetDevice(c->device));
    const bool ready = stream == TIMG_HIP_COMM_PAYLOAD_READY;
    hipStream_t st   = (stream && !ready) ? (hipStream_t)stream : c->stream;
    // stream == NULL: the exchange runs on the communicator's own stream, which knows nothing of the stream(s) the
    // caller produced `payload` on -- the device is made idle first, so that the payload is complete whatever the
    // caller did or did not synchronise (the call ends with a synchronisation anyway).  A caller that passes its
    // producing stream gets stream order instead, and one that says TIMG_HIP_COMM_PAYLOAD_READY (it has waited for its
    // producer itself) runs beside the kernels of its other streams: neither pays a device-wide wait.
    if (!stream) COMM_HIP(c, hipDeviceSynchronize());
    // every rank learns the root's capacity: a short buffer fails everywhere, nobody waits in a send
    const uint64_t my_cap = (c->rank == root && recv) ? (uint64_t)recv_cap : 0;
    std::vector<uint64_t> caps((size_t)c->world, 0);
    if (c->world > 1) {
        int rc = c->AllGatherWords(&my_cap, 1, caps.data(), st);
        if (rc != 0) return rc;
    } else {
        caps[0] = my_cap;
    }
    if (sum > caps[root])
        return c->Fail(TIMG_HIP_COMM_ERR_CAP, "gathered frames need %zu bytes, the root's recv_cap is %llu", sum,
                       (unsigned long long)caps[root]);
    if (c->rank == root) {
        *recv_bytes = sum;
        if (my_bytes)
            COMM_HIP(c, hipMemcpyAsync(recv + offset[root], payload, my_bytes, hipMemcpyDeviceToDevice, st));
        bool any = false;
        for (int r = 0; r < c->world; ++r) any = any || (r != root && total[r]);
        if (any) {
            COMM_NCCL(c, c->api->GroupStart());
            for (int r = 0; r < c->world; ++r)
                if (r != root && total[r])
                    COMM_NCCL(c, c->api->Recv(recv + offset[r], total[r], ncclUint8, r, c->nccl, st));
            COMM_NCCL(c, c->api->GroupEnd());
        }
    } else if (my_bytes) {
        COMM_NCCL(c, c->api->GroupStart());
        COMM_NCCL(c, c->api->Send(payload, my_bytes, ncclUint8, root, c->nccl, st));
        COMM_NCCL(c, c->api->GroupEnd());
    }
    COMM_HIP(c, hipStreamSynchronize(st));
    return 0;
}

int timg_hip_gather_to_root(timg_hip_comm *c, int root, const uint8_t *payload, const uint64_t *lengths, int n_frames,
                            int n_frames_max, uint64_t *all_lengths, uint8_t *recv, size_t recv_cap,
                            size_t *recv_bytes, void *stream) {
    if (!c) return TIMG_HIP_COMM_ERR;
    if (root < 0 || root >= c->world || n_frames < 0 || n_frames > n_frames_max || n_frames_max < 1 ||
        (n_frames && (!lengths || !payload)))
        return c->Fail(TIMG_HIP_COMM_ERR, "bad argument");
    if (c->rank == root && (!all_lengths || !recv || !recv_bytes))
        return c->Fail(TIMG_HIP_COMM_ERR, "the root needs its output buffers");
    std::vector<uint64_t> scratch;
    if (!all_lengths) {
        scratch.resize((size_t)c->world * n_frames_max);
        all_lengths = scratch.data();
    }
    int rc = timg_hip_gather_lengths(c, lengths, n_frames, n_frames_max, all_lengths, stream);
    if (rc != 0) return rc;
    return timg_hip_gather_payload(c, root, payload, all_lengths, n_frames_max, recv, recv_cap, recv_bytes, stream);
}

int timg_hip_shard_count(int n_total, int world, int round_robin, int rank) {
    if (n_total <= 0 || world <= 0 || rank < 0 || rank >= world) return 0;
    if (round_robin) return (n_total - rank + world - 1) / world;
    const int per = (n_total + world - 1) / world;
    const int lo = rank * per < n_total ? rank * per : n_total, hi = (rank + 1) * per < n_total ? (rank + 1) * per : n_total;
    return hi - lo;
}

void timg_hip_shard_locate(int n_total, int world, int round_robin, int frame, int *rank, int *index) {
    int r = 0, i = 0;
    if (n_total > 0 && world > 0 && frame >= 0 && frame < n_total) {
        if (round_robin) {
            r = frame % world;
            i = frame / world;
        } else {
            const int per = (n_total + world - 1) / world;
            r             = frame / per;
            i             = frame % per;
        }
    }
    if (rank) *rank = r;
    if (index) *index = i;
}

}  // extern "C"
