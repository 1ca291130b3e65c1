// timg_amd/csrc/comm.hip -- libtimg_hip_comm.so: ordered gather of encoded frames to rank 0 over
// RCCL (include/timg_hip_comm.h).  One process per GPU; the only exchange step of the path.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/timg_hip_comm.h"

struct timg_hip_comm {
    ncclComm_t nccl = nullptr;
    int device = 0, world = 1, rank = 0;
    hipStream_t stream  = nullptr;
    uint64_t *len_dev   = nullptr;  // [world * cap_frames] all-gathered lengths
    uint64_t *len_mine  = nullptr;  // [cap_frames]
    int cap_frames      = 0;
    std::string last_error;
    int Fail(const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        last_error = buf;
        return -1;
    }
};

static std::string g_last_error;

#define COMM_HIP(c, expr)                                                       \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) return (c)->Fail("%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define COMM_NCCL(c, expr)                                                       \
    do {                                                                         \
        ncclResult_t _r = (expr);                                                \
        if (_r != ncclSuccess) return (c)->Fail("%s: %s", #expr, ncclGetErrorString(_r)); \
    } while (0)

extern "C" {

int timg_hip_comm_unique_id(uint8_t *id) {
    static_assert(sizeof(ncclUniqueId) == TIMG_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id) return -1;
    ncclUniqueId u;
    ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) {
        g_last_error = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r);
        return -1;
    }
    memcpy(id, &u, sizeof(u));
    return 0;
}

int timg_hip_comm_create(int device, int world, int rank, const uint8_t *id, timg_hip_comm **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return -1;
    timg_hip_comm *c = new timg_hip_comm();
    c->device = device;
    c->world  = world;
    c->rank   = rank;
    auto fail = [&](const std::string &m) {
        g_last_error = m;
        delete c;
        return -1;
    };
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r = ncclCommInitRank(&c->nccl, world, u, rank);
    if (r != ncclSuccess) return fail(std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    *out = c;
    return 0;
}

void timg_hip_comm_destroy(timg_hip_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    if (c->len_dev) (void)hipFree(c->len_dev);
    if (c->len_mine) (void)hipFree(c->len_mine);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *timg_hip_comm_last_error(const timg_hip_comm *c) { return c ? c->last_error.c_str() : g_last_error.c_str(); }

int timg_hip_gather_to_root(timg_hip_comm *c, int root, const uint8_t *payload, const uint64_t *lengths, int n_frames,
                            int n_frames_max, uint64_t *all_lengths, uint8_t *recv, size_t recv_cap,
                            size_t *recv_bytes, void *stream) {
    if (!c || root < 0 || root >= c->world || n_frames < 0 || n_frames > n_frames_max || n_frames_max < 1 ||
        (n_frames && (!lengths || !payload)))
        return c ? c->Fail("bad argument") : -1;
    if (c->rank == root && (!all_lengths || !recv || !recv_bytes)) return c->Fail("the root needs its output buffers");
    COMM_HIP(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (n_frames_max > c->cap_frames) {
        if (c->len_dev) (void)hipFree(c->len_dev);
        if (c->len_mine) (void)hipFree(c->len_mine);
        c->len_dev = c->len_mine = nullptr;
        c->cap_frames            = 0;
        COMM_HIP(c, hipMalloc((void **)&c->len_dev, sizeof(uint64_t) * (size_t)c->world * n_frames_max));
        COMM_HIP(c, hipMalloc((void **)&c->len_mine, sizeof(uint64_t) * (size_t)n_frames_max));
        c->cap_frames = n_frames_max;
    }
    // 1. byte counts of every rank's frames
    std::vector<uint64_t> mine((size_t)n_frames_max, 0);
    size_t my_bytes = 0;
    for (int i = 0; i < n_frames; ++i) {
        mine[i] = lengths[i];
        my_bytes += (size_t)lengths[i];
    }
    COMM_HIP(c, hipMemcpyAsync(c->len_mine, mine.data(), sizeof(uint64_t) * n_frames_max, hipMemcpyHostToDevice, st));
    COMM_NCCL(c, ncclAllGather(c->len_mine, c->len_dev, (size_t)n_frames_max, ncclUint64, c->nccl, st));
    std::vector<uint64_t> all((size_t)c->world * n_frames_max);
    COMM_HIP(c, hipMemcpyAsync(all.data(), c->len_dev, sizeof(uint64_t) * all.size(), hipMemcpyDeviceToHost, st));
    COMM_HIP(c, hipStreamSynchronize(st));
    // 2. payloads: one message per peer into the root (each peer has its own xGMI link)
    std::vector<size_t> total((size_t)c->world, 0), offset((size_t)c->world, 0);
    size_t sum = 0;
    for (int r = 0; r < c->world; ++r) {
        for (int i = 0; i < n_frames_max; ++i) total[r] += (size_t)all[(size_t)r * n_frames_max + i];
        offset[r] = sum;
        sum += total[r];
    }
    if (c->rank == root) {
        if (sum > recv_cap) return c->Fail("gathered frames need %zu bytes, recv_cap is %zu", sum, recv_cap);
        memcpy(all_lengths, all.data(), sizeof(uint64_t) * all.size());
        *recv_bytes = sum;
        if (my_bytes)
            COMM_HIP(c, hipMemcpyAsync(recv + offset[root], payload, my_bytes, hipMemcpyDeviceToDevice, st));
        COMM_NCCL(c, ncclGroupStart());
        for (int r = 0; r < c->world; ++r)
            if (r != root && total[r]) COMM_NCCL(c, ncclRecv(recv + offset[r], total[r], ncclUint8, r, c->nccl, st));
        COMM_NCCL(c, ncclGroupEnd());
    } else if (my_bytes) {
        COMM_NCCL(c, ncclGroupStart());
        COMM_NCCL(c, ncclSend(payload, my_bytes, ncclUint8, root, c->nccl, st));
        COMM_NCCL(c, ncclGroupEnd());
    }
    COMM_HIP(c, hipStreamSynchronize(st));
    return 0;
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
