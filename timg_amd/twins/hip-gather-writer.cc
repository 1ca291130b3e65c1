#include "hip-gather-writer.h"

#include <cstdio>
#include <cstring>

namespace timg {

HipGatherWriter::HipGatherWriter(timg_hip_ctx *ctx, timg_hip_comm *comm, int world, int rank,
                                 BufferedWriteSequencer *sequencer, int root)
    : ctx_(ctx), comm_(comm), world_(world), rank_(rank), root_(root), sequencer_(sequencer) {}

HipGatherWriter::~HipGatherWriter() {
    if (recv_) (void)timg_hip_free(ctx_, recv_);
}

bool HipGatherWriter::GatherAndWrite(const uint8_t *payload, const uint64_t *lengths, int n_local, int n_total,
                                     bool round_robin, SeqType seq_type) {
    int n_max = 0;
    for (int r = 0; r < world_; ++r) {
        const int n = timg_hip_shard_count(n_total, world_, round_robin, r);
        if (n > n_max) n_max = n;
    }
    if (n_max < 1) return true;
    // step 1: the byte counts of every rank's frames, on every rank -- the root sizes its buffer exactly
    std::vector<uint64_t> all((size_t)world_ * n_max);
    if (timg_hip_gather_lengths(comm_, lengths, n_local, n_max, all.data(), nullptr) != 0) {
        fprintf(stderr, "timg: gather of the frame lengths failed: %s\n", timg_hip_comm_last_error(comm_));
        return false;
    }
    size_t got = 0;
    if (rank_ == root_) {
        size_t want = 0;
        for (uint64_t v : all) want += (size_t)v;
        if (want > recv_cap_) {
            if (recv_) (void)timg_hip_free(ctx_, recv_);
            recv_     = nullptr;
            recv_cap_ = 0;
            want += want / 4 + 4096;  // (headroom: the next call usually fits)
            // an allocation failure must not leave the peers waiting: enter step 2 with capacity 0,
            // every rank then returns TIMG_HIP_COMM_ERR_CAP
            if (timg_hip_malloc(ctx_, want, (void **)&recv_) == TIMG_HIP_OK) recv_cap_ = want;
        }
    }
    // step 2: the payloads
    if (timg_hip_gather_payload(comm_, root_, payload, all.data(), n_max, recv_, recv_cap_, &got, nullptr) != 0) {
        fprintf(stderr, "timg: gather of the encoded frames failed: %s\n", timg_hip_comm_last_error(comm_));
        return false;
    }
    if (rank_ != root_) return true;
    host_.resize(got);
    if (got && timg_hip_memcpy_d2h(ctx_, host_.data(), recv_, got, nullptr) != TIMG_HIP_OK) return false;
    // where every frame of every rank starts
    std::vector<size_t> start((size_t)world_ * n_max);
    size_t at = 0;
    for (size_t i = 0; i < start.size(); ++i) {
        start[i] = at;
        at += (size_t)all[i];
    }
    for (int f = 0; f < n_total; ++f) {  // global frame order: what the FIFO in front of stdout needs
        int r = 0, i = 0;
        timg_hip_shard_locate(n_total, world_, round_robin, f, &r, &i);
        const size_t slot = (size_t)r * n_max + i, len = (size_t)all[slot];
        if (len == 0) continue;  // (the sequencer does not take empty buffers: src/buffered-write-sequencer.cc:84)
        char *data = new char[len];  // freed with delete[] by the writer thread (buffered-write-sequencer.h:39)
        memcpy(data, host_.data() + start[slot], len);
        sequencer_->WriteBuffer(OutBuffer(data, len), seq_type);
    }
    return true;
}

}  // namespace timg
