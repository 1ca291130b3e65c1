// timg_amd/twins/hip-gather-writer.h -- multi-GPU: the frames every rank encoded, on rank 0, in
// frame order, into the reference's own write sequencer.
//
// timg emits through BufferedWriteSequencer::WriteBuffer (src/buffered-write-sequencer.h:113-118),
// a FIFO in front of stdout (src/buffered-write-sequencer.cc:70-105).  With one process per GPU the
// frames of a grid / video stream are sharded over the ranks (blocks for grids, round-robin for
// streams); each rank encodes its share as one device batch, timg_hip_gather_to_root
// (include/timg_hip_comm.h: RCCL over xGMI) brings the bytes to rank 0, and this class hands them to
// the unchanged sequencer -- one WriteBuffer per frame, `new char[]` buffers, in global frame order.
#ifndef TIMG_AMD_TWINS_HIP_GATHER_WRITER_H
#define TIMG_AMD_TWINS_HIP_GATHER_WRITER_H

#include <cstdint>
#include <vector>

#include "buffered-write-sequencer.h"
#include "timg_hip.h"
#include "timg_hip_comm.h"

namespace timg {

class HipGatherWriter {
public:
    // sequencer: only used on the root (may be null elsewhere).
    HipGatherWriter(timg_hip_ctx *ctx, timg_hip_comm *comm, int world, int rank, BufferedWriteSequencer *sequencer,
                    int root = 0);
    ~HipGatherWriter();

    // Collective: every rank calls it with the frames it encoded -- `payload` (device memory) holds
    // them back to back, lengths[i] bytes each -- of a stream of n_total frames sharded as
    // timg_hip_shard_locate describes.  Returns false on a communication error (message on stderr).
    bool GatherAndWrite(const uint8_t *payload, const uint64_t *lengths, int n_local, int n_total, bool round_robin,
                        SeqType seq_type = SeqType::FrameImmediate);

private:
    timg_hip_ctx *const ctx_;
    timg_hip_comm *const comm_;
    const int world_, rank_, root_;
    BufferedWriteSequencer *const sequencer_;
    uint8_t *recv_   = nullptr;  // device
    size_t recv_cap_ = 0;
    std::vector<uint8_t> host_;
};

}  // namespace timg
#endif
