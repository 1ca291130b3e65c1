/* include/timg_hip_comm.h -- the one exchange step of the path, behind a C-ABI.
 *
 * hzeller/timg is a single process: its only "collective" is the in-order FIFO in front of
 * stdout, BufferedWriteSequencer (src/buffered-write-sequencer.h:113-118 WriteBuffer, the FIFO
 * at src/buffered-write-sequencer.cc:70-105).  When a grid or a video stream is sharded over
 * several GPUs (one process per GPU, frames are independent) that FIFO needs the
 * encoded frames of every rank on rank 0, in frame order.  This library is that gather:
 *
 *     byte counts : ncclAllGather of the per-frame lengths
 *     payload     : grouped ncclSend (every rank != root) / ncclRecv (root), one message per peer --
 *                   RCCL has no gatherv, and over xGMI every peer has its own link into the root
 *
 * It is a separate shared object (libtimg_hip_comm.so, links librccl) so that the single-GPU
 * library carries no RCCL dependency.  The communicator is bootstrapped like every NCCL/RCCL
 * program: rank 0 creates a unique id, the host program hands those 128 bytes to the other
 * ranks by whatever means it has (environment, file, MPI, torch.distributed), every rank calls
 * timg_hip_comm_create.  Plain C types only; every function returns 0 or a negative status
 * (timg_hip_comm_last_error has the text).
 *
 * The C++ side that feeds the gathered frames to the reference's sequencer in frame order:
 * timg_amd/twins/hip-gather-writer.h. */
#ifndef TIMG_HIP_COMM_H
#define TIMG_HIP_COMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TIMG_HIP_COMM_ID_BYTES 128
typedef struct timg_hip_comm timg_hip_comm;

/* rank 0: fills id[TIMG_HIP_COMM_ID_BYTES] (ncclGetUniqueId). */
int timg_hip_comm_unique_id(uint8_t *id);
/* every rank: device = the GPU this process drives. */
int timg_hip_comm_create(int device, int world, int rank, const uint8_t *id, timg_hip_comm **out);
void timg_hip_comm_destroy(timg_hip_comm *comm);
const char *timg_hip_comm_last_error(const timg_hip_comm *comm); /* comm may be NULL */

/* Gathers this rank's n_frames encoded frames -- `payload` holds them back to back in DEVICE
 * memory, lengths[i] bytes each (host array) -- to `root`.  Every rank must pass the same
 * n_frames_max >= its own n_frames (ranks may own different numbers of frames).
 *
 * On the root, on return:
 *   all_lengths[r * n_frames_max + i]  length of frame i of rank r (0 beyond that rank's count)
 *   recv (DEVICE memory, recv_cap bytes) the payloads of ranks 0 .. world-1 back to back in rank order
 *   *recv_bytes                          their total
 * Other ranks may pass NULL for all_lengths / recv / recv_bytes.  `stream` (hipStream_t or NULL):
 * the collective is enqueued on it and the call returns after synchronising it. */
int timg_hip_gather_to_root(timg_hip_comm *comm, int root, const uint8_t *payload, const uint64_t *lengths,
                            int n_frames, int n_frames_max, uint64_t *all_lengths, uint8_t *recv,
                            size_t recv_cap, size_t *recv_bytes, void *stream);

/* Which (rank, index-on-that-rank) holds global frame f of n_total frames sharded over `world`
 * ranks: round_robin != 0 -> frame f on rank f % world (video streams), else contiguous blocks of
 * ceil(n_total / world) frames (grids).  The inverse of the sharding bench.py and the host
 * programs use; pure arithmetic (no communicator needed). */
void timg_hip_shard_locate(int n_total, int world, int round_robin, int frame, int *rank, int *index);
int timg_hip_shard_count(int n_total, int world, int round_robin, int rank);

#ifdef __cplusplus
}
#endif
#endif /* TIMG_HIP_COMM_H */
