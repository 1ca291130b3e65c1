/* include/timg_hip_comm.h -- the one exchange step of the path, behind a C-ABI.
 *
 * hzeller/timg is a single process: its only "collective" is the in-order FIFO in front of
 * stdout, BufferedWriteSequencer (src/buffered-write-sequencer.h:113-118 WriteBuffer, the FIFO
 * at src/buffered-write-sequencer.cc:70-105).  When a grid or a video stream is sharded over
 * several GPUs (one process per GPU, frames are independent) that FIFO needs the
 * encoded frames of every rank on rank 0, in frame order.  This library is that gather:
 *
 *     byte counts : ncclAllGather of the per-frame lengths      (timg_hip_gather_lengths)
 *     payload     : grouped ncclSend (every rank != root) / ncclRecv (root), one message per peer --
 *                   RCCL has no gatherv, and over xGMI every peer has its own link into the root
 *                                                                (timg_hip_gather_payload)
 *
 * It is a separate shared object (libtimg_hip_comm.so) so that the single-GPU library carries no
 * RCCL dependency.  RCCL itself is NOT linked: the library binds the RCCL the process already
 * maps (a host program that brings its own -- e.g. one that also holds PyTorch -- gets exactly
 * that one, never a second copy), and only when none is mapped loads the file TIMG_HIP_RCCL_LIB
 * names, else librccl.so.1 by the usual search path, then /opt/rocm/lib.  timg_hip_comm_rccl_info
 * says which library answered.
 * The communicator is bootstrapped like every NCCL/RCCL program: rank 0 creates a unique id, the
 * host program hands those 128 bytes to the other ranks by whatever means it has (environment,
 * file, MPI, torch.distributed), every rank calls timg_hip_comm_create.  Plain C types only; every
 * function returns 0 or a negative status (timg_hip_comm_last_error has the text).
 *
 * Collective discipline: the gather calls are collectives -- every rank of the communicator makes
 * the same call with the same root / n_frames_max.  Data-dependent failures (a receive buffer that
 * is too small) are detected from all-gathered values, so EVERY rank returns the same error and
 * none is left waiting in a send.  Argument errors (NULL pointers, counts out of range) are
 * rejected before any rank can enter the exchange; they are the caller's bug on that rank.
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
#define TIMG_HIP_COMM_ERR (-1)        /* argument / RCCL / HIP error: see timg_hip_comm_last_error */
#define TIMG_HIP_COMM_ERR_NO_RCCL (-2) /* no usable librccl in this process or on the search path */
#define TIMG_HIP_COMM_ERR_CAP (-3)     /* the root's receive buffer is too small (every rank returns it) */
typedef struct timg_hip_comm timg_hip_comm;

/* Which RCCL serves this library: path of the shared object (as the dynamic loader reports it) and
 * ncclGetVersion's number.  Binds RCCL on first use; TIMG_HIP_COMM_ERR_NO_RCCL when there is none. */
int timg_hip_comm_rccl_info(char *path, size_t path_cap, int *version);

/* rank 0: fills id[TIMG_HIP_COMM_ID_BYTES] (ncclGetUniqueId). */
int timg_hip_comm_unique_id(uint8_t *id);
/* every rank: device = the GPU this process drives. */
int timg_hip_comm_create(int device, int world, int rank, const uint8_t *id, timg_hip_comm **out);
void timg_hip_comm_destroy(timg_hip_comm *comm);
const char *timg_hip_comm_last_error(const timg_hip_comm *comm); /* comm may be NULL */

/* Step 1 (collective): the byte counts.  This rank owns n_frames frames of lengths[i] bytes (host
 * array); every rank passes the same n_frames_max >= its own n_frames (ranks may own different
 * numbers of frames).  On return, on EVERY rank:
 *   all_lengths[r * n_frames_max + i]  length of frame i of rank r (0 beyond that rank's count)
 * so the root can size its receive buffer exactly before step 2. */
int timg_hip_gather_lengths(timg_hip_comm *comm, const uint64_t *lengths, int n_frames, int n_frames_max,
                            uint64_t *all_lengths, void *stream);

/* Step 2 (collective): the payload.  `payload` holds this rank's frames back to back in DEVICE
 * memory; all_lengths is step 1's result.  On the root, on return, recv (DEVICE memory, recv_cap
 * bytes) holds the payloads of ranks 0 .. world-1 back to back in rank order and *recv_bytes their
 * total.  Other ranks pass recv = NULL, recv_cap = 0.  The root's recv_cap travels with a one-word
 * all-gather first: when it is too small EVERY rank returns TIMG_HIP_COMM_ERR_CAP and no payload
 * moves.  `stream`: the hipStream_t the payload was produced on -- the exchange is enqueued behind it;
 * TIMG_HIP_COMM_PAYLOAD_READY: the caller has already waited for whatever produced the payload (a snapshot it
 * synchronised, bench.py's packed_output) -- the exchange runs on the communicator's own stream BESIDE whatever the
 * caller's other streams are doing (the next step's kernels), no device-wide wait;
 * NULL: the last resort for a caller that knows nothing about its producer -- the communicator's own stream, after
 * the whole device has gone idle (the payload is complete whatever stream produced it; everything else the device
 * was doing is waited for too). */
#define TIMG_HIP_COMM_PAYLOAD_READY ((void *)(uintptr_t)1)
int timg_hip_gather_payload(timg_hip_comm *comm, int root, const uint8_t *payload, const uint64_t *all_lengths,
                            int n_frames_max, uint8_t *recv, size_t recv_cap, size_t *recv_bytes, void *stream);

/* Both steps in one call, for callers whose receive buffer has a known bound.  all_lengths may be
 * NULL on ranks other than the root.  `stream` (hipStream_t or NULL): the exchange is enqueued on
 * it and the call returns after synchronising it. */
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
