// timg_amd/twins/held-rows.h -- grid rows encoded by ONE device call without changing a byte
// of the terminal stream.
//
// MultiColumnRenderer (src/renderer.cc:81-189) issues one Send per image, and between the
// Sends other writes reach the sequencer without passing the canvas: CursorOn() after every
// image (src/terminal-canvas.cc:87-96, called from src/timg.cc:371-375).  So a canvas twin that
// wants to encode a grid row as a batch must not hold back its WriteBuffer calls -- that would
// reorder the stream.  Instead every Send hands the sequencer a FUTURE at once, in Send order,
// and the futures of a row are fulfilled together when the row is encoded: the sequencer's FIFO
// (src/buffered-write-sequencer.cc:70-105) keeps every byte where the reference puts it.
//
// A row is encoded when it is complete, when a Send arrives that cannot join it, when nothing
// has arrived for kIdle (so a partial last row never waits for a Send that does not come: in
// src/timg.cc:392 sequencer->Flush() runs before the canvas is destroyed), or when the canvas
// is destroyed -- on a thread the canvas owns, never on the caller's encoder pool, whose
// destructor drops queued work (src/thread-pool.h:43-49).
//
// How many Sends may be outstanding is bounded by the sequencer's queue: the writer thread
// blocks on the first unfulfilled future while the caller keeps queueing (each image also
// queues a CursorOn), and a full queue blocks the caller (src/buffered-write-sequencer.cc:73-78).
// HoldLimit() keeps a row's futures plus their CursorOn writes inside the queue; with the
// reference's queue of 4 (src/timg.cc:972) that is 2 images per device call -- raise the queue
// length to 2 * columns + 1 for whole rows (integration/timg-hip.patch: 4 * columns + 1, two rows -- one is
// encoded while the next is gathered).
#ifndef TIMG_AMD_TWINS_HELD_ROWS_H
#define TIMG_AMD_TWINS_HELD_ROWS_H

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <thread>
#include <vector>

#include "buffered-write-sequencer.h"
#include "timg_hip.h"

namespace timg {

struct HeldFrame {
    char *buffer   = nullptr;  // new char[]: cursor prefix in front, room for the frame behind it
    size_t prefix  = 0;
    size_t cap     = 0;
    int x          = 0;
    int dy         = 0;
    std::promise<OutBuffer> promise;
};

struct HeldBatch {
    int w = 0, h = 0;
    timg_hip_blend pad{};          // (sixel: background of the pad rows)
    std::vector<uint8_t> pixels;   // the frames, back to back (host frames)
    // frames that came from a device-resident source (hip-device-frames.h) are gathered back to
    // back in device memory instead -- at once: the source may be gone when the row is encoded
    uint8_t *dev_pixels = nullptr;
    bool on_device      = false;
    std::vector<HeldFrame> frames;
    const uint8_t *data() const { return on_device ? dev_pixels : pixels.data(); }
};

class HeldRows {
public:
    // encode(batch, ctx): runs on a worker thread; must fulfil every promise of the batch with device calls on `ctx`.
    // workers > 1 (canvases without state between frames: sixel): that many batches are encoded at the same time,
    // each worker on a context of its own (own stream and scratch) -- the per-frame serial stages of one batch run
    // beside the wide kernels of another, which is where a batched encoder leaves most of the chip idle.  The
    // futures may be fulfilled in any order: the sequencer takes them in Send order.
    HeldRows(timg_hip_ctx *ctx, std::function<void(HeldBatch &, timg_hip_ctx *)> encode, int workers = 1);
    ~HeldRows();  // encodes what is held, then joins the workers

    // Largest number of Sends one device call may cover.
    static int HoldLimit(int grid_columns, size_t sequencer_queue_len);
    // ... and the twins' own bound on it (hip-sixel-canvas.cc has the measurements): 8, or TIMG_HIP_TWIN_BATCH_CAP.
    static int BatchCap();

    // Adds a frame to the open batch (sealing an open batch of another size first) and returns
    // the future the caller hands to the sequencer.  Seals the batch when it reaches `limit`.
    // pixels: host memory, or device memory when on_device.
    std::future<OutBuffer> Hold(int w, int h, const uint8_t *pixels, bool on_device, const timg_hip_blend *pad,
                                HeldFrame &&frame, int limit);
    void Seal();   // the open batch goes to the worker now
    void Drain();  // Seal() and wait until everything handed over has been encoded

private:
    static constexpr std::chrono::milliseconds kIdle{3};
    void Work(timg_hip_ctx *worker_ctx);

    void SealLocked();
    timg_hip_ctx *const ctx_;
    const std::function<void(HeldBatch &, timg_hip_ctx *)> encode_;
    std::mutex mu_;
    std::condition_variable wake_, idle_;
    std::deque<HeldBatch> sealed_;
    HeldBatch open_;
    bool have_open_ = false;
    std::chrono::steady_clock::time_point deadline_;
    int busy_     = 0;  // batches being encoded
    bool exiting_ = false;
    std::vector<std::thread> workers_;
};

}  // namespace timg
#endif
