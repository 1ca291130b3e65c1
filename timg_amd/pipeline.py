"""Hot-path pipelines on device-resident frames (bench / test plumbing).

torch is used for device memory, streams and torch.distributed only; every
pixel goes through libtimg_hip.so (C-ABI calls on the current torch stream).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from .hip import Blend, TimgHip


class GridPipeline:
    """scale(+blend) -> canvas encode for a batch of equally sized frames."""

    def __init__(self, hip: TimgHip, n_frames: int, in_w: int, in_h: int, out_w: int, out_h: int,
                 mode: str = "sixel", blend: Blend | None = None, device: str = "cuda", pieces: int = 0):
        """pieces (sixel only): 0 (library's choice: 1) or 1 = scale, then encode, as two calls; > 1 = ONE fused call
        (timg_hip_scale_sixel_encode) that cuts the batch into pieces and runs a piece's scale beside the serial
        stages of the pieces in front of it."""
        self.hip, self.n = hip, n_frames
        if pieces == 0:  # (the library's own rule -- profiles/r3/fused_pieces.txt -- made explicit so that callers can report it)
            pieces = 1
        self.pieces = pieces if mode == "sixel" else 1
        self.fused = mode == "sixel" and pieces != 1
        self.last_scale_ms = None
        self.in_w, self.in_h, self.out_w, self.out_h = in_w, in_h, out_w, out_h
        self.mode = mode
        self.blend = blend
        self.scaler = hip.scaler(in_w, in_h, out_w, out_h)
        self.scaled = torch.empty((n_frames, out_h, out_w, 4), dtype=torch.uint8, device=device)
        if mode == "sixel":
            self.cap = hip.sixel_max_bytes(out_w, out_h)
        elif mode in ("kitty", "iterm2", "png"):  # graphics protocols at --compress=0
            self.cap = int(hip.L.timg_hip_gfx_max_bytes(out_w, out_h))
        else:
            self.cap = hip.block_max_bytes(out_w, out_h)
        self.out = torch.empty((n_frames, self.cap), dtype=torch.uint8, device=device)
        self.lengths = None
        # A dedicated (non-null) HIP stream: the C-ABI reads a NULL stream as
        # "the context's own stream", and torch's default stream is NULL.
        self.stream = torch.cuda.Stream(device=device)

    def stream_ptr(self):
        return self.stream.cuda_stream

    def scale(self, src: torch.Tensor):
        self.hip.scale_blend(self.scaler, src.data_ptr(), self.scaled.data_ptr(), self.n, self.blend,
                             stream=self.stream_ptr())

    def encode(self):
        st = self.stream_ptr()
        if self.mode == "sixel":
            lens = self.hip.sixel_encode(self.scaled.data_ptr(), self.out_w, self.out_h,
                                         pad_blend=self.blend, n_frames=self.n,
                                         out=self.out.data_ptr(), out_cap=self.cap, stream=st)
        elif self.mode in ("kitty", "iterm2", "png"):
            lens = self.hip.gfx_encode(self.mode, self.scaled.data_ptr(), self.out_w, self.out_h, n_frames=self.n,
                                       image_ids=range(1, self.n + 1) if self.mode == "kitty" else None,
                                       out=self.out.data_ptr(), out_cap=self.cap, stream=st)
        else:
            flags = {"half": 0, "quarter": TimgHip.QUARTER}[self.mode]
            # --grid=8x8: frame i sits in grid column i % 8, i.e. is Sent at that column's x
            # (the renderer leaves two pixel columns between images)
            xs = [(i % 8) * (self.out_w + 2) for i in range(self.n)]
            lens = self.hip.block_encode(self.scaled.data_ptr(), self.out_w, self.out_h, flags=flags,
                                         n_frames=self.n, out=self.out.data_ptr(), out_cap=self.cap,
                                         stream=st, x_indents=xs)
        self.lengths = lens
        return lens

    # -- sixel, asynchronous: a step is ENQUEUED (scale + encode + the byte counts' copy behind an event), its counts
    # are read when the caller asks for them -- usually after the next step has been enqueued, so that the device
    # never waits for the host between steps (timg_hip_sixel_encode_async; include/timg_hip.h)
    def can_async(self):
        return self.mode == "sixel" and not self.fused

    def encode_begin(self, slot: int = 0):
        assert self.can_async()
        if not hasattr(self, "_jobs"):
            self._jobs, self._outs = {}, {}
        if slot not in self._jobs:
            self._jobs[slot] = self.hip.sixel_job(self.n)
            # every job slot has its OWN output buffer: the header's contract is that `out` stays untouched until the wait
            # returns, and the next step is enqueued (into the other slot) before this one is waited for -- with one
            # buffer, finish(k) would have found step k + 1 writing over the bytes it reports
            self._outs[slot] = self.out if not self._outs else torch.empty_like(self.out)
        self.hip.sixel_encode_async(self._jobs[slot], self.scaled.data_ptr(), self.out_w, self.out_h,
                                    self._outs[slot].data_ptr(), self.cap, n_frames=self.n, pad_blend=self.blend,
                                    stream=self.stream_ptr())

    def begin(self, src: torch.Tensor, slot: int = 0):
        self.scale(src)
        self.encode_begin(slot)

    def finish(self, slot: int = 0):
        self.lengths = self.hip.sixel_encode_wait(self._jobs[slot], self.n)
        self.out = self._outs[slot]  # (frame_bytes / packed_output now speak of THIS step: its own buffer, its own counts)
        return self.lengths

    def step(self, src: torch.Tensor):
        if not self.fused:
            self.scale(src)
            return self.encode()
        self.lengths, self.last_scale_ms = self.hip.scale_sixel_encode(
            self.scaler, src.data_ptr(), self.scaled.data_ptr(), self.n, self.blend, self.out.data_ptr(), self.cap,
            pieces=self.pieces, stream=self.stream_ptr())
        return self.lengths

    def packed_output(self):
        """(payload uint8 tensor, lengths int64 tensor): frames back to back, as a SNAPSHOT -- new
        tensors, complete when this returns, so the pipeline may overwrite its buffers right away
        (run_batched_streams lets the next step start while this step's bytes are being gathered)."""
        lens = torch.tensor(self.lengths, dtype=torch.int64)
        parts = [self.out[i, :n] for i, n in enumerate(self.lengths)]
        payload, lens = torch.cat(parts), lens.to(self.out.device)
        torch.cuda.current_stream(self.out.device).synchronize()
        return payload, lens

    def frame_bytes(self, i: int) -> bytes:
        return self.out[i, :self.lengths[i]].cpu().numpy().tobytes()

    def close(self):
        for j in getattr(self, "_jobs", {}).values():
            self.hip.sixel_job_destroy(j)
        self._jobs, self._outs = {}, {}
        self.scaler.close()


class PartitionedSixelPipeline:
    """scale -> sixel encode with the two calls on TWO streams of one context (timg_hip_stream_create): the scale
    stream's kernels stay off `reserved_cus_per_xcd` CUs of every XCD, the encode stream is unrestricted and has the
    greater priority.  Step k + 1's scale kernel then runs beside the latency-bound kernels of step k's sixel chain
    (histogram, median cut, table, diffusion), which find the reserved CUs free; on one stream -- or two plain ones --
    the scale kernel holds every CU until its last workgroup retires (profiles/r6/overlap_streams.txt).  The scaled
    frames, the output and the job are double-buffered; a step's byte counts are read after the next step has been
    enqueued.  Same bytes as GridPipeline (tests/test_gpu_parity.py)."""

    def __init__(self, hip: TimgHip, n_frames: int, in_w: int, in_h: int, out_w: int, out_h: int,
                 blend: Blend | None = None, reserved_cus_per_xcd: int = 12, device: str = "cuda"):
        self.hip, self.n, self.blend = hip, n_frames, blend
        self.out_w, self.out_h = out_w, out_h
        self.reserved = reserved_cus_per_xcd
        self.scaler = hip.scaler(in_w, in_h, out_w, out_h)
        self.cap = hip.sixel_max_bytes(out_w, out_h)
        self.scaled = [torch.empty((n_frames, out_h, out_w, 4), dtype=torch.uint8, device=device) for _ in range(2)]
        self.outs = [torch.empty((n_frames, self.cap), dtype=torch.uint8, device=device) for _ in range(2)]
        self.jobs = [hip.sixel_job(n_frames) for _ in range(2)]
        self.scale_stream = hip.stream_create(reserved_cus_per_xcd=reserved_cus_per_xcd)
        self.encode_stream = hip.stream_create(high_priority=True)
        self.lengths, self.out = None, None

    def run(self, src: torch.Tensor, n_steps: int):
        """n_steps passes over `src`; returns when the last step's byte counts have been read."""
        hip = self.hip
        for k in range(n_steps):
            sl = k & 1
            # (scaled[sl] / outs[sl] / jobs[sl] were step k - 2's: that step was waited for one iteration ago)
            hip.scale_blend(self.scaler, src.data_ptr(), self.scaled[sl].data_ptr(), self.n, self.blend, stream=self.scale_stream)
            hip.stream_wait_stream(self.encode_stream, self.scale_stream)
            hip.sixel_encode_async(self.jobs[sl], self.scaled[sl].data_ptr(), self.out_w, self.out_h, self.outs[sl].data_ptr(),
                                   self.cap, n_frames=self.n, pad_blend=self.blend, stream=self.encode_stream)
            if k >= 1:
                hip.sixel_encode_wait(self.jobs[sl ^ 1], self.n)
        if n_steps > 0:
            last = (n_steps - 1) & 1
            self.lengths = hip.sixel_encode_wait(self.jobs[last], self.n)
            self.out = self.outs[last]
        hip.sync(self.scale_stream)
        hip.sync(self.encode_stream)
        return self.lengths

    def frame_bytes(self, i: int) -> bytes:
        return self.out[i, :self.lengths[i]].cpu().numpy().tobytes()

    def close(self):
        for j in self.jobs:
            self.hip.sixel_job_destroy(j)
        self.jobs = []
        self.hip.stream_destroy(self.scale_stream)
        self.hip.stream_destroy(self.encode_stream)
        self.scaler.close()


def run_batched_streams(pipes, src, n_steps, n_pipes, world=1, gather=None, timed_events=None,
                        record_event=None, async_encode=True):
    """n_steps passes of the hot path over `src`, step k on pipeline k % n_pipes, every pipeline
    driven by its own host thread (one batch in flight per pipeline).  With several ranks the
    variable-length outputs are handed to `gather(payload, lengths)` by the CALLING thread in
    step order -- collectives must be issued in the same order on every rank.  packed_output()
    returns a snapshot, so a pipeline starts its next step as soon as its previous output has been
    PACKED: the gather of step k (all-gather of lengths, payloads to the root) runs beside the
    kernels of step k + 1.

    pipes: objects with scale(src), encode() (or, when `fused`, step(src)), packed_output() (and .stream when record_event is
    given); record_event(stream) -> event, used to bracket the two stages of each step."""
    import threading
    done = [threading.Event() for _ in range(n_steps)]
    consumed = [threading.Event() for _ in range(n_steps)]
    errors = []

    def worker(i):
        try:
            p = pipes[i]
            # (with several ranks the gather needs a step's byte counts before the next step starts: synchronous there)
            use_async = async_encode and world == 1 and getattr(p, "can_async", lambda: False)()
            for k in range(i, n_steps, n_pipes):
                if world > 1 and k >= n_pipes:
                    consumed[k - n_pipes].wait()
                e0 = record_event(p.stream) if record_event else None
                if getattr(p, "fused", False):  # one call; the library times its scale kernels itself
                    p.step(src)
                    e1 = None
                elif use_async:
                    # the step is only ENQUEUED: its byte counts are read after the NEXT step of this pipeline has been
                    # enqueued behind it (two jobs alternate), so the stream never runs dry between steps
                    p.scale(src)
                    e1 = record_event(p.stream) if record_event else None
                    p.encode_begin(slot=(k // n_pipes) & 1)
                else:
                    p.scale(src)
                    e1 = record_event(p.stream) if record_event else None
                    p.encode()
                # (ONE event between two steps of a pipeline: the next step's e0 ends this step's encode stage -- an event
                # record is ~5 us of the stream's time, and a step is bracketed by three of them already; only a step
                # with no step behind it on its stream, a fused step and several pipelines record their own)
                lean = n_pipes == 1 and world == 1 and not getattr(p, "fused", False) and k + n_pipes < n_steps
                e2 = record_event(p.stream) if (record_event and not lean) else None
                if timed_events is not None and record_event:
                    timed_events.append((e0, e1, e2, getattr(p, "last_scale_ms", None)))
                if use_async:
                    if k - n_pipes >= i:
                        p.finish(slot=((k - n_pipes) // n_pipes) & 1)
                    if k + n_pipes >= n_steps:  # this pipeline's last step
                        p.finish(slot=(k // n_pipes) & 1)
                done[k].set()
        except Exception as exc:  # surface worker failures instead of hanging the gather loop
            errors.append(exc)
            for ev in done:
                ev.set()

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(min(n_pipes, n_steps))]
    for t in threads:
        t.start()
    if world > 1:
        for k in range(n_steps):
            done[k].wait()
            if errors:
                break
            payload, lens = pipes[k % n_pipes].packed_output()
            consumed[k].set()
            gather(payload, lens)
    if errors:
        for ev in consumed:  # surviving workers may be parked on consumed[k - n_pipes]: release them BEFORE joining
            ev.set()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


def synth_frames_on_device(n: int, w: int, h: int, kind: str = "photo", seed: int = 0,
                           device: str = "cuda") -> torch.Tensor:
    """Seeded synthetic RGBA8 frames generated on the device (nothing crosses
    PCIe): the S-photo / S-noise / S-alpha families of SURVEY.md 8d, torch RNG."""
    g = torch.Generator(device=device)
    g.manual_seed(0x71170000 + seed)
    out = torch.empty((n, h, w, 4), dtype=torch.uint8, device=device)
    if kind == "noise":
        out.copy_(torch.randint(0, 256, (n, h, w, 4), generator=g, device=device, dtype=torch.uint8))
        return out
    yy = torch.arange(h, device=device, dtype=torch.float32)[:, None] / h
    xx = torch.arange(w, device=device, dtype=torch.float32)[None, :] / w
    for i in range(n):
        for c in range(3):
            acc = torch.zeros((h, w), device=device)
            for _ in range(3):
                fx, fy, ph = (torch.rand(3, generator=g, device=device) * torch.tensor(
                    [5.5, 5.5, 6.2832], device=device) + torch.tensor([0.5, 0.5, 0.0], device=device)).tolist()
                acc += torch.sin(6.2832 * (fx * xx + fy * yy) + ph)
            v = (acc / 6.0 + 0.5) * 255.0
            v += torch.randn((h, w), generator=g, device=device) * (0.04 * 255.0)
            out[i, :, :, c] = v.clamp_(0, 255).to(torch.uint8)
        if kind == "alpha":
            ry = (torch.arange(h, device=device, dtype=torch.float32)[:, None] - h / 2) / (h / 2)
            rx = (torch.arange(w, device=device, dtype=torch.float32)[None, :] - w / 2) / (w / 2)
            a = ((1.0 - torch.sqrt(rx * rx + ry * ry) / 1.2).clamp_(0, 1) * 255.0).to(torch.uint8)
            b = max(1, min(64, min(w, h) // 8))
            a[:b] = 0
            a[-b:] = 0
            a[:, :b] = 0
            a[:, -b:] = 0
            pick = torch.rand((h, w), generator=g, device=device) < 0.10
            special = torch.tensor([0, 0x5F, 0x60, 0xFF], dtype=torch.uint8, device=device)[
                torch.randint(0, 4, (h, w), generator=g, device=device)]
            out[i, :, :, 3] = torch.where(pick, special, a)
        else:
            out[i, :, :, 3] = 255
    return out
