"""Ordered gather of variable-length escape-sequence buffers to rank 0.

Frames are independent, so a grid / video stream shards one block of frames
per GPU with no collective on the data path; the only exchange step is the
one the reference already has as an in-order FIFO in front of stdout
(src/buffered-write-sequencer.cc:70-79).  Across GPUs it becomes: all-gather of
per-frame byte counts, then every non-root rank sends its concatenated payload
to rank 0 (RCCL has no gatherv; point-to-point over xGMI gives each peer its own
link into rank 0).  Backend-agnostic torch.distributed calls: `nccl` (= RCCL) on
GPUs, `gloo` in the CPU tests.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def gather_frames_to_root(payload: torch.Tensor, lengths: torch.Tensor, root: int = 0
                          ) -> Optional[List[List[torch.Tensor]]]:
    """payload: uint8 tensor holding this rank's frames back to back (only the
    first lengths.sum() bytes are meaningful); lengths: int64[n_frames] byte
    count per frame.  Returns, on root, result[rank][frame] -> uint8 tensor in
    frame order; None elsewhere.  Every rank must hold the same n_frames."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lengths = lengths.to(torch.int64)
    if world > 1 and dist.get_backend() == "gloo" and payload.is_cuda:
        # (gloo has no point-to-point for device tensors: through host memory -- a testing configuration)
        payload, lengths = payload.cpu(), lengths.cpu()
    if world == 1:
        offs = torch.cumsum(lengths, 0) - lengths
        return [[payload[int(o):int(o) + int(n)] for o, n in zip(offs.tolist(), lengths.tolist())]]
    # (flat output: gloo's allgather_base insists on it, RCCL accepts either)
    all_len = torch.empty(world * lengths.numel(), dtype=torch.int64, device=lengths.device)
    dist.all_gather_into_tensor(all_len, lengths.contiguous())
    all_len = all_len.view(world, lengths.numel())
    totals = all_len.sum(1).tolist()
    if rank != root:
        dist.send(payload[:int(totals[rank])].contiguous(), dst=root)
        return None
    bufs, reqs = [], []
    for r in range(world):
        if r == root:
            bufs.append(payload[:int(totals[r])])
            continue
        b = torch.empty(int(totals[r]), dtype=torch.uint8, device=payload.device)
        reqs.append(dist.irecv(b, src=r))
        bufs.append(b)
    for q in reqs:
        q.wait()
    out = []
    for r in range(world):
        ln = all_len[r].tolist()
        off, frames = 0, []
        for n in ln:
            frames.append(bufs[r][off:off + n])
            off += n
        out.append(frames)
    return out


def shard_frames(n_frames: int, world: int, rank: int, round_robin: bool = False) -> List[int]:
    """Frame indices a rank owns: contiguous blocks for grids (keeps a grid row
    on one GPU), round-robin for video streams (uniform latency)."""
    if round_robin:
        return list(range(rank, n_frames, world))
    per = (n_frames + world - 1) // world
    return list(range(min(rank * per, n_frames), min((rank + 1) * per, n_frames)))
