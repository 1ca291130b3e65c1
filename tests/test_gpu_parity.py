"""-m gpu: the HIP path, called through the C-ABI, against the oracle on the
same seeded inputs.  Bit-exact for scale, blend and block bytes."""
import os

import numpy as np
import pytest

import timg_amd
from timg_amd import synth

pytestmark = pytest.mark.gpu

BG, PAT = (30, 30, 46, 255), (200, 190, 180, 255)

SCALE_CASES = [
    # (kind, sw, sh, dw, dh)  -- covers V-first gather, H-first gather, scatter,
    # enlarging, point axes, odd sizes, identity
    ("alpha", 640, 480, 67, 50),
    ("noise", 640, 480, 67, 50),
    ("alpha", 64, 48, 20, 15),
    ("noise", 50, 40, 120, 90),
    ("alpha", 64, 64, 64, 64),
    ("alpha", 64, 48, 64, 20),
    ("alpha", 64, 48, 20, 48),
    ("noise", 500, 400, 13, 11),
    ("alpha", 1000, 1000, 100, 100),
    ("noise", 37, 29, 111, 87),
    ("alpha", 300, 300, 7, 3),
    ("alpha", 1920, 1080, 200, 56),
    ("noise", 17, 1000, 5, 20),
    ("photo", 1280, 720, 400, 225),
    ("alpha", 1, 1, 5, 5),
    ("alpha", 5, 5, 1, 1),
    ("alpha", 320, 200, 100, 56),  # __graft_entry__.smoke()'s geometry
]


@pytest.mark.parametrize("kind,sw,sh,dw,dh", SCALE_CASES)
@pytest.mark.parametrize("kernel", [1, 0])
def test_scale_bit_exact(hip, oracle, kind, sw, sh, dw, dh, kernel):
    src = synth.make(kind, sw, sh, seed=sw * 7 + dh)
    got = hip.scale(src, dw, dh, kernel=kernel)
    want = oracle.scale(src, dw, dh)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} differing bytes"


@pytest.mark.parametrize("sw,sh,dw,dh", [(1000, 300, 6000, 280), (1024, 1024, 5500, 900), (800, 600, 5000, 500),
                                         (1500, 900, 6100, 850)])
def test_wide_upscales_that_overflow_the_matrix_kernels_lds(hip, oracle, sw, sh, dw, dh):
    """One- or two-tap rows with thousands of outputs per strip: the VALU streaming layout fits 128 KB of LDS, the
    matrix kernel's float4 weight groups do not -- the launch must fall back (same bytes), not fail."""
    src = synth.make("alpha", sw, sh, seed=sw + dh)
    assert np.array_equal(hip.scale(src, dw, dh), oracle.scale(src, dw, dh))


@pytest.mark.parametrize("pieces", [1, 2, 3, 4, 0])
def test_fused_scale_sixel_call_equals_the_two_calls(hip, oracle, pieces):
    """timg_hip_scale_sixel_encode (the batch cut into pieces on the context's side streams, a piece's scale beside
    the serial sixel stages of the pieces in front of it) delivers the bytes of timg_hip_scale_blend followed by
    timg_hip_sixel_encode -- ragged piece sizes included -- and those are the oracle's."""
    n, sw, sh, dw, dh = 7, 640, 360, 133, 75
    frames = np.stack([synth.make("alpha" if i % 2 else "photo", sw, sh, seed=40 + i) for i in range(n)])
    blend = timg_amd.Blend.make(BG, PAT, 4, 4)
    dsrc = hip.upload(frames)
    dscaled = hip.malloc(n * dw * dh * 4)
    cap = hip.sixel_max_bytes(dw, dh)
    dout = hip.malloc(n * cap)
    sc = hip.scaler(sw, sh, dw, dh)
    for _ in range(2):  # (twice: the tile bookkeeping of the pieces' slots carries over between calls)
        lens, scale_ms = hip.scale_sixel_encode(sc, dsrc, dscaled, n, blend, dout, cap, pieces=pieces)
        assert scale_ms > 0
        scaled = hip.download(dscaled, n * dw * dh * 4).reshape(n, dh, dw, 4)
        out = hip.download(dout, n * cap).reshape(n, cap)
        for i in range(n):
            want, _ = oracle.alpha_compose(oracle.scale(frames[i], dw, dh), BG, PAT, 4, 4)
            assert np.array_equal(scaled[i], want), i
            assert out[i, :lens[i]].tobytes() == oracle.sixel_encode(want, BG, PAT, 4, 4, lookup_mode=1), i
    for p in (dsrc, dscaled, dout):
        hip.free(p)
    sc.close()


def test_scale_bgra_input(hip, oracle):
    src = synth.alpha(200, 150, seed=3)
    assert np.array_equal(hip.scale(src, 77, 41, in_fmt=1), oracle.scale(src, 77, 41, in_fmt=1))


def test_scale_triangle_filter(hip, oracle):
    src = synth.alpha(320, 240, seed=4)
    for dw, dh in [(100, 75), (500, 300)]:
        assert np.array_equal(hip.scale(src, dw, dh, filter=2), oracle.scale(src, dw, dh, filter=2))


@pytest.mark.parametrize("pattern,pw,ph", [((0, 0, 0, 0), 0, 0), (PAT, 1, 1), (PAT, 9, 10), (PAT, 10, 9)])
@pytest.mark.parametrize("start_row", [0, 7])
def test_fused_blend_bit_exact(hip, oracle, pattern, pw, ph, start_row):
    src = synth.alpha(400, 300, seed=11)
    blend = timg_amd.Blend.make(BG, pattern, pw, ph, start_row)
    got = hip.scale(src, 133, 100, blend=blend)
    want, calls = oracle.alpha_compose(oracle.scale(src, 133, 100), BG, pattern, pw, ph, start_row)
    assert calls == 1
    assert np.array_equal(got, want)


def test_blend_lazy_flag_and_disabled(hip, oracle):
    opaque = synth.photo(200, 100, seed=1)
    sc = hip.scaler(200, 100, 50, 25)
    dst = np.empty((25, 50, 4), np.uint8)
    flags = hip.scale_blend(sc, opaque, dst, 1, timg_amd.Blend.make(BG), want_transparent=True)
    assert flags == [0]  # the reference would never have called the bg getter
    transparent = synth.alpha(200, 100, seed=1)
    flags = hip.scale_blend(sc, transparent, dst, 1, timg_amd.Blend.make((0, 0, 0, 0)),
                            want_transparent=True)
    assert flags == [1]
    assert np.array_equal(dst, oracle.scale(transparent, 50, 25))  # bg alpha 0: untouched
    sc.close()


def test_standalone_alpha_compose(hip, oracle):
    fb = synth.alpha(123, 77, seed=5)
    for start_row in (0, 70, 77):
        for pattern, pw, ph in [((0, 0, 0, 0), 0, 0), (PAT, 5, 3)]:
            mine = fb.copy()
            hip.alpha_compose(mine, 123, 77, timg_amd.Blend.make(BG, pattern, pw, ph, start_row))
            want, _ = oracle.alpha_compose(fb, BG, pattern, pw, ph, start_row)
            assert np.array_equal(mine, want)


def test_batched_device_resident_frames(hip, oracle):
    n, sw, sh, dw, dh = 5, 320, 180, 100, 56
    frames = np.stack([synth.alpha(sw, sh, seed=i) for i in range(n)])
    dsrc = hip.upload(frames)
    ddst = hip.malloc(n * dw * dh * 4)
    sc = hip.scaler(sw, sh, dw, dh)
    blend = timg_amd.Blend.make(BG, PAT, 4, 4)
    hip.scale_blend(sc, dsrc, ddst, n, blend)
    hip.sync()
    out = hip.download(ddst, n * dw * dh * 4).reshape(n, dh, dw, 4)
    for i in range(n):
        want, _ = oracle.alpha_compose(oracle.scale(frames[i], dw, dh), BG, PAT, 4, 4)
        assert np.array_equal(out[i], want), i
    hip.free(dsrc)
    hip.free(ddst)
    sc.close()


BLOCK_SIZES = [(67, 50), (100, 28 * 2), (33, 21), (1, 1), (2, 3), (200, 56), (65, 7), (129, 64)]


@pytest.mark.parametrize("w,h", BLOCK_SIZES)
@pytest.mark.parametrize("flags", [0, 1, 2, 4, 5, 6])
def test_block_bytes_exact(hip, oracle, w, h, flags):
    for kind, seed in [("noise", 1), ("alpha", 2), ("photo", 3)]:
        fb = synth.make(kind, w, h, seed + w)
        if kind == "photo":  # long runs of identical colours exercise the elision
            fb[..., :3] = (fb[..., :3] // 64) * 64
        got = hip.block_encode(fb, w, h, flags=flags, x_indent=0)[0]
        want = oracle.block_encode(fb, quarter=bool(flags & 1), upper=bool(flags & 2),
                                   color256=bool(flags & 4))
        assert got == want, (kind, len(got), len(want))


def test_block_indent_and_batch(hip, oracle):
    w, h, n = 100, 56, 6
    frames = np.stack([synth.alpha(w, h, seed=20 + i) for i in range(n)])
    for flags, x in [(1, 200), (0, 37), (1, 1)]:
        outs = hip.block_encode(frames, w, h, flags=flags, x_indent=x, n_frames=n)
        for i in range(n):
            assert outs[i] == oracle.block_encode(frames[i], quarter=bool(flags & 1), x=x), i


def _cursor_up_prefix(dy, rows_of):
    # TerminalCanvas::MoveCursorDY as consumed by Send (src/terminal-canvas.cc:66-73)
    return b"" if dy >= 0 else b"\x1b[%dA" % rows_of(-dy)


@pytest.mark.parametrize("flags", [0, 1, 2, 4, 5])
def test_block_frame_diff_sequences(hip, oracle, flags):
    """Animation frames through one stateful canvas: the frame-difference
    encoding (skipped cells -> cursor moves) byte for byte against the oracle's
    canvas, which test_oracle_vs_ref pins to the real UnicodeBlockCanvas."""
    rng = np.random.default_rng(100 + flags)
    for trial in range(12):
        w, h, x = int(rng.integers(2, 90)), int(rng.integers(2, 60)), int(rng.integers(0, 7))
        if trial == 0:
            w, h, x = 200, 56, 0
        q, up, c256 = bool(flags & 1), bool(flags & 2), bool(flags & 4)
        oc = oracle.block_canvas(q, up, c256)
        hc = hip.block_canvas(flags)
        fb = synth.photo(w, h, seed=trial)
        for f in range(7):
            dy = 0 if f == 0 else -h
            if f == 4:
                dy = -h - 2  # breaks the emit_difference condition (:344-346)
            k = int(rng.integers(0, 6))
            fb = fb.copy()
            if k == 1:
                fb[rng.integers(0, h), rng.integers(0, w)] = [1, 2, 3, 255]
            elif k == 2:
                fb[rng.integers(0, h):] = rng.integers(0, 256, 3).tolist() + [255]
            elif k == 3:
                fb = synth.noise(w, h, seed=trial * 10 + f, opaque=True)
            elif k == 4:  # a few scattered rows change, the rest is skipped
                for r in rng.integers(0, h, 3):
                    fb[r, rng.integers(0, w):] = rng.integers(0, 256, 4).tolist()
            # k == 0 / 5: identical frame -> the reference emits an empty buffer
            want = oc.send(x, dy, fb)
            body = hc.send(x, dy, fb, w, h)
            # the reference drops the cursor prefix too when nothing was emitted (:390-395)
            got = _cursor_up_prefix(dy, lambda px: (px + 1) // 2) + body if body else b""
            assert got == want, (trial, f, k, len(got), len(want), got[:60], want[:60])
        oc.close()
        hc.close()


def test_block_output_too_small_is_reported(hip):
    fb = synth.noise(64, 64, 1)
    with pytest.raises(timg_amd.TimgHipError) as e:
        hip.block_encode(fb, 64, 64, flags=1, out_cap=100)
    assert e.value.code == -4


def test_autocrop_bbox(hip, oracle):
    fb = np.zeros((90, 160, 4), np.uint8)
    fb[...] = (10, 20, 30, 255)
    fb[17:60, 33:120] = (200, 0, 0, 255)
    assert hip.autocrop_bbox(fb, 160, 90).tolist() == [[33, 17, 87, 43]]
    assert hip.autocrop_bbox(fb, 160, 90, crop_border=40).tolist() == [[0, 0, 0, 0]]
    blank = np.zeros((20, 30, 4), np.uint8)
    assert hip.autocrop_bbox(blank, 30, 20).tolist() == [[0, 0, 0, 0]]
    # random frames against the restatement of trim(): differently coloured borders per side,
    # content touching an edge, a crop_border that cuts into the content, batches
    rng = np.random.default_rng(11)
    for _ in range(60):
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
        fb = np.empty((h, w, 4), np.uint8)
        fb[...] = rng.integers(0, 256, 4)
        if rng.random() < 0.5:   # right / bottom margins of another colour than the top-left one
            fb[:, w - int(rng.integers(0, w // 2 + 1)):] = rng.integers(0, 256, 4)
            fb[h - int(rng.integers(0, h // 2 + 1)):, :] = rng.integers(0, 256, 4)
        if rng.random() < 0.85:
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            x1, y1 = int(rng.integers(x0, w)) + 1, int(rng.integers(y0, h)) + 1
            fb[y0:y1, x0:x1] = rng.integers(0, 256, (y1 - y0, x1 - x0, 4))
        border = int(rng.integers(0, 6)) if rng.random() < 0.5 else 0
        assert hip.autocrop_bbox(fb, w, h, crop_border=border).tolist()[0] == oracle.autocrop_bbox(fb, border), (w, h, border)
    batch = np.stack([synth.make("alpha", 64, 40, seed=s) for s in range(5)])
    got = hip.autocrop_bbox(batch, 64, 40, n_frames=5)
    assert got.tolist() == [oracle.autocrop_bbox(batch[i]) for i in range(5)]


def test_autocrop_then_scale_reads_the_window_in_place(hip, oracle):
    """--crop-border + --auto-crop as the reference applies them (before scaling,
    src/graphics-magick-source.cc:231-254): the box comes from the reduction, the scaler is created for
    the cropped size and reads the window through pointer offset + stride -- no cropped copy."""
    w, h = 640, 360
    frame = np.empty((h, w, 4), np.uint8)
    frame[...] = (12, 12, 12, 255)
    frame[40:300, 100:580] = synth.make("photo", 480, 260, seed=4)
    x, y, cw, ch = hip.autocrop_bbox(frame, w, h, crop_border=8).tolist()[0]
    assert (x, y, cw, ch) == tuple(oracle.autocrop_bbox(frame, 8)) and cw < w - 16 and ch < h - 16
    dw, dh = 120, 65
    sc = hip.scaler(cw, ch, dw, dh)
    dev = hip.upload(frame)
    out = hip.malloc(dw * dh * 4)
    hip.scale_blend(sc, dev + y * w * 4 + x * 4, out, 1, src_stride=w * 4)
    hip.sync()
    got = hip.download(out, dw * dh * 4).reshape(dh, dw, 4)
    hip.free(dev)
    hip.free(out)
    sc.close()
    assert np.array_equal(got, oracle.scale(np.ascontiguousarray(frame[y:y + ch, x:x + cw]), dw, dh))


# ---- sixel: the HIP path implements oracle lookup_mode 1 byte for byte -------
SIXEL_CASES = [("photo", 800, 450), ("alpha", 320, 203), ("noise", 200, 100), ("photo", 64, 7),
               ("photo", 100, 56), ("noise", 33, 6), ("photo", 2, 13), ("alpha", 1, 1)]


@pytest.mark.parametrize("kind,w,h", SIXEL_CASES)
def test_sixel_bytes_match_oracle(hip, oracle, kind, w, h):
    fb = synth.make(kind, w, h, seed=5)
    got = hip.sixel_encode(fb, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 4, 4),
                           out_cap=hip.sixel_max_bytes(w, h) * 4)[0]
    want = oracle.sixel_encode(fb, BG, PAT, 4, 4, lookup_mode=1)
    if got != want:
        n = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
        raise AssertionError(f"len {len(got)} vs {len(want)}, first diff at {n}: "
                             f"{got[max(0, n - 20):n + 20]!r} vs {want[max(0, n - 20):n + 20]!r}")


def test_sixel_few_colours_and_no_background(hip, oracle):
    fb = np.zeros((36, 120, 4), np.uint8)
    fb[..., 3] = 255
    for i in range(20):
        fb[:, 6 * i:6 * i + 6, :3] = (8 * i % 256, 16 * (i % 16), 248 - 8 * (i % 32))
    assert hip.sixel_encode(fb, 120, 36)[0] == oracle.sixel_encode(fb, has_getter=False)
    fb2 = synth.photo(60, 20, 2)
    got = hip.sixel_encode(fb2, 60, 20, flags=1, pad_blend=timg_amd.Blend.make(BG, (200, 10, 10, 255), 9, 9))[0]
    assert got == oracle.sixel_encode(fb2, BG, (200, 10, 10, 255), 9, 9, broken_cursor=True)


def test_sixel_batch_device_resident(hip, oracle):
    n, w, h = 4, 200, 112
    frames = np.stack([synth.photo(w, h, 40 + i) for i in range(n)])
    d = hip.upload(frames)
    outs = hip.sixel_encode(d, w, h, pad_blend=timg_amd.Blend.make(BG), n_frames=n)
    # device input, host output
    for i in range(n):
        assert outs[i] == oracle.sixel_encode(frames[i], BG), i
    hip.free(d)


def test_sixel_async_encode_is_the_blocking_call_read_late(hip, oracle):
    """timg_hip_sixel_encode_async / _wait (round 5): three batches of different sizes are ENQUEUED on one stream, on two
    alternating jobs, before any byte count is looked at -- the scratch of call k is reused by call k + 1 in stream
    order, the counts travel in the job's own pinned words -- and every frame is then byte for byte what the blocking
    call and the restatement produce.  The reference contract kept: SixelCanvas::Send hands over a future and returns
    (src/sixel-canvas.cc:128-154)."""
    import torch
    w, h = 200, 112
    cap = hip.sixel_max_bytes(w, h)
    batches = [np.stack([synth.make(kind, w, h, 60 + 7 * b + i) for i in range(n)])
               for b, (kind, n) in enumerate([("photo", 5), ("noise", 2), ("alpha", 7)])]
    blend = timg_amd.Blend.make(BG, PAT, 5, 3)
    devs = [hip.upload(b) for b in batches]
    outs = [torch.empty((len(b), cap), dtype=torch.uint8, device="cuda") for b in batches]
    st = torch.cuda.Stream()
    jobs = [hip.sixel_job(8), hip.sixel_job(8)]
    # job 0 <- batch 0, job 1 <- batch 1: both in flight; batch 2 needs job 0 back first
    hip.sixel_encode_async(jobs[0], devs[0], w, h, outs[0].data_ptr(), cap, n_frames=5, pad_blend=blend, stream=st.cuda_stream)
    hip.sixel_encode_async(jobs[1], devs[1], w, h, outs[1].data_ptr(), cap, n_frames=2, pad_blend=blend, stream=st.cuda_stream)
    with pytest.raises(timg_amd.TimgHipError):  # a job holds ONE call
        hip.sixel_encode_async(jobs[0], devs[2], w, h, outs[2].data_ptr(), cap, n_frames=7, pad_blend=blend, stream=st.cuda_stream)
    lens0 = hip.sixel_encode_wait(jobs[0], 5)
    hip.sixel_encode_async(jobs[0], devs[2], w, h, outs[2].data_ptr(), cap, n_frames=7, pad_blend=blend, stream=st.cuda_stream)
    lens1 = hip.sixel_encode_wait(jobs[1], 2)
    lens2 = hip.sixel_encode_wait(jobs[0], 7)
    with pytest.raises(timg_amd.TimgHipError):  # nothing in flight any more
        hip.sixel_encode_wait(jobs[0], 7)
    with pytest.raises(timg_amd.TimgHipError):  # more frames than the job was created for
        hip.sixel_encode_async(jobs[1], devs[2], w, h, outs[2].data_ptr(), cap, n_frames=9, pad_blend=blend, stream=st.cuda_stream)
    st.synchronize()
    for b, (frames, lens, out) in enumerate(zip(batches, (lens0, lens1, lens2), outs)):
        host = out.cpu().numpy()
        blocking = hip.sixel_encode(devs[b], w, h, pad_blend=blend, n_frames=len(frames))
        for i in range(len(frames)):
            got = host[i, :lens[i]].tobytes()
            assert got == blocking[i], (b, i)
            assert got == oracle.sixel_encode(frames[i], BG, PAT, 5, 3, lookup_mode=1), (b, i)
    for j in jobs:
        hip.sixel_job_destroy(j)
    for d in devs:
        hip.free(d)


def test_partitioned_streams_write_the_same_bytes(hip, oracle):
    """timg_hip_stream_create (round 6): the scale call on a stream whose kernels stay off 12 CUs of every XCD, the sixel
    chain on an unrestricted stream of the greatest priority, step k + 1's scale beside step k's chain
    (PartitionedSixelPipeline).  Five steps over frames with alpha: every frame of the last step is the restatement's of
    the reference scaler's output, and what the two plain calls on one stream write.  Then the API's edges: a reserve
    that is not a multiple of four is rounded down, one that leaves no CU is refused, a stream of another owner is not
    destroyed."""
    import torch
    from timg_amd.pipeline import PartitionedSixelPipeline
    n, sw, sh, dw, dh = 6, 640, 360, 200, 112
    frames = np.stack([synth.make("alpha" if i % 2 else "photo", sw, sh, 700 + i) for i in range(n)])
    blend = timg_amd.Blend.make(BG, PAT, 5, 3)
    src = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    p = PartitionedSixelPipeline(hip, n, sw, sh, dw, dh, blend, reserved_cus_per_xcd=12)
    lens = p.run(src, 5)
    got = [p.frame_bytes(i) for i in range(n)]
    assert len(lens) == n
    # the two plain calls, blocking, on the context's own stream
    sc = hip.scaler(sw, sh, dw, dh)
    scaled = torch.empty((n, dh, dw, 4), dtype=torch.uint8, device="cuda")
    hip.scale_blend(sc, src.data_ptr(), scaled.data_ptr(), n, blend)
    hip.sync()
    plain = hip.sixel_encode(scaled.data_ptr(), dw, dh, pad_blend=blend, n_frames=n)
    scaled_host = scaled.cpu().numpy()
    for i in range(n):
        assert got[i] == plain[i], i
        want_px = oracle.alpha_compose(oracle.scale(frames[i], dw, dh), BG, PAT, 5, 3, 0)[0]
        assert np.array_equal(scaled_host[i], want_px), i
        assert got[i] == oracle.sixel_encode(want_px, BG, PAT, 5, 3, lookup_mode=1), i
    sc.close()
    p.close()
    a = hip.stream_create(reserved_cus_per_xcd=7)   # -> 4 CUs of every XCD
    b = hip.stream_create(reserved_cus_per_xcd=3)   # -> none: a plain stream
    hip.stream_wait_stream(a, b)
    hip.stream_wait_stream(0, a)                     # (0 = the context's own stream)
    hip.sync(a)
    hip.stream_destroy(a)
    hip.stream_destroy(b)
    with pytest.raises(timg_amd.TimgHipError) as e:
        hip.stream_create(reserved_cus_per_xcd=32)
    assert e.value.code == -5
    with pytest.raises(timg_amd.TimgHipError):
        hip.stream_destroy(torch.cuda.Stream().cuda_stream)


def test_sixel_calls_on_two_streams_of_one_context_are_ordered_by_the_library(hip, oracle):
    """a call of timg_hip_sixel_encode_async leaves kernels running on the context's scratch after it has
    returned; nothing but stream order kept the NEXT sixel call off that scratch -- so a call on another stream, a
    blocking call on the context's own stream, or one that has to GROW the scratch (which frees the old block) raced
    with it.  The library now orders them itself (context.h: sixel_done).  Here: a large batch in flight on stream A,
    then at once a small batch on stream B, a blocking call on the default stream, and a batch big enough to grow the
    scratch on stream B -- every frame must be the restatement's, every time."""
    import torch
    w, h = 320, 180
    cap = hip.sixel_max_bytes(w, h)
    blend = timg_amd.Blend.make(BG, PAT, 5, 3)
    big = np.stack([synth.make("photo", w, h, 300 + i) for i in range(24)])
    small = np.stack([synth.make("noise", w, h, 400 + i) for i in range(3)])
    bigger = np.stack([synth.make("alpha", w, h, 500 + i) for i in range(40)])
    want = {name: [oracle.sixel_encode(f, BG, PAT, 5, 3, lookup_mode=1) for f in frames]
            for name, frames in (("big", big), ("small", small), ("bigger", bigger))}
    d_big, d_small, d_bigger = hip.upload(big), hip.upload(small), hip.upload(bigger)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(3):
        o_big = torch.empty((24, cap), dtype=torch.uint8, device="cuda")
        o_small = torch.empty((3, cap), dtype=torch.uint8, device="cuda")
        o_bigger = torch.empty((40, cap), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ja, jb = hip.sixel_job(24), hip.sixel_job(40)
        hip.sixel_encode_async(ja, d_big, w, h, o_big.data_ptr(), cap, n_frames=24, pad_blend=blend, stream=sa.cuda_stream)
        hip.sixel_encode_async(jb, d_small, w, h, o_small.data_ptr(), cap, n_frames=3, pad_blend=blend, stream=sb.cuda_stream)
        blocking = hip.sixel_encode(d_small, w, h, pad_blend=blend, n_frames=3)  # (the context's own stream)
        l_small = hip.sixel_encode_wait(jb, 3)
        hip.sixel_encode_async(jb, d_bigger, w, h, o_bigger.data_ptr(), cap, n_frames=40, pad_blend=blend, stream=sb.cuda_stream)
        l_big = hip.sixel_encode_wait(ja, 24)
        l_bigger = hip.sixel_encode_wait(jb, 40)
        torch.cuda.synchronize()
        for name, out, lens in (("big", o_big, l_big), ("small", o_small, l_small), ("bigger", o_bigger, l_bigger)):
            host = out.cpu().numpy()
            for i, ref in enumerate(want[name]):
                assert host[i, :lens[i]].tobytes() == ref, (rep, name, i)
        assert list(blocking) == want["small"], rep
        hip.sixel_job_destroy(ja)
        hip.sixel_job_destroy(jb)
    for d in (d_big, d_small, d_bigger):
        hip.free(d)


@pytest.mark.parametrize("kind,w,h", [
    ("noise", 800, 450),   # > 8192 distinct colours: median cut runs on the global-memory table
    ("photo", 64, 1100),   # 1104 padded rows: the diffusion pipeline goes round three times (16 waves x 32 rows)
    ("photo", 1200, 600),  # wide boundary rows leave LDS for 7 diffusion waves only: three rounds of 224 rows
    ("photo", 766, 450),   # 13 diffusion waves fill the 160 KB of LDS to the byte: the static part must still fit
    ("alpha", 801, 77),    # odd width: padded index rows, pad rows (77 -> 78) with a checkerboard
    ("photo", 1365, 30),   # widest frame the band encoder takes
    ("noise", 3, 130),     # narrower than the row skew
])
def test_sixel_geometry_corner_cases(hip, oracle, kind, w, h):
    fb = synth.make(kind, w, h, seed=17)
    got = hip.sixel_encode(fb, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 5, 3),
                           out_cap=hip.sixel_max_bytes(w, h) * 4)[0]
    want = oracle.sixel_encode(fb, BG, PAT, 5, 3, lookup_mode=1)
    assert len(got) == len(want) and got == want, (len(got), len(want))


@pytest.mark.parametrize("trips", [1, 2])
@pytest.mark.parametrize("kind,w,h,parts", [
    ("photo", 800, 450, -1),   # the bench frame: four parts of four row groups
    ("noise", 800, 450, 1),    # ... in one workgroup (five waves beside the colour tables, twelve beside the small ones)
    ("alpha", 320, 203, -1),   # seven row groups: one workgroup
    ("photo", 100, 56, -1),    # narrower than a wave's skew + the index delay (no column is ever "steady")
    ("noise", 33, 6, -1),
    ("photo", 3, 130, -1),
    ("alpha", 801, 77, -1),    # odd width: the last group of four indices is completed by junk in the row's padding
    ("photo", 802, 64, -1), ("photo", 803, 64, -1),
    ("photo", 1920, 1080, -1), # a full-HD frame alone: twelve parts of three row groups (small tables)
    ("noise", 1000, 500, -1),
    ("photo", 4095, 40, -1),   # the widest frame: one wave, ONE boundary row beside the colour tables (it follows itself)
    ("photo", 2600, 100, -1),  # one wave going round four times on that one row
    ("photo", 64, 1100, 3),    # parts of 12 / 12 / 11 row groups
])
def test_sixel_both_lookup_forms(hip, oracle, monkeypatch, kind, w, h, parts, trips):
    """DitherKernel<., ., kOneTrip>: the palette colour of a cell in ONE LDS round trip on the serial chain (96 KB of
    colour tables, the palette index from memory kDitherAhead steps later) or in two (cell -> index -> colour, 34 KB).
    sixel_launch.h picks by geometry; TIMG_HIP_DITHER_TRIPS forces either -- both give the oracle's bytes for every
    placement (one workgroup, parts, a single wave that follows itself)."""
    monkeypatch.setenv("TIMG_HIP_DITHER_TRIPS", str(trips))
    if parts >= 0:
        monkeypatch.setenv("TIMG_HIP_DITHER_PARTS", str(parts))
    fb = synth.make(kind, w, h, seed=23)
    got = hip.sixel_encode(fb, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 5, 3),
                           out_cap=hip.sixel_max_bytes(w, h) * 4)[0]
    want = oracle.sixel_encode(fb, BG, PAT, 5, 3, lookup_mode=1)
    if got != want:
        n = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
        raise AssertionError(f"len {len(got)} vs {len(want)}, first diff at {n}: "
                             f"{got[max(0, n - 20):n + 20]!r} vs {want[max(0, n - 20):n + 20]!r}")


@pytest.mark.parametrize("kind,w,h,parts", [
    ("photo", 800, 450, -1), ("noise", 800, 450, 1), ("alpha", 320, 203, -1), ("photo", 100, 56, -1), ("noise", 34, 6, -1),
    ("photo", 4, 130, -1), ("photo", 802, 64, -1), ("photo", 64, 1100, 3), ("noise", 1000, 500, -1),
])
def test_sixel_pixel_pairs_equal_single_pixels(hip, oracle, monkeypatch, kind, w, h, parts):
    """DitherKernel<., ., true, kPix2>: frames of even width request their pixels two at a time (one 8-byte load
    every second step); TIMG_HIP_DITHER_PIX=1 keeps the one-pixel requests.  Same bytes, the oracle's, in both -- and a
    frame whose rows are only 4-byte aligned (a view into a wider image) takes the one-pixel form by itself."""
    if parts >= 0:
        monkeypatch.setenv("TIMG_HIP_DITHER_PARTS", str(parts))
    fb = synth.make(kind, w, h, seed=29)
    want = oracle.sixel_encode(fb, BG, PAT, 5, 3, lookup_mode=1)
    for pix in ("2", "1"):
        monkeypatch.setenv("TIMG_HIP_DITHER_PIX", pix)
        got = hip.sixel_encode(fb, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 5, 3),
                               out_cap=hip.sixel_max_bytes(w, h) * 4)[0]
        assert got == want, f"pixel requests of {pix}: {len(got)} vs {len(want)} bytes"


@pytest.mark.parametrize("kind,w,h,n", [("photo", 800, 450, 3), ("noise", 101, 37, 5), ("alpha", 2, 9, 2)])
def test_sixel_byte_counts_written_by_the_kernels_or_copied(hip, oracle, monkeypatch, kind, w, h, n):
    """Round 6: the chain's first kernel clears the error word and its last one writes the frames' byte counts and the
    error word into the caller's pinned words; TIMG_HIP_SIXEL_COPY_LENGTHS=1 keeps the memset + copy form.  The same bytes
    and counts, the oracle's, either way -- in a batch, on frames of both lookup forms' sizes (w = 2: the narrow kernel)."""
    frames = np.stack([np.asarray(synth.make(kind, w, h, seed=41 + i)).reshape(h, w, 4) for i in range(n)])
    want = [oracle.sixel_encode(frames[i], BG, PAT, 4, 2, lookup_mode=1) for i in range(n)]
    d = hip.upload(frames)
    try:
        for form in ("", "1"):
            if form:
                monkeypatch.setenv("TIMG_HIP_SIXEL_COPY_LENGTHS", form)
            else:
                monkeypatch.delenv("TIMG_HIP_SIXEL_COPY_LENGTHS", raising=False)
            outs = hip.sixel_encode(d, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 4, 2), n_frames=n,
                                    out_cap=hip.sixel_max_bytes(w, h) * 2)
            for i in range(n):
                assert outs[i] == want[i], f"form {form!r} frame {i}: {len(outs[i])} vs {len(want[i])} bytes"
    finally:
        hip.free(d)


@pytest.mark.parametrize("trips", [1, 2])
@pytest.mark.parametrize("kind,w,h,parts", [
    ("photo", 800, 450, -1),   # the bench frame: four parts of four row groups
    ("noise", 800, 450, 1),    # ... in one workgroup (five waves beside the colour tables, twelve beside the small ones)
    ("alpha", 320, 203, -1),   # seven row groups: one workgroup
    ("photo", 100, 56, -1),    # narrower than a wave's skew + the index delay (no column is ever "steady")
    ("noise", 33, 6, -1),
    ("photo", 3, 130, -1),
    ("alpha", 801, 77, -1),    # odd width: the last group of four indices is completed by junk in the row's padding
    ("photo", 802, 64, -1), ("photo", 803, 64, -1),
    ("photo", 1920, 1080, -1), # a full-HD frame alone: twelve parts of three row groups (small tables)
    ("noise", 1000, 500, -1),
    ("photo", 4095, 40, -1),   # the widest frame: one wave, ONE boundary row beside the colour tables (it follows itself)
    ("photo", 2600, 100, -1),  # one wave going round four times on that one row
    ("photo", 64, 1100, 3),    # parts of 12 / 12 / 11 row groups
])
def test_sixel_both_lookup_forms(hip, oracle, monkeypatch, kind, w, h, parts, trips):
    """DitherKernel<., ., kOneTrip>: the palette colour of a cell in ONE LDS round trip on the serial chain (96 KB of
    colour tables, the palette index from memory kDitherAhead steps later) or in two (cell -> index -> colour, 34 KB).
    sixel_launch.h picks by geometry; TIMG_HIP_DITHER_TRIPS forces either -- both give the oracle's bytes for every
    placement (one workgroup, parts, a single wave that follows itself)."""
    monkeypatch.setenv("TIMG_HIP_DITHER_TRIPS", str(trips))
    if parts >= 0:
        monkeypatch.setenv("TIMG_HIP_DITHER_PARTS", str(parts))
    fb = synth.make(kind, w, h, seed=23)
    got = hip.sixel_encode(fb, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 5, 3),
                           out_cap=hip.sixel_max_bytes(w, h) * 4)[0]
    want = oracle.sixel_encode(fb, BG, PAT, 5, 3, lookup_mode=1)
    if got != want:
        n = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
        raise AssertionError(f"len {len(got)} vs {len(want)}, first diff at {n}: "
                             f"{got[max(0, n - 20):n + 20]!r} vs {want[max(0, n - 20):n + 20]!r}")


@pytest.mark.parametrize("kind,w,h,parts", [
    ("photo", 800, 450, -1), ("noise", 800, 450, 1), ("alpha", 320, 203, -1), ("photo", 100, 56, -1), ("noise", 34, 6, -1),
    ("photo", 4, 130, -1), ("photo", 802, 64, -1), ("photo", 64, 1100, 3), ("noise", 1000, 500, -1),
])
def test_sixel_pixel_pairs_equal_single_pixels(hip, oracle, monkeypatch, kind, w, h, parts):
    """DitherKernel<., ., true, kPix2>: frames of even width request their pixels two at a time (one 8-byte load
    every second step); TIMG_HIP_DITHER_PIX=1 keeps the one-pixel requests.  Same bytes, the oracle's, in both -- and a
    frame whose rows are only 4-byte aligned (a view into a wider image) takes the one-pixel form by itself."""
    if parts >= 0:
        monkeypatch.setenv("TIMG_HIP_DITHER_PARTS", str(parts))
    fb = synth.make(kind, w, h, seed=29)
    want = oracle.sixel_encode(fb, BG, PAT, 5, 3, lookup_mode=1)
    for pix in ("2", "1"):
        monkeypatch.setenv("TIMG_HIP_DITHER_PIX", pix)
        got = hip.sixel_encode(fb, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 5, 3),
                               out_cap=hip.sixel_max_bytes(w, h) * 4)[0]
        assert got == want, f"pixel requests of {pix}: {len(got)} vs {len(want)} bytes"


@pytest.mark.parametrize("kind,w,h,n", [("photo", 800, 450, 3), ("noise", 101, 37, 5), ("alpha", 2, 9, 2)])
def test_sixel_byte_counts_written_by_the_kernels_or_copied(hip, oracle, monkeypatch, kind, w, h, n):
    """Round 6: the chain's first kernel clears the error word and its last one writes the frames' byte counts and the
    error word into the caller's pinned words; TIMG_HIP_SIXEL_COPY_LENGTHS=1 keeps the memset + copy form.  The same bytes
    and counts, the oracle's, either way -- in a batch, on frames of both lookup forms' sizes (w = 2: the narrow kernel)."""
    import torch
    fbs = [synth.make(kind, w, h, seed=41 + i) for i in range(n)]
    want = [oracle.sixel_encode(fb, BG, PAT, 4, 2, lookup_mode=1) for fb in fbs]
    dev = torch.from_numpy(np.stack([np.frombuffer(fb, dtype=np.uint8).reshape(h, w, 4) for fb in fbs])).cuda()
    cap = hip.sixel_max_bytes(w, h) * 2
    for form in ("", "1"):
        if form:
            monkeypatch.setenv("TIMG_HIP_SIXEL_COPY_LENGTHS", form)
        else:
            monkeypatch.delenv("TIMG_HIP_SIXEL_COPY_LENGTHS", raising=False)
        out = torch.zeros((n, cap), dtype=torch.uint8, device="cuda")
        lens = hip.sixel_encode(dev.data_ptr(), w, h, n_frames=n, pad_blend=timg_amd.Blend.make(BG, PAT, 4, 2),
                                out=out.data_ptr(), out_cap=cap)
        torch.cuda.synchronize()
        for i in range(n):
            assert lens[i] == len(want[i]), f"form {form!r} frame {i}: {lens[i]} vs {len(want[i])} bytes"
            assert out[i, :lens[i]].cpu().numpy().tobytes() == want[i], f"form {form!r} frame {i}"


@pytest.mark.parametrize("trips", [1, 2])
@pytest.mark.parametrize("pix", [1, 2])
@pytest.mark.parametrize("kind,w,h,parts,view", [
    ("photo", 800, 450, 2, False),    # two parts a frame
    ("photo", 800, 450, 16, False),   # the most parts the placement takes (one row group each where it can)
    ("alpha", 801, 77, -1, False),    # odd width: never pixel pairs, whatever TIMG_HIP_DITHER_PIX says
    ("noise", 402, 130, -1, True),    # rows 4-byte but not 8-byte aligned (a view into a wider image): one-pixel form
    ("photo", 2600, 100, -1, False),  # one wave going round four times on ONE boundary row
    ("photo", 64, 1100, 3, False),    # three rounds of a sixteen-wave workgroup per part
])
def test_sixel_every_diffusion_instantiation(hip, oracle, monkeypatch, kind, w, h, parts, view, pix, trips):
    """the diffusion's hand-counted s_waitcnt rings (two-trip / one-trip lookup x pixel pairs / single
    pixels) are chosen by geometry and alignment, so a parity run may only ever exercise one of them; a wrong count
    corrupts pixels silently.  The cross product, forced through TIMG_HIP_DITHER_TRIPS x TIMG_HIP_DITHER_PIX, on the
    geometries that pick different placements (parts 2 and 16, odd widths, misaligned rows, a single wave that follows
    itself, several rounds): the oracle's bytes every time."""
    monkeypatch.setenv("TIMG_HIP_DITHER_TRIPS", str(trips))
    monkeypatch.setenv("TIMG_HIP_DITHER_PIX", str(pix))
    if parts >= 0:
        monkeypatch.setenv("TIMG_HIP_DITHER_PARTS", str(parts))
    if view:
        wide = synth.make(kind, w + 7, h, seed=31)
        fb = np.ascontiguousarray(wide[:, 3:3 + w])
        d = hip.upload(wide)
        got = hip.sixel_encode(d + 3 * 4, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 5, 3), stride=(w + 7) * 4,
                               frame_stride=(w + 7) * 4 * h)[0]
        hip.free(d)
    else:
        fb = synth.make(kind, w, h, seed=31)
        got = hip.sixel_encode(fb, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 5, 3),
                               out_cap=hip.sixel_max_bytes(w, h) * 4)[0]
    want = oracle.sixel_encode(fb, BG, PAT, 5, 3, lookup_mode=1)
    assert got == want, f"trips {trips} pix {pix}: {len(got)} vs {len(want)} bytes"


@pytest.mark.parametrize("parts", [1, 2, 3, 4, 8])
def test_sixel_diffusion_spread_over_several_cus(hip, oracle, monkeypatch, parts):
    """Frames of eight row groups and more are diffused by several workgroups (CUs) per frame, the boundary row
    between two of them handed over through memory by helper waves (DitherKernel<., true>; default: the fewest
    parts that fit, from two).  Every part count gives the oracle's bytes -- on a batch large enough that parts of
    many frames are in flight at once (80 frames x up to 4 workgroups), device-resident, and on single frames whose
    row groups do not divide evenly."""
    monkeypatch.setenv("TIMG_HIP_DITHER_PARTS", str(parts))
    n, w, h = 80, 300, 282  # 9 row groups (47 bands of 6 rows)
    frames = np.stack([synth.make(("photo", "noise", "alpha")[i % 3], w, h, seed=300 + i) for i in range(n)])
    d = hip.upload(frames)
    outs = None
    for rep in range(3):  # (a stale word of a hand-over would be intermittent: every byte of every frame, three times)
        again = hip.sixel_encode(d, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 4, 4), n_frames=n,
                                 out_cap=hip.sixel_max_bytes(w, h) * 4)
        assert outs is None or again == outs, (parts, rep)
        outs = again
    monkeypatch.setenv("TIMG_HIP_DITHER_PARTS", "1")
    one_cu = hip.sixel_encode(d, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 4, 4), n_frames=n,
                              out_cap=hip.sixel_max_bytes(w, h) * 4)
    monkeypatch.setenv("TIMG_HIP_DITHER_PARTS", str(parts))
    hip.free(d)
    assert outs == one_cu, [i for i in range(n) if outs[i] != one_cu[i]]
    for i in (0, 1, 2, 39, 40, 77, 78, 79):
        assert outs[i] == oracle.sixel_encode(frames[i], BG, PAT, 4, 4, lookup_mode=1), (parts, i)
    for kind, w1, h1 in (("photo", 800, 450), ("alpha", 257, 353), ("noise", 640, 480)):
        fb = synth.make(kind, w1, h1, seed=parts)
        got = hip.sixel_encode(fb, w1, h1, pad_blend=timg_amd.Blend.make(BG), out_cap=hip.sixel_max_bytes(w1, h1) * 4)[0]
        assert got == oracle.sixel_encode(fb, BG, lookup_mode=1), (parts, kind, w1, h1)


@pytest.mark.parametrize("n", [200, 256, 300])
def test_sixel_batches_as_long_as_the_chip_has_cus_and_longer(hip, oracle, n):
    """bench.py cuts BASELINE configs 4 and 5 into the fewest launches of at most 256 frames (round 6: the chain is
    latency-bound, 3 x 200 frames cost 11.2 ms where ten launches of 64 cost 13.8).  With more than 64 frames a call a
    frame's diffusion no longer gets four CUs but one (PlanDither: parts capped by CUs / frames, the two-trip lookup
    beside up to sixteen boundary rows), with more than 256 the workgroups of a launch no longer fit the chip at once:
    every frame of such a batch is the frame the oracle encodes, and the frame a 64-frame call produced."""
    w, h, distinct = 800, 450, 7
    base = np.stack([synth.make(("photo", "alpha")[i % 2], w, h, seed=900 + i) for i in range(distinct)])
    frames = base[np.arange(n) % distinct]
    d = hip.upload(np.ascontiguousarray(frames))
    cap = hip.sixel_max_bytes(w, h)
    outs = hip.sixel_encode(d, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 9, 9), n_frames=n, out_cap=cap)
    first = hip.sixel_encode(d, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 9, 9), n_frames=64, out_cap=cap)
    hip.free(d)
    want = [oracle.sixel_encode(base[i], BG, PAT, 9, 9, lookup_mode=1) for i in range(distinct)]
    assert [i for i in range(n) if outs[i] != want[i % distinct]] == []
    assert first == outs[:64]


def test_sixel_diffusion_hand_over_under_uneven_load(hip):
    """The memory hand-over between the parts of a frame, with the chip shared unevenly: three contexts on three host
    threads encode batches of different sizes at the same time (192 + 64 + 7 frames' worth of workgroups queue for 256
    CUs, parts of different batches interleave on the queues) while a fourth thread keeps a scale kernel streaming.
    Every byte of every step equals what the same context produced alone."""
    import threading
    import torch
    from timg_amd.pipeline import synth_frames_on_device
    w, h = 800, 450
    sizes = (48, 16, 7)
    ctxs = [timg_amd.TimgHip(0) for _ in sizes]
    srcs = [synth_frames_on_device(n, w, h, ("photo", "noise", "alpha")[i], seed=60 + i) for i, n in enumerate(sizes)]
    torch.cuda.synchronize()
    blend = timg_amd.Blend.make(BG, PAT, 4, 4)

    def encode(i):
        return ctxs[i].sixel_encode(srcs[i].data_ptr(), w, h, pad_blend=blend, n_frames=sizes[i],
                                    out_cap=ctxs[i].sixel_max_bytes(w, h) * 4)
    alone = [encode(i) for i in range(len(sizes))]
    big = synth_frames_on_device(16, 3840, 2160, "photo", seed=9)
    dst = torch.empty((16, 450, 800, 4), dtype=torch.uint8, device="cuda")
    sc = hip.scaler(3840, 2160, 800, 450)
    stop, bad = threading.Event(), []

    def streamer():
        st = torch.cuda.Stream()
        while not stop.is_set():
            hip.scale_blend(sc, big.data_ptr(), dst.data_ptr(), 16, blend, stream=st.cuda_stream)
            st.synchronize()

    def worker(i):
        for rep in range(6):
            if encode(i) != alone[i]:
                bad.append((i, rep))
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(len(sizes))] + [threading.Thread(target=streamer)]
    for t in ths:
        t.start()
    for t in ths[:-1]:
        t.join()
    stop.set()
    ths[-1].join()
    sc.close()
    for c in ctxs:
        c.close()
    assert not bad, bad


@pytest.mark.parametrize("kind,w,h", [("photo", 320, 203), ("alpha", 200, 100), ("noise", 97, 61), ("photo", 64, 7),
                                      ("noise", 33, 6), ("photo", 2, 13), ("alpha", 1, 1), ("photo", 800, 450)])
def test_sixel_first_hit_lookup_is_libsixels_cache(hip, oracle, kind, w, h):
    """The checker in libtimg_hip_debug.so (timg_hip_debug_sixel_encode_first_hit): the 15-bit lookup cache filled the
    way sixel_encode fills it (the entry of a cell is the palette colour nearest to the first pixel value that lands
    in it, raster order, diffused errors included) -- byte-identical to the restatement's lookup_mode 0, pad rows and
    checkerboard included.  The product library has one rule (lookup_mode 1) and rejects unknown flags."""
    fb = synth.make(kind, w, h, seed=9)
    got = hip.sixel_encode_first_hit(fb, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 4, 4),
                                     out_cap=hip.sixel_max_bytes(w, h) * 4)[0]
    want = oracle.sixel_encode(fb, BG, PAT, 4, 4, lookup_mode=0)
    assert len(got) == len(want) and got == want, (len(got), len(want))


def test_sixel_first_hit_batch_and_few_colours(hip, oracle):
    n, w, h = 3, 120, 40
    frames = np.stack([synth.photo(w, h, 70 + i) for i in range(n)])
    outs = hip.sixel_encode_first_hit(frames, w, h, n_frames=n)
    for i in range(n):
        assert outs[i] == oracle.sixel_encode(frames[i], has_getter=False, lookup_mode=0), i
    fb = np.zeros((36, 120, 4), np.uint8)  # <= 256 colours: no diffusion, the palette is the histogram
    fb[..., 3] = 255
    for i in range(20):
        fb[:, 6 * i:6 * i + 6, :3] = (8 * i % 256, 16 * (i % 16), 248 - 8 * (i % 32))
    assert hip.sixel_encode_first_hit(fb, 120, 36)[0] == oracle.sixel_encode(fb, has_getter=False, lookup_mode=0)
    with pytest.raises(timg_amd.TimgHipError):  # (what used to select the checker through the product entry point)
        hip.sixel_encode(fb, 120, 36, flags=2)


def test_sixel_within_stated_delta_e_of_the_libsixel_like_lookup(hip, oracle):
    """north_star: sixel palette selection within a stated dE.  The HIP path (exact nearest
    colour per 15-bit cell) against the restatement's libsixel-like lossy lookup cache
    (lookup_mode 0), both decoded by the independent decoder: mean CIE76 dE to the source below
    4.0 for both, and within 0.5 of each other (the tolerances of tests/test_sixel_oracle.py)."""
    from test_sixel_oracle import DE_PHOTO, mean_delta_e
    fb = synth.photo(800, 450, seed=3)
    got = hip.sixel_encode(fb, 800, 450, pad_blend=timg_amd.Blend.make(BG))[0]
    like_libsixel = oracle.sixel_encode(fb, BG, lookup_mode=0)
    de_hip = mean_delta_e(oracle.sixel_decode(got)[0][:450, :, :3], fb[..., :3])
    de_ref = mean_delta_e(oracle.sixel_decode(like_libsixel)[0][:450, :, :3], fb[..., :3])
    assert de_hip < DE_PHOTO and de_ref < DE_PHOTO and abs(de_hip - de_ref) < 0.5, (de_hip, de_ref)


def test_sixel_round_trip_decodes_to_the_palette_image(hip, oracle):
    """Size-independent property at the BASELINE frame size: the stream decodes (independent
    decoder) to a full 800x450 raster whose every pixel is a palette colour close to the
    source -- mean error well below a just-noticeable step thanks to the diffusion."""
    fb = synth.photo(800, 450, seed=21)
    data = hip.sixel_encode(fb, 800, 450, pad_blend=timg_amd.Blend.make(BG))[0]
    img, ncolors = oracle.sixel_decode(data)
    assert img.shape[:2] == (450, 800) and 2 <= ncolors <= 256
    assert (img[..., 3] == 255).all()
    err = np.abs(img[..., :3].astype(np.int32) - fb[..., :3].astype(np.int32))
    assert err.mean() < 12.0 and np.percentile(err, 99) < 80


@pytest.mark.parametrize("kind,w,h", [("photo", 1366, 40), ("noise", 1920, 27), ("alpha", 3840, 20),
                                      ("photo", 4095, 11)])
def test_sixel_wide_frames(hip, oracle, kind, w, h):
    """Frames wider than 1365 px (8192 band entries): the band kernels sort in global scratch."""
    fb = synth.make(kind, w, h, seed=23)
    got = hip.sixel_encode(fb, w, h, pad_blend=timg_amd.Blend.make(BG, PAT, 16, 4),
                           out_cap=hip.sixel_max_bytes(w, h) * 4)[0]
    want = oracle.sixel_encode(fb, BG, PAT, 16, 4, lookup_mode=1)
    assert len(got) == len(want) and got == want, (len(got), len(want))


def test_sixel_too_wide_is_refused(hip):
    fb = np.zeros((6, 4096, 4), np.uint8)
    with pytest.raises(timg_amd.TimgHipError) as e:
        hip.sixel_encode(fb, 4096, 6)
    assert e.value.code == -5


# ---- BASELINE.json full sizes, streaming kernel vs generic kernel vs oracle ----
@pytest.mark.parametrize("kind,dw,dh", [("alpha", 800, 450), ("photo", 800, 450), ("noise", 200, 56),
                                        ("photo", 200, 56), ("alpha", 1280, 720)])
def test_full_size_4k_frames(hip, oracle, kind, dw, dh):
    src = synth.make(kind, 3840, 2160, seed=9)
    sc = hip.scaler(3840, 2160, dw, dh)
    assert sc.info()["streaming_ok"] == 1 and sc.info()["vertical_first"] == 1
    want = oracle.scale(src, dw, dh)
    for kernel in (2, 3, 4, 1):
        sc.set_kernel(kernel)
        got = np.empty((dh, dw, 4), np.uint8)
        hip.scale_blend(sc, src, got)
        assert np.array_equal(got, want), (kernel, int(np.count_nonzero(got != want)))
    sc.close()


@pytest.mark.parametrize("kind", ["photo", "alpha", "noise"])
@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 67, 50), (1000, 1000, 100, 100), (2048, 1536, 200, 150),
                                         (1600, 1200, 133, 100),
                                         # strips whose source window is wider than 512 columns: four 16-byte loads
                                         # per lane and row, two rows in flight (the LOADS = 4 instantiations)
                                         (4096, 300, 256, 150), (4608, 240, 256, 120), (5120, 128, 280, 64)])
def test_horizontal_first_streaming_kernel(hip, oracle, kind, sw, sh, dw, dh):
    """Plans for which stb resamples horizontally first (BASELINE configs 1 and 5 are of that
    kind) have their own streaming kernel: every channel set (opaque / premultiplied / full)
    against the oracle, plus the generic kernel."""
    src = synth.make(kind, sw, sh, seed=sw + dh)
    sc = hip.scaler(sw, sh, dw, dh)
    info = sc.info()
    assert info["vertical_first"] == 0 and info["streaming_ok"] == 1
    want = oracle.scale(src, dw, dh)
    for kernel in (2, 3, 4, 1):
        sc.set_kernel(kernel)
        got = np.empty((dh, dw, 4), np.uint8)
        hip.scale_blend(sc, src, got)
        assert np.array_equal(got, want), (kernel, int(np.count_nonzero(got != want)))
    sc.close()


def test_all_four_matrix_kernel_instantiations(hip, oracle):
    """ScaleStreamMKernel<opaque | premultiplied, without | with the overflow row>: each of the four is
    a separately compiled kernel with its own register budget and its own ring of rows in flight
    (check_ring_isa.py proves the ring on the assembly; this proves the bytes).  The shapes are picked
    by what the scaler reports, so the test fails if none of them reaches an instantiation."""
    shapes = [(1280, 720, 800, 450), (640, 480, 600, 450), (1000, 1000, 990, 999), (1000, 1000, 700, 700),
              (1000, 1000, 800, 800), (1000, 1000, 900, 900), (300, 300, 290, 290), (1000, 1000, 600, 600),
              (900, 700, 640, 500), (1024, 1024, 1000, 1000), (777, 555, 500, 400),
              # (ratios whose plans have five output rows live at once)
              (400, 400, 57, 57), (480, 480, 70, 70), (1000, 1000, 160, 160), (720, 720, 110, 110),
              (1366, 768, 200, 112), (2000, 1400, 300, 200)]
    seen = set()
    for sw, sh, dw, dh in shapes:
        sc = hip.scaler(sw, sh, dw, dh)
        info = sc.info()
        if not (info["streaming_ok"] and info["matrix_kernel"]):
            sc.close()
            continue
        sc.set_kernel(2)
        for kind in ("photo", "alpha"):
            if (kind, info["matrix_overflow_row"]) in seen and sw > 1000:
                continue
            src = synth.make(kind, sw, sh, seed=sw + dh)
            got = np.empty((dh, dw, 4), np.uint8)
            hip.scale_blend(sc, src, got)
            want = oracle.scale(src, dw, dh)
            assert np.array_equal(got, want), (sw, sh, dw, dh, kind, info, int(np.count_nonzero(got != want)))
            seen.add((kind, info["matrix_overflow_row"]))
        sc.close()
    assert seen == {("photo", 0), ("photo", 1), ("alpha", 0), ("alpha", 1)}, seen


def test_two_column_horizontal_first_kernel_and_its_fallback(hip, oracle, monkeypatch):
    """Round 6: horizontal-first plans whose columns share most of their taps (8K -> 800: 39 taps, columns 9.6 pixels
    apart) run ScaleStreamH2Kernel -- two output columns per lane pair, strips of 64 columns, the few columns at the
    clamped edges alone in their pair -- for the opaque and the premultiplied channel set; tiles that need all seven
    channels (a filtered alpha below 2^-120 that is NOT composed away) fall back to the one-column kernel on the two
    halves of their strip.  One frame with all three kinds of tiles, uncomposed and composed, against the reference's
    arithmetic; the same frame with the kernel switched off (TIMG_HIP_H2=0) gives the same bytes."""
    sw, sh, dw, dh = 7680, 1000, 800, 100
    assert oracle.plan_info(sw, sh, dw, dh)["vertical_first"] == 0
    src = synth.photo(sw, sh, seed=41)
    a = synth.alpha(sw, sh, seed=42)
    src[:, 2600:5200] = a[:, 2600:5200]          # a band of columns with real alpha
    src[200:700, 5200:7000, 3] = 0               # fully transparent: filtered alpha 0 -> the straight RGB sums are needed
    want = oracle.scale(src, dw, dh)
    sc = hip.scaler(sw, sh, dw, dh)
    info = sc.info()
    assert info["streaming_ok"] == 1 and info["vertical_first"] == 0 and info["two_column_kernel"] == 1, info
    got = np.empty((dh, dw, 4), np.uint8)
    hip.scale_blend(sc, src, got, 1, None)
    assert np.array_equal(got, want)
    blend = timg_amd.Blend.make(BG, PAT, 18, 18)
    hip.scale_blend(sc, src, got, 1, blend)
    assert np.array_equal(got, oracle.alpha_compose(want, BG, PAT, 18, 18)[0])
    sc.close()
    monkeypatch.setenv("TIMG_HIP_H2", "0")
    sc0 = hip.scaler(sw, sh, dw, dh)
    assert sc0.info()["two_column_kernel"] == 0
    got0 = np.empty((dh, dw, 4), np.uint8)
    hip.scale_blend(sc0, src, got0, 1, None)
    assert np.array_equal(got0, want)
    sc0.close()
    # other widths that pair up (first step 4 or 5: 33 to 40 taps) and a bgra source
    for (w2, d2) in ((7000, 800), (6601, 777), (8000, 800)):
        s2 = synth.alpha(w2, 400, seed=w2)
        sc2 = hip.scaler(w2, 400, d2, 40, in_fmt=1)
        if oracle.plan_info(w2, 400, d2, 40)["vertical_first"] == 0:
            g2 = np.empty((40, d2, 4), np.uint8)
            hip.scale_blend(sc2, s2, g2, 1, None)
            assert np.array_equal(g2, oracle.scale(s2, d2, 40, in_fmt=1)), (w2, d2, sc2.info())
        sc2.close()


def test_two_column_kernel_on_random_geometries(hip, oracle):
    """Sixty random horizontal-first geometries over the two-column kernel's whole range (17 to 40 taps: ratios 4.2 to 10.6,
    first steps 2 to 5, two to four loads a row), odd widths, b,g,r,a sources, frames with opaque / alpha / fully
    transparent regions, composed and not -- byte for byte against the restatement (the long form of this sweep is
    scratch/r6_h2stress.py: 1 776 geometries)."""
    import random
    rng = random.Random(66)
    n = on_h2 = 0
    while n < 60:
        dw = rng.randint(40, 900)
        ratio = rng.uniform(4.2, 10.6)
        sw = max(dw + 1, int(dw * ratio) + rng.randint(-3, 3))
        dh = rng.randint(6, 40)
        sh = max(dh, int(dh * (rng.uniform(8.0, 11.0) if ratio > 8 else rng.uniform(1.0, 12.0))))
        if oracle.plan_info(sw, sh, dw, dh)["vertical_first"]:
            continue
        kind = rng.choice(["photo", "alpha", "mixed"])
        src = synth.photo(sw, sh, seed=n) if kind == "photo" else synth.alpha(sw, sh, seed=n)
        if kind == "mixed":
            src[:, : sw // 3] = synth.photo(sw, sh, seed=n + 1000)[:, : sw // 3]
            src[sh // 4: sh // 2, sw // 2: sw // 2 + sw // 5, 3] = 0
        fmt = rng.randint(0, 1)
        sc = hip.scaler(sw, sh, dw, dh, in_fmt=fmt)
        on_h2 += sc.info()["two_column_kernel"]
        want = oracle.scale(src, dw, dh, in_fmt=fmt)
        got = np.empty((dh, dw, 4), np.uint8)
        blend = None if rng.random() < 0.5 else timg_amd.Blend.make(BG, PAT, rng.randint(1, 20), rng.randint(1, 20))
        hip.scale_blend(sc, src, got, 1, blend)
        if blend is not None:
            want = oracle.alpha_compose(want, BG, PAT, blend.pattern_w, blend.pattern_h)[0]
        assert np.array_equal(got, want), (sw, sh, dw, dh, kind, fmt, sc.info())
        sc.close()
        n += 1
    assert on_h2 >= 40, on_h2  # (most of them really ran on the two-column kernel)


@pytest.mark.parametrize("sw,sh,dw,dh", [(1366, 768, 200, 112), (1366, 768, 455, 256), (999, 1333, 333, 444),
                                         (1023, 767, 341, 255), (6, 1000, 3, 100), (5, 500, 2, 100),  # vertical-first
                                         (1001, 999, 100, 100), (1275, 1650, 150, 194), (2561, 1441, 320, 180),
                                         (1365, 767, 91, 51)])                                        # horizontal-first
def test_streaming_kernels_take_any_source_width(hip, oracle, sw, sh, dw, dh):
    """Source widths that are not a multiple of 4: rows are only 4-byte aligned and one lane's
    16-byte load straddles the end of the row."""
    for kind in ("alpha", "photo"):
        src = synth.make(kind, sw, sh, seed=sw)
        sc = hip.scaler(sw, sh, dw, dh)
        assert sc.info()["streaming_ok"] == 1, (sw, sh, dw, dh)
        want = oracle.scale(src, dw, dh)
        for kernel in (2, 4, 1):
            sc.set_kernel(kernel)
            got = np.empty((dh, dw, 4), np.uint8)
            hip.scale_blend(sc, src, got)
            assert np.array_equal(got, want), (kind, kernel, int(np.count_nonzero(got != want)))
        sc.close()


def test_horizontal_first_batch_with_bgra_and_blend(hip, oracle):
    n, sw, sh, dw, dh = 3, 1280, 960, 120, 90
    frames = np.stack([synth.alpha(sw, sh, seed=70 + i) for i in range(n)])
    sc = hip.scaler(sw, sh, dw, dh, in_fmt=1)
    assert sc.info()["vertical_first"] == 0 and sc.info()["streaming_ok"] == 1
    blend = timg_amd.Blend.make(BG, PAT, 7, 5, start_row=3)
    d_src = hip.upload(frames)
    d_dst = hip.malloc(n * dw * dh * 4)
    flags = hip.scale_blend(sc, d_src, d_dst, n, blend, want_transparent=True)
    hip.sync()
    out = hip.download(d_dst, n * dw * dh * 4).reshape(n, dh, dw, 4)
    assert flags == [1] * n
    for i in range(n):
        want, _ = oracle.alpha_compose(oracle.scale(frames[i], dw, dh, 1), BG, PAT, 7, 5, 3)
        assert np.array_equal(out[i], want), i
    hip.free(d_src)
    hip.free(d_dst)
    sc.close()


@pytest.mark.parametrize("no_matrix", [0, 1])
@pytest.mark.parametrize("sw,sh,dw,dh", [(1280, 720, 400, 225), (1280, 960, 120, 90), (2048, 1536, 200, 150)])
def test_composed_frames_with_transparent_pixels(hip, oracle, monkeypatch, no_matrix, sw, sh, dw, dh):
    """Frames with fully transparent pixels (S-alpha: a transparent border and 2.5 % zero alphas) that are composed
    over a background stay on the premultiplied channel set in all three streaming kernels (matrix-core, all-VALU
    [TIMG_HIP_NO_MATRIX], horizontal-first): a filtered alpha below 2^-120 ends as alpha byte 0 = the background
    alone.  Bands above start_row still need the straight sums.  Bytes against the oracle, all channel sets."""
    monkeypatch.setenv("TIMG_HIP_NO_MATRIX", str(no_matrix))
    src = synth.alpha(sw, sh, seed=sw + 3 * dh)
    sc = hip.scaler(sw, sh, dw, dh)
    assert sc.info()["streaming_ok"] == 1
    scaled = oracle.scale(src, dw, dh)
    for blend_args in ((BG,), (BG, PAT, 5, 7), (BG, PAT, 4, 4, 31)):
        blend = timg_amd.Blend.make(*blend_args)
        want, _ = oracle.alpha_compose(scaled.copy(), *blend_args)
        for kernel in (2, 3, 4, 1):
            sc.set_kernel(kernel)
            got = np.empty((dh, dw, 4), np.uint8)
            hip.scale_blend(sc, src, got, 1, blend)
            assert np.array_equal(got, want), (blend_args, kernel, int(np.count_nonzero(got != want)))
    sc.close()


def test_streaming_batch_with_blend_matches_generic(hip):
    """Size-independent property at full batch shape: both kernel families give
    identical bytes on device-resident frames (the generic one is pinned to the
    oracle above)."""
    import torch
    from timg_amd.pipeline import synth_frames_on_device
    n = 6
    src = synth_frames_on_device(n, 3840, 2160, "alpha", seed=3)
    sc = hip.scaler(3840, 2160, 800, 450)
    blend = timg_amd.Blend.make(BG, PAT, 9, 9)
    outs = []
    for kernel in (1, 2):
        sc.set_kernel(kernel)
        dst = torch.zeros((n, 450, 800, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        hip.scale_blend(sc, src.data_ptr(), dst.data_ptr(), n, blend)
        hip.sync()
        outs.append(dst.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    sc.close()


def test_random_geometries_streaming_equals_generic_and_oracle(hip, oracle):
    """200 seeded random cases (source sizes 5..2000, ratios 1..38 per axis incl. the one- and two-pixel shrinks
    around 1:1, batches of 1-5, opaque / alpha / noise / opaque-with-an-alpha-block content, no / solid /
    checkerboard background): the streaming kernels (matrix-core, all-VALU, horizontal-first -- the plan chooses)
    give the generic kernel's bytes on device-resident batches, and every tenth case is checked against the
    oracle as well.  (scratch/scale_stress.py is the open-ended form: 84 580 cases without a difference.)"""
    import random
    import torch
    rnd = random.Random(2024)
    for case in range(200):
        sw, sh = rnd.randint(5, 2000), rnd.randint(5, 1400)
        rx, ry = rnd.choice([1.0, 1.3, 2.0, 3.7, 4.8, 9.6, 19.2]), rnd.choice([1.0, 1.3, 2.0, 3.7, 4.8, 9.6, 38.0])
        dw, dh = max(1, int(sw / rx) - rnd.randint(0, 2)), max(1, int(sh / ry) - rnd.randint(0, 2))
        n = rnd.choice([1, 2, 5])
        kind = rnd.choice(["photo", "alpha", "noise", "mixed"])
        src = torch.empty((n, sh, sw, 4), dtype=torch.uint8, device="cuda")
        hip.synth_frames("alpha" if kind == "mixed" else kind, sw, sh, seed=case, first_frame=0, n_frames=n, dst=src.data_ptr())
        hip.sync()
        if kind == "mixed":
            src[..., 3] = 255
            y0, x0 = rnd.randrange(sh), rnd.randrange(sw)
            src[:, y0:y0 + max(1, sh // 7), x0:x0 + max(1, sw // 5), 3] = 77
        sc = hip.scaler(sw, sh, dw, dh)
        info = sc.info()
        blend = rnd.choice([None, timg_amd.Blend.make(BG), timg_amd.Blend.make(BG, PAT, 5, 7)])
        outs = []
        for kernel in ((1, 2) if info["streaming_ok"] else (1,)):
            sc.set_kernel(kernel)
            dst = torch.zeros((n, dh, dw, 4), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()  # (torch fills on its own stream; the library's stream is non-blocking)
            hip.scale_blend(sc, src.data_ptr(), dst.data_ptr(), n, blend)
            hip.sync()
            outs.append(dst.cpu().numpy())
        sc.close()
        what = (case, sw, sh, dw, dh, n, kind, info)
        if len(outs) == 2:
            assert np.array_equal(outs[0], outs[1]), what
        if case % 10 == 0:
            host = src.cpu().numpy()
            for i in range(n):
                want = oracle.scale(host[i], dw, dh)
                if blend is not None:
                    want, _ = oracle.alpha_compose(want, BG, PAT if blend.pattern_w else (0, 0, 0, 0), blend.pattern_w,
                                                   blend.pattern_h, 0)
                assert np.array_equal(outs[-1][i], want), what + (i,)


def test_baseline_config1_640x480_half_block(hip, oracle):
    """BASELINE.json config 1: 640x480 RGBA -> -p half -g80x25 = 67x50 px -> 67 columns x 25 rows
    (geometry C1), scale + alpha compose + half-block bytes end to end."""
    src = synth.alpha(640, 480, seed=1)
    blend = timg_amd.Blend.make(BG)
    fb = hip.scale(src, 67, 50, blend=blend)
    want_fb, _ = oracle.alpha_compose(oracle.scale(src, 67, 50), BG)
    assert np.array_equal(fb, want_fb)
    got = hip.block_encode(fb, 67, 50, flags=0)[0]
    assert got == oracle.block_encode(want_fb)
    assert got.count(b"\n") == 25


def test_baseline_config3_grid_of_4k_frames_quarter_block(hip, oracle):
    """BASELINE.json config 3 (a grid row of it): 4K frames -> 200x56 px -> -p quarter, 100x28 cells
    each, placed side by side (Send's x = column * 202 px), device-resident batch, one launch
    per stage; bit-exact bytes per frame."""
    import torch
    n = 4
    frames = np.stack([synth.make(k, 3840, 2160, seed=30 + i) for i, k in enumerate(["photo", "alpha", "noise", "photo"])])
    src = torch.from_numpy(frames).cuda()
    sc = hip.scaler(3840, 2160, 200, 56)
    blend = timg_amd.Blend.make(BG, PAT, 2, 2)
    dst = torch.empty((n, 56, 200, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    hip.scale_blend(sc, src.data_ptr(), dst.data_ptr(), n, blend)
    hip.sync()
    scaled = dst.cpu().numpy()
    for i in range(n):
        want, _ = oracle.alpha_compose(oracle.scale(frames[i], 200, 56), BG, PAT, 2, 2)
        assert np.array_equal(scaled[i], want), i
        # every grid column has its own indent, so each is its own (single-frame) encode call
        got = hip.block_encode(dst[i].data_ptr(), 200, 56, flags=timg_amd.TimgHip.QUARTER, x_indent=i * 202)[0]
        assert got == oracle.block_encode(want, quarter=True, x=i * 202), i
    outs = hip.block_encode(dst.data_ptr(), 200, 56, flags=timg_amd.TimgHip.QUARTER, n_frames=n)
    for i in range(n):
        assert outs[i] == oracle.block_encode(scaled[i], quarter=True), i
    # the whole grid row in ONE launch, every frame with its own column offset
    xs = [i * 202 for i in range(n)]
    outs = hip.block_encode(dst.data_ptr(), 200, 56, flags=timg_amd.TimgHip.QUARTER, n_frames=n, x_indents=xs)
    for i in range(n):
        assert outs[i] == oracle.block_encode(scaled[i], quarter=True, x=xs[i]), i
    sc.close()


def test_baseline_config5_8k_alpha_checkerboard_sixel(hip, oracle):
    """BASELINE.json config 5, one frame: 7680x4320 RGBA with alpha -> 800x450 with a
    checkerboard background (-b colour -B colour) -> sixel, against the oracle end to end."""
    src = synth.alpha(7680, 4320, seed=5)
    blend = timg_amd.Blend.make(BG, PAT, 18, 18)
    got = hip.scale(src, 800, 450, blend=blend)
    want, _ = oracle.alpha_compose(oracle.scale(src, 800, 450), BG, PAT, 18, 18)
    assert np.array_equal(got, want)
    six = hip.sixel_encode(got, 800, 450, pad_blend=blend)[0]
    assert six == oracle.sixel_encode(want, BG, PAT, 18, 18, lookup_mode=1)


def test_baseline_config4_video_frames_round_robin(hip, oracle):
    """BASELINE.json config 4 in miniature: a stream of frames sharded round-robin over
    `world` ranks, every rank encodes its share as one batch, the gather restores stream
    order.  (Ranks are emulated in-process: the exchange itself is covered by
    tests/test_gather_gloo.py, the GPUs by bench.py --gpus N.)"""
    from timg_amd.gather import shard_frames
    n, world, w, h = 12, 4, 160, 90
    frames = [synth.photo(w, h, seed=100 + i) for i in range(n)]
    per_rank = []
    for rank in range(world):
        mine = shard_frames(n, world, rank, round_robin=True)
        batch = np.stack([frames[i] for i in mine])
        per_rank.append(dict(zip(mine, hip.sixel_encode(batch, w, h, n_frames=len(mine),
                                                        pad_blend=timg_amd.Blend.make(BG)))))
    for i in range(n):
        assert per_rank[i % world][i] == oracle.sixel_encode(frames[i], BG), i


def test_streaming_fallback_chain_on_mixed_tiles(hip, oracle):
    """An opaque frame with one translucent patch and one fully transparent
    patch: tiles fall through opaque -> premultiplied -> full channel sets."""
    src = synth.photo(1920, 1080, seed=12)
    src[100:300, 200:900, 3] = 128          # translucent: needs A, RA, GA, BA
    src[600:1000, 1000:1800, 3] = 0         # transparent: filtered alpha == 0 -> straight RGB
    want = oracle.scale(src, 400, 225)
    sc = hip.scaler(1920, 1080, 400, 225)
    assert sc.info()["streaming_ok"] == 1
    got = np.empty((225, 400, 4), np.uint8)
    hip.scale_blend(sc, src, got)
    assert np.array_equal(got, want)
    sc.close()


# ---- graphics protocols at --compress=0: png / kitty / iTerm2 -----------------
# The oracle (oracle/png.c) is pinned against the real png::Encode + libdeflate and the real
# canvases (tests/test_png_oracle.py); the golden vectors come from the real reference.
GFX_SIZES = [(1, 1), (5, 3), (67, 50), (200, 56), (129, 127), (400, 300), (4095, 5), (800, 450)]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", GFX_SIZES)
def test_png_kitty_iterm2_bytes_match_oracle(hip, oracle, w, h):
    fb = synth.make("noise" if (w + h) % 2 else "alpha", w, h, seed=w * 3 + h)
    for rgb24 in (False, True):
        assert hip.gfx_encode("png", fb, w, h, rgb24=rgb24)[0] == oracle.png_encode(fb, not rgb24), (w, h, rgb24)
        got = hip.gfx_encode("kitty", fb, w, h, rgb24=rgb24, image_ids=[4_000_000_123])[0]
        assert got == oracle.kitty_encode(fb, 4_000_000_123, not rgb24), (w, h, rgb24)
        assert hip.gfx_encode("iterm2", fb, w, h, rgb24=rgb24)[0] == oracle.iterm2_encode(fb, not rgb24), (w, h, rgb24)


@pytest.mark.gpu
def test_gfx_batch_device_resident(hip, oracle):
    import torch
    w, h, n = 200, 90, 5
    frames = np.stack([synth.make("photo", w, h, seed=40 + i) for i in range(n)])
    dev = torch.from_numpy(frames).cuda()
    cap = hip.L.timg_hip_gfx_max_bytes(w, h)
    out = torch.zeros((n, cap), dtype=torch.uint8, device="cuda")
    ids = [7, 70, 700, 7000, 70000]
    lens = hip.gfx_encode("kitty", dev.data_ptr(), w, h, n_frames=n, image_ids=ids, out=out.data_ptr(), out_cap=cap)
    host = out.cpu().numpy()
    for i in range(n):
        assert host[i, :lens[i]].tobytes() == oracle.kitty_encode(frames[i], ids[i], True), i
    pngs = hip.gfx_encode("png", frames, w, h, n_frames=n)
    for i in range(n):
        assert pngs[i] == oracle.png_encode(frames[i], True), i


@pytest.mark.gpu
def test_gfx_golden_vectors_from_the_real_reference(hip):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "png.npz"))
    for i in range(int(g["count"])):
        fb, with_alpha = np.ascontiguousarray(g[f"fb{i}"]), bool(g[f"alpha{i}"])
        h, w = fb.shape[:2]
        assert hip.gfx_encode("png", fb, w, h, rgb24=not with_alpha)[0] == g[f"png{i}"].tobytes(), i
        assert hip.gfx_encode("kitty", fb, w, h, rgb24=not with_alpha,
                              image_ids=[int(g[f"id{i}"])])[0] == g[f"kitty{i}"].tobytes(), i
        assert hip.gfx_encode("iterm2", fb, w, h, rgb24=not with_alpha)[0] == g[f"iterm{i}"].tobytes(), i


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["noise", "photo", "alpha"])
def test_synthetic_frames_device_equals_host(hip, kind):
    """timg_hip_synth_frames (what bench.py and HipRawRGBASource's synth: names run on) is the same
    integer function of (kind, seed, frame, x, y) as timg_amd.synth.hash_frame."""
    for w, h, seed, first, n in [(64, 48, 0, 0, 3), (333, 77, 9, 5, 2), (1, 1, 3, 0, 1), (3840, 2160, 1, 63, 1)]:
        got = hip.synth_frames(kind, w, h, seed, first, n)
        for i in range(n):
            assert np.array_equal(got[i], synth.hash_frame(kind, w, h, seed, first + i)), (kind, w, h, seed, first + i)
    # device-resident destination with a frame stride
    w, h = 100, 40
    stride = w * h * 4 + 256
    d = hip.malloc(stride * 2)
    hip.synth_frames(kind, w, h, 4, 10, 2, dst=d, frame_stride=stride)
    hip.sync()
    raw = hip.download(d, stride * 2)
    hip.free(d)
    for i in range(2):
        frame = raw[i * stride:i * stride + w * h * 4].reshape(h, w, 4)
        assert np.array_equal(frame, synth.hash_frame(kind, w, h, 4, 10 + i))
