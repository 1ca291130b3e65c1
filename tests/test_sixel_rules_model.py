"""Host-side models of two restatements inside timg_amd/csrc/sixel_canvas.hip (round 4).  The kernels themselves are
checked byte for byte against oracle/sixel.c on the GPU (tests/test_gpu_parity.py); these tests check the RULES the
kernels are built on, exhaustively or on random states, where no GPU is needed:

* BuildLutKernel compares a cell only with the palette entries whose smallest squared distance to the box of its
  coarse cell (4 x 4 x 4 cells) is not above the smallest of the entries' LARGEST distances to that box -- the claim
  is that the minimum (first of equals) over those candidates is the minimum over the whole palette, for every cell;
* MedianCutKernel does libsixel's list bookkeeping (src/sixel-canvas.cc:137-145 -> libsixel's quant.c: "take the first
  box of the list that can be split; split it; replace it by its low half, append the high half; stable-sort by sum")
  for a whole round at once: prepared boxes in descending key order, the leading ones whose key is above every
  unprepared key -- those of the list and those of the halves made in front of them -- are the ones that get split.
"""
import random

import numpy as np


# ---- nearest-colour table ---------------------------------------------------------------------------------------
def _exhaustive(palette):
    """index of the nearest palette entry (first of equals) for all 32768 cells, centres (c5 << 3) | 4"""
    c5 = np.arange(32768)
    centre = np.stack([((c5 >> 10) & 31) << 3 | 4, ((c5 >> 5) & 31) << 3 | 4, (c5 & 31) << 3 | 4], axis=1).astype(np.int64)
    d = ((centre[:, None, :] - palette[None, :, :].astype(np.int64)) ** 2).sum(axis=2)
    return d.argmin(axis=1)  # (numpy's argmin returns the first minimum)


def _pruned(palette):
    out = np.zeros(32768, dtype=np.int64)
    sizes = []
    pal = palette.astype(np.int64)
    for coarse in range(512):
        lo = np.array([(coarse >> 6) & 7, (coarse >> 3) & 7, coarse & 7]) * 32 + 4
        hi = lo + 24
        near = np.maximum(0, np.maximum(lo - pal, pal - hi))
        far = np.maximum(np.abs(pal - lo), np.abs(pal - hi))
        d_min, d_max = (near ** 2).sum(axis=1), (far ** 2).sum(axis=1)
        cand = np.nonzero(d_min <= d_max.min())[0]  # (ascending: the first of equals stays the first)
        sizes.append(len(cand))
        for lane in range(64):
            cell = (((coarse >> 6) & 7) << 12) | (((lane >> 4) & 3) << 10) | (((coarse >> 3) & 7) << 7) | \
                   (((lane >> 2) & 3) << 5) | ((coarse & 7) << 2) | (lane & 3)
            centre = np.array([((cell >> 10) & 31) << 3 | 4, ((cell >> 5) & 31) << 3 | 4, (cell & 31) << 3 | 4])
            d = ((pal[cand] - centre) ** 2).sum(axis=1)
            out[cell] = cand[d.argmin()]
    return out, sizes


def test_pruned_nearest_colour_search_equals_the_exhaustive_one():
    rng = np.random.default_rng(7)
    palettes = [
        rng.integers(0, 256, size=(256, 3)),                       # spread over the cube
        rng.integers(96, 160, size=(256, 3)),                      # crowded in the middle: many ties and near-ties
        np.repeat(rng.integers(0, 256, size=(16, 3)), 16, axis=0),  # every colour sixteen times: ties by index
        rng.integers(0, 256, size=(3, 3)),                         # a tiny palette
        (rng.integers(0, 32, size=(200, 3)) << 3),                 # colours on the 5-bit grid, as a median cut makes them
    ]
    for palette in palettes:
        want = _exhaustive(palette)
        got, sizes = _pruned(palette)
        assert np.array_equal(got, want)
        assert max(sizes) <= len(palette) and min(sizes) >= 1


# ---- median cut bookkeeping -------------------------------------------------------------------------------------
class _Box:
    def __init__(self, colors, total, tie):
        self.colors, self.sum, self.tie = colors, total, tie
        self.ready = False      # its split has been prepared
        self.median = self.lower = None

    def key(self):
        return (self.sum << 9) | (511 - self.tie)

    def prepare(self, rng):
        self.median = rng.randint(1, self.colors - 1)                      # colours of the low half
        self.lower = rng.randint(0, self.sum)                              # its pixels
        self.ready = True


def _children(box, nb):
    lo = _Box(box.median, box.lower, 256 - nb)
    hi = _Box(box.colors - box.median, box.sum - box.lower, 256 + nb)
    return lo, hi


def _sequential(slots, nb, limit):
    """the replay of round 3: one split at a time, as libsixel orders them"""
    log = []
    while nb < limit:
        best = max((b for b in slots if b is not None and b.colors >= 2), key=lambda b: b.key(), default=None)
        if best is None or not best.ready:
            break
        i = slots.index(best)
        lo, hi = _children(best, nb)
        slots[i], slots[nb] = lo, hi
        log.append((i, nb, lo.key(), hi.key()))
        nb += 1
    return log, nb


def _one_pass(slots, nb, limit):
    """the bookkeeping of round 4: ranked prepared boxes, exclusive prefix maximum over the halves' keys"""
    ready = sorted((b for b in slots if b is not None and b.colors >= 2 and b.ready), key=lambda b: -b.key())
    u0 = max((b.key() for b in slots if b is not None and b.colors >= 2 and not b.ready), default=0)
    log, above = [], 0
    for i, b in enumerate(ready):
        nb_i = nb + i
        if not (b.key() > u0 and b.key() > above and nb_i < limit):
            break
        lo, hi = _children(b, nb_i)
        log.append((slots.index(b), nb_i, lo.key(), hi.key()))
        above = max(above, lo.key() if lo.colors >= 2 else 0, hi.key() if hi.colors >= 2 else 0)
    for slot, nb_i, _, _ in log:  # (all halves written at once)
        lo, hi = _children(slots[slot], nb_i)
        slots[slot], slots[nb_i] = lo, hi
    return log, nb + len(log)


def test_one_pass_bookkeeping_splits_what_the_sequential_replay_splits():
    rng = random.Random(11)
    for trial in range(300):
        limit = 256
        slots_a = [None] * limit
        # a list in the middle of a cut: some boxes, distinct ties, some prepared
        nb = rng.randint(1, 200)
        # (ties as a cut makes them: 256 for the root, 256 -+ step for the halves of an earlier step -- all distinct,
        # and below what the steps from nb on will hand out)
        ties = rng.sample([256] + [256 - k for k in range(1, nb)] + [256 + k for k in range(1, nb)], nb)
        for i in range(nb):
            colors = rng.choice([1, 1, 2, 3, 5, 40, 700])
            total = rng.choice([rng.randint(colors, colors + 3), rng.randint(colors, 30000)])  # (equal sums happen: ties decide)
            slots_a[i] = _Box(colors, total, ties[i])
        for b in slots_a[:nb]:
            if b.colors >= 2 and rng.random() < 0.6:
                b.prepare(rng)
        import copy
        slots_b = copy.deepcopy(slots_a)
        log_a, nb_a = _sequential(slots_a, nb, limit)
        log_b, nb_b = _one_pass(slots_b, nb, limit)
        assert log_a == log_b and nb_a == nb_b, trial


def test_threshold_picks_are_the_head_of_the_list():
    """pick(): every lane's largest unprepared slot word is ranked among the 64 lane maxima, the word of rank P - 1 is
    the threshold, every unprepared word at or above it is picked.  Claim: the picks are exactly the first |picks|
    unprepared boxes of the list (descending words), P <= |picks| <= 4 P, and the fallback P = max(1, room / 4) fits."""
    rng = random.Random(5)
    for trial in range(400):
        n = rng.randint(1, 256)
        words = [0] * 256                       # slot q * 64 + lane  ->  lane-major: lane = slot & 63
        for slot in rng.sample(range(256), n):
            words[slot] = rng.randint(1, 1 << 27)
        while len({w for w in words if w}) != n:  # (keys are distinct)
            words = [w + (i if w else 0) for i, w in enumerate(words)]
        lane_max = [max(words[q * 64 + lane] for q in range(4)) for lane in range(64)]
        order = sorted((w for w in words if w), reverse=True)
        room = rng.randint(1, 64)
        for P in (min(40, room), max(1, room // 4)):
            ranked = sorted((m for m in lane_max if m), reverse=True)
            thr = ranked[P - 1] if len(ranked) >= P else 1
            picks = sorted((w for w in words if w >= thr and w), reverse=True)
            assert picks == order[:len(picks)]
            assert min(P, len(ranked)) <= len(picks) <= 4 * P
        assert len(picks) <= max(room, 1)  # (the second choice of P: at most four boxes a lane)


def test_index_rows_make_every_group_of_eight_aligned():
    """IndexRow(): pixel x of row r lives at byte r * stride + 2 (r & 3) + x.  The diffusion's lane pair rl = r mod 32
    works on column t + k - 2 rl at step t + k and has the indices of columns t - 8 - 2 rl ... t - 1 - 2 rl complete at the
    last step of a block of eight (t a multiple of 8): their bytes start at t - 8 - 8 (rl >> 2), a multiple of eight for
    every row -- one aligned 8-byte store per lane, all lanes in the same step."""
    for r in range(0, 96):
        rl = r % 32
        for t in range(0, 200, 8):
            first_px = t - 8 - 2 * rl
            byte = 2 * (r & 3) + first_px
            assert byte == t - 8 - 8 * (rl >> 2) and byte % 8 == 0
