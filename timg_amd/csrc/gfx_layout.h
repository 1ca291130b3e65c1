// timg_amd/csrc/gfx_layout.h -- per-element arithmetic of the graphics-protocol path at
// --compress=0 (png::Encode over stored deflate blocks, base64, kitty / iTerm2 framing;
// src/timg-png.cc:91-153, src/timg-base64.h:28-55, src/kitty-canvas.cc:167-214,
// src/iterm2-canvas.cc:52-71).
//
// Everything here is a pure function of an index, usable from host and device alike: the
// kernels of gfx_canvas.hip call these per lane, and the host-only debug entry point
// timg_hip_debug_gfx_emulate (debug_api.hip) walks the same indices in plain loops, so the
// whole layout -- offsets, block headers, checksum composition, chunk framing -- is checked
// on the CPU against the real reference before a GPU is involved (tests/test_gfx_layout.py).
#ifndef TIMG_AMD_GFX_LAYOUT_H_
#define TIMG_AMD_GFX_LAYOUT_H_

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#if defined(__HIPCC__)
#define TIMG_HD __host__ __device__ inline
#else
#define TIMG_HD inline
#endif

namespace timg_amd {

constexpr uint32_t kStoredBlock  = 65535;  // libdeflate level 0: stored blocks of at most this
constexpr uint32_t kPngIdatData  = 8 + 25 + 8;  // signature, IHDR chunk, IDAT length + type
constexpr uint32_t kCrcChunk     = 512;    // bytes per lane in the chunk CRC pass
constexpr uint32_t kCrcSegmentLog = 10;    // a workgroup combines 2^10 chunk CRCs as a binary tree ...
constexpr uint32_t kCrcSegment   = 1u << kCrcSegmentLog;
constexpr uint32_t kCrcLevels    = 24;     // ... and the frame's tree has at most this many levels
constexpr uint32_t kKittyChunk   = 3072;   // bytes per kitty escape (4096 base64 characters)
constexpr uint32_t kKittySepLen  = 13;     // ESC \ ESC _ G q = 2 , m = X ;
constexpr uint32_t kGfxHeaderCap = 96;

enum GfxKind { kGfxPng = 0, kGfxKitty = 1, kGfxIterm2 = 2 };

struct PngGeom {
    int w, h, bpp;         // bpp: 4 (RGBA) or 3 (RGB, alpha dropped)
    uint32_t row;          // 1 + w * bpp filtered bytes per row
    uint32_t raw_n;        // h * row
    uint32_t n_blocks;     // stored blocks
    uint32_t zlen;         // zlib stream: 2 + 5 * n_blocks + raw_n + 4
    uint32_t png_n;        // whole file: 57 + zlen
    uint32_t crc_len;      // "IDAT" + zlib stream
    uint32_t n_chunks;     // CRC chunks of kCrcChunk bytes
    uint32_t n_segments;   // groups of kCrcSegment chunks
    // The chunk CRCs are combined as a binary tree by index: the node r of level k covers chunks
    // [r 2^k, (r + 1) 2^k) (cut off at n_chunks).  crc(A || B) = x^(8 |B|) crc(A) + crc(B): the shift a
    // level-(k+1) node applies to its left child is x^(8 * 512 * 2^k) for a complete right child and
    // x^(8 * len) for the one right child that contains the (short) last chunk -- node edge_node[k]
    uint32_t x_full[kCrcLevels], x_edge[kCrcLevels], edge_node[kCrcLevels];
    uint8_t head[kPngIdatData + 2];  // signature, IHDR (with its CRC), IDAT length + "IDAT", 78 01
};

TIMG_HD uint32_t MultModP(uint32_t a, uint32_t b) {  // a(x) * b(x) mod the (reflected) CRC-32 polynomial
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xedb88320u : b >> 1;
    }
    return p;
}

TIMG_HD uint32_t XPowBytes(uint64_t n_bytes) {  // x^(8 n) mod P
    uint32_t sq = 1u << 30, r = 1u << 31;
    uint64_t e = n_bytes * 8;
    while (e) {
        if (e & 1) r = MultModP(sq, r);
        sq = MultModP(sq, sq);
        e >>= 1;
    }
    return r;
}

TIMG_HD uint32_t Crc32Bytes(const uint8_t *p, uint32_t n) {  // standard CRC-32 of n bytes
    uint32_t crc = 0xffffffffu;
    for (uint32_t i = 0; i < n; ++i) {
        crc ^= p[i];
        for (int k = 0; k < 8; ++k) crc = (crc & 1u) ? (crc >> 1) ^ 0xedb88320u : crc >> 1;
    }
    return ~crc;
}

TIMG_HD void PutBE32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

inline PngGeom MakePngGeom(int w, int h, bool with_alpha) {
    PngGeom g{};
    g.w   = w;
    g.h   = h;
    g.bpp = with_alpha ? 4 : 3;
    g.row      = 1u + (uint32_t)w * g.bpp;
    g.raw_n    = (uint32_t)h * g.row;
    g.n_blocks = (g.raw_n + kStoredBlock - 1) / kStoredBlock;
    g.zlen     = 2 + 5 * g.n_blocks + g.raw_n + 4;
    g.png_n    = 8 + 25 + 12 + g.zlen + 12;
    g.crc_len  = 4 + g.zlen;
    g.n_chunks   = (g.crc_len + kCrcChunk - 1) / kCrcChunk;
    g.n_segments = (g.n_chunks + kCrcSegment - 1) / kCrcSegment;
    for (uint32_t k = 0; k < kCrcLevels; ++k) {
        const uint64_t span = (uint64_t)kCrcChunk << k;  // bytes under a complete level-k node
        const uint64_t last = ((uint64_t)g.crc_len + span - 1) / span - 1;  // the last node of the level
        g.edge_node[k] = (uint32_t)last;
        g.x_full[k]    = XPowBytes(span);
        g.x_edge[k]    = XPowBytes((uint64_t)g.crc_len - last * span);
    }
    static const uint8_t sig[8] = {0x89, 0x50, 0x4e, 0x47, '\r', '\n', 0x1a, '\n'};
    uint8_t *p = g.head;
    for (int i = 0; i < 8; ++i) *p++ = sig[i];
    PutBE32(p, 13);
    p[4] = 'I'; p[5] = 'H'; p[6] = 'D'; p[7] = 'R';
    PutBE32(p + 8, (uint32_t)w);
    PutBE32(p + 12, (uint32_t)h);
    p[16] = 8;
    p[17] = with_alpha ? 6 : 2;
    p[18] = p[19] = p[20] = 0;
    PutBE32(p + 21, Crc32Bytes(p + 4, 17));
    p += 25;
    PutBE32(p, g.zlen);
    p[4] = 'I'; p[5] = 'D'; p[6] = 'A'; p[7] = 'T';
    p[8] = 0x78;
    p[9] = 0x01;
    return g;
}

// ---- PNG body: one filtered byte per index j < raw_n --------------------------------------
// value of filtered byte j (row filter type 1 = Sub, src/timg-png.cc:96,:124-135)
TIMG_HD uint8_t PngRawByte(const uint8_t *frame, size_t stride, const PngGeom &g, uint32_t j) {
    const uint32_t y = j / g.row, i = j - y * g.row;
    if (i == 0) return 1;
    const uint32_t x = (i - 1) / (uint32_t)g.bpp, c = (i - 1) - x * (uint32_t)g.bpp;
    const uint8_t *line = frame + (size_t)y * stride;
    const uint8_t cur   = line[x * 4 + c];
    return x == 0 ? cur : (uint8_t)(cur - line[(x - 1) * 4 + c]);
}
// where filtered byte j lies in the file: behind the zlib header and one 5-byte block header
// per stored block started so far
TIMG_HD uint32_t PngRawOffset(uint32_t j) { return kPngIdatData + 2 + 5 * (j / kStoredBlock + 1) + j; }
// header of stored block b (BFINAL | BTYPE=00, LEN, NLEN) and its place
TIMG_HD uint32_t PngBlockHeaderOffset(uint32_t b) { return kPngIdatData + 2 + b * (5 + kStoredBlock); }
TIMG_HD void PngBlockHeader(const PngGeom &g, uint32_t b, uint8_t hdr[5]) {
    const uint32_t left = g.raw_n - b * kStoredBlock, len = left < kStoredBlock ? left : kStoredBlock;
    hdr[0] = len == left ? 1 : 0;
    hdr[1] = (uint8_t)len;
    hdr[2] = (uint8_t)(len >> 8);
    hdr[3] = (uint8_t)~len;
    hdr[4] = (uint8_t)(~len >> 8);
}
// Adler-32 from the plain sums  A' = sum d_j,  B' = sum (raw_n - j) d_j  over the filtered bytes
TIMG_HD uint32_t AdlerFromSums(const PngGeom &g, unsigned long long sum_a, unsigned long long sum_b) {
    const uint32_t a = (uint32_t)((1ull + sum_a) % 65521ull);
    const uint32_t b = (uint32_t)(((unsigned long long)g.raw_n + sum_b) % 65521ull);
    return (b << 16) | a;
}
TIMG_HD uint32_t PngAdlerOffset(const PngGeom &g) { return kPngIdatData + g.zlen - 4; }
TIMG_HD uint32_t PngCrcOffset(const PngGeom &g) { return kPngIdatData + g.zlen; }
TIMG_HD uint32_t PngCrcRegion() { return kPngIdatData - 4; }  // "IDAT" starts the checksummed bytes
// CRC of chunk c of the checksummed region
TIMG_HD uint32_t PngChunkCrc(const uint8_t *png, const PngGeom &g, uint32_t c) {
    const uint32_t at = c * kCrcChunk, left = g.crc_len - at;
    return Crc32Bytes(png + PngCrcRegion() + at, left < kCrcChunk ? left : kCrcChunk);
}
// crc(A || B) = x^(8|B|) * crc(A) + crc(B): the parent `parent` of level k + 1 from its children of level k
// (child[] holds the level-k nodes from index child_base on)
TIMG_HD uint32_t PngTreeParent(const uint32_t *child, uint32_t child_base, const PngGeom &g, uint32_t k, uint32_t parent) {
    const uint32_t left = 2u * parent, right = left + 1u;
    const uint32_t lv   = child[left - child_base];
    if (right > g.edge_node[k]) return lv;  // no right child: the node is its left child
    return MultModP(right == g.edge_node[k] ? g.x_edge[k] : g.x_full[k], lv) ^ child[right - child_base];
}
// levels k0 .. until one node is left, in place, serially (the host's form; the kernels walk the same levels
// with a lane per parent).  nodes[] holds `count` level-k0 nodes starting at node index `base`.
TIMG_HD uint32_t PngTreeReduce(uint32_t *nodes, uint32_t count, uint32_t base, uint32_t k0, uint32_t levels,
                               const PngGeom &g) {
    for (uint32_t k = k0; k < k0 + levels && k < kCrcLevels; ++k) {
        const uint32_t pbase = base >> 1, parents = ((base + count + 1u) >> 1) - pbase;
        for (uint32_t r = 0; r < parents; ++r) nodes[r] = PngTreeParent(nodes, base, g, k, pbase + r);  // (r <= 2 r: in place)
        base  = pbase;
        count = parents;
    }
    return count;
}
TIMG_HD void PngTail(uint8_t *png, const PngGeom &g, uint32_t crc) {  // IDAT's CRC and the IEND chunk
    uint8_t *p = png + PngCrcOffset(g);
    PutBE32(p, crc);
    p[4] = p[5] = p[6] = p[7] = 0;
    p[8] = 'I'; p[9] = 'E'; p[10] = 'N'; p[11] = 'D';
    p[12] = 0xae; p[13] = 0x42; p[14] = 0x60; p[15] = 0x82;
}

// ---- PNG body, group-wise: FOUR pixels of a row per index (what PngBodyKernel runs) ----------
// Same bytes as PngRawByte / PngRawOffset element by element, but a lane handles the 16 (RGBA) or
// 12 (RGB) filtered bytes of four adjacent pixels: dword loads, the Sub filter as one byte-wise
// subtraction per pixel, one run of (unaligned) dword stores unless a stored-block boundary cuts
// through the run, and the lane's share of the two Adler sums from byte sums / dot products.
TIMG_HD uint32_t GfxLoadU32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
TIMG_HD void GfxStoreU32(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
// a - b (mod 256) in each of the four bytes
TIMG_HD uint32_t SubBytes4(uint32_t a, uint32_t b) {
    return ((a | 0x80808080u) - (b & 0x7f7f7f7fu)) ^ ((a ^ ~b) & 0x80808080u);
}
TIMG_HD uint32_t SumBytes4(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sad_u8(v, 0u, 0u);
#else
    return (v & 255u) + ((v >> 8) & 255u) + ((v >> 16) & 255u) + (v >> 24);
#endif
}
TIMG_HD uint32_t Dot0123(uint32_t v) {  // 0 * byte0 + 1 * byte1 + 2 * byte2 + 3 * byte3
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(v, 0x03020100u, 0u, false);
#else
    return ((v >> 8) & 255u) + 2u * ((v >> 16) & 255u) + 3u * (v >> 24);
#endif
}
// group xg (pixels 4 xg .. 4 xg + 3) of row y.  Returns what the Adler sums need from it (the row's
// filter-type byte rides with group 0): a = sum of its bytes d, j0 = index of its first pixel byte,
// t = sum i * d_i over its pixel bytes (i from 0) minus 1 if it carries a filter byte -- then
//   A' += a,   B' += (raw_n - j0) * a - t     [B' = sum (raw_n - j) d_j]
struct PngGroupSums {
    uint32_t a, j0;
    int32_t t;
};
TIMG_HD PngGroupSums PngBodyGroup(const uint8_t *frame, size_t stride, const PngGeom &g, uint32_t y, uint32_t xg,
                                  uint8_t *png) {
    const uint32_t x0 = xg * 4u, npx = (uint32_t)g.w - x0 < 4u ? (uint32_t)g.w - x0 : 4u;
    const uint8_t *line = frame + (size_t)y * stride;
    uint32_t prev = x0 ? GfxLoadU32(line + 4u * (x0 - 1u)) : 0u;  // (the first pixel of a row is stored as it is)
    uint32_t cur[4] = {0u, 0u, 0u, 0u}, d[4] = {0u, 0u, 0u, 0u};
    if (npx == 4u) {
        __builtin_memcpy(cur, line + 4u * x0, 16);  // (one 16-byte load)
    } else {
        for (uint32_t k = 0; k < npx; ++k) cur[k] = GfxLoadU32(line + 4u * (x0 + k));
    }
    for (uint32_t k = 0; k < 4u; ++k) {
        if (k < npx) d[k] = SubBytes4(cur[k], prev);
        prev = cur[k];
    }
    uint32_t o[4], nbytes;
    if (g.bpp == 4) {
        o[0] = d[0], o[1] = d[1], o[2] = d[2], o[3] = d[3];
        nbytes = 4u * npx;
    } else {  // alpha dropped: 3 bytes per pixel, packed
        const uint32_t t0 = d[0] & 0xffffffu, t1 = d[1] & 0xffffffu, t2 = d[2] & 0xffffffu, t3 = d[3] & 0xffffffu;
        o[0] = t0 | (t1 << 24);
        o[1] = (t1 >> 8) | (t2 << 16);
        o[2] = (t2 >> 16) | (t3 << 8);
        o[3] = 0u;
        nbytes = 3u * npx;
    }
    PngGroupSums r;
    r.j0 = y * g.row + 1u + (uint32_t)g.bpp * x0;
    uint32_t a = 0, kd = 0;
    for (uint32_t q = 0; q < 4u; ++q) {  // (bytes past nbytes are zero)
        const uint32_t sq = SumBytes4(o[q]);
        a += sq;
        kd += Dot0123(o[q]) + 4u * q * sq;
    }
    r.a = a;
    r.t = (int32_t)kd;
    if (xg == 0) {  // filter type 1 in front of the row: value 1 at index j0 - 1
        png[PngRawOffset(r.j0 - 1u)] = 1;
        r.a += 1u;   // (raw_n - (j0 - 1)) * 1 = (raw_n - j0) * 1 + 1
        r.t -= 1;
    }
    if (r.j0 / kStoredBlock == (r.j0 + nbytes - 1u) / kStoredBlock) {
        uint8_t *dst = png + PngRawOffset(r.j0);
        if (nbytes == 16u) {
            __builtin_memcpy(dst, o, 16);
        } else {
            for (uint32_t q = 0; q < 4u; ++q)
                if (4u * q + 4u <= nbytes) GfxStoreU32(dst + 4u * q, o[q]);
            for (uint32_t i = nbytes & ~3u; i < nbytes; ++i) dst[i] = (uint8_t)(o[i >> 2] >> (8u * (i & 3u)));
        }
    } else {  // a block header lies inside the run
        for (uint32_t i = 0; i < nbytes; ++i) png[PngRawOffset(r.j0 + i)] = (uint8_t)(o[i >> 2] >> (8u * (i & 3u)));
    }
    return r;
}
TIMG_HD uint32_t PngBodyGroups(const PngGeom &g) { return (uint32_t)g.h * (((uint32_t)g.w + 3u) >> 2); }

// ---- base64 + framing: one group of three PNG bytes per index g ----------------------------
struct GfxFraming {
    int kind;               // GfxKind
    uint32_t header_len;    // bytes in front of the first base64 character
    uint32_t n_groups;      // ceil(png_n / 3)
    uint32_t n_kitty_chunks;
    uint32_t total;         // bytes of the whole frame
};

TIMG_HD GfxFraming MakeFraming(int kind, const PngGeom &g, uint32_t header_len) {
    GfxFraming f{};
    f.kind       = kind;
    f.header_len = header_len;
    f.n_groups   = (g.png_n + 2) / 3;
    f.n_kitty_chunks = (g.png_n + kKittyChunk - 1) / kKittyChunk;
    if (kind == kGfxKitty)
        f.total = header_len + 4 * f.n_groups + (f.n_kitty_chunks - 1) * kKittySepLen + 3;  // ESC \ LF
    else
        f.total = header_len + 4 * f.n_groups + 2;  // BEL LF
    return f;
}

TIMG_HD uint32_t Base64Quad(const uint8_t *png, uint32_t png_n, uint32_t grp) {  // four characters, first in the low byte
    const char *b64 = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    const uint32_t at = grp * 3, left = png_n - at;
    const uint32_t b0 = png[at], b1 = left > 1 ? png[at + 1] : 0, b2 = left > 2 ? png[at + 2] : 0;
    const uint32_t c0 = (uint8_t)b64[b0 >> 2], c1 = (uint8_t)b64[((b0 & 3) << 4) | (b1 >> 4)];
    const uint32_t c2 = left > 1 ? (uint8_t)b64[((b1 & 15) << 2) | (b2 >> 6)] : '=';
    const uint32_t c3 = left > 2 ? (uint8_t)b64[b2 & 63] : '=';
    return c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
}
// where group grp's four characters go
TIMG_HD uint32_t GfxGroupOffset(const GfxFraming &f, uint32_t grp) {
    const uint32_t groups_per_chunk = kKittyChunk / 3;
    const uint32_t sep = f.kind == kGfxKitty ? (grp / groups_per_chunk) * kKittySepLen : 0;
    return f.header_len + 4 * grp + sep;
}
// FOUR groups per index (what GfxFrameKernel runs): 12 PNG bytes as three dwords -> 16 characters
// as four (unaligned) dword stores; the alphabet by byte-wise arithmetic instead of a table:
// 0..25 -> +65 ('A'), 26..51 -> +71 ('a' - 26), 52..61 -> -4 ('0' - 52), 62 -> -19 ('+'), 63 -> -16 ('/')
TIMG_HD uint32_t Base64Chars4(uint32_t v) {  // four 6-bit values, one per byte
    const uint32_t k = 0x01010101u;
    const uint32_t ge26 = ((v + (128u - 26u) * k) >> 7) & k, ge52 = ((v + (128u - 52u) * k) >> 7) & k;
    const uint32_t ge62 = ((v + (128u - 62u) * k) >> 7) & k, ge63 = ((v + (128u - 63u) * k) >> 7) & k;
    return v + 65u * k + 6u * ge26 - 75u * ge52 - 15u * ge62 + 3u * ge63;
}
TIMG_HD uint32_t Base64FromTriple(uint32_t n) {  // n = b0 << 16 | b1 << 8 | b2 -> characters, first in the low byte
    return Base64Chars4((n >> 18) | ((n >> 4) & 0x3f00u) | ((n << 10) & 0x3f0000u) | ((n << 24) & 0x3f000000u));
}
static_assert((kKittyChunk / 3) % 4 == 0, "four groups never straddle a kitty chunk");
TIMG_HD void GfxFrameQuad(const uint8_t *png, const PngGeom &g, const GfxFraming &fr, uint32_t quad, uint8_t *out) {
    const uint32_t g0 = 4u * quad;
    if (g0 >= fr.n_groups) return;
    if (12u * (quad + 1u) <= g.png_n) {  // four complete groups
        const uint32_t d0 = GfxLoadU32(png + 12u * quad), d1 = GfxLoadU32(png + 12u * quad + 4u),
                       d2 = GfxLoadU32(png + 12u * quad + 8u);
        // bytes b0..b11 in memory order; triples (b0 b1 b2) (b3 b4 b5) (b6 b7 b8) (b9 b10 b11), big-endian each
        const uint32_t n0 = ((d0 & 0xffu) << 16) | (d0 & 0xff00u) | ((d0 >> 16) & 0xffu);
        const uint32_t n1 = ((d0 >> 24) << 16) | ((d1 & 0xffu) << 8) | ((d1 >> 8) & 0xffu);
        const uint32_t n2 = (d1 & 0xff0000u) | ((d1 >> 24) << 8) | (d2 & 0xffu);
        const uint32_t n3 = ((d2 & 0xff00u) << 8) | ((d2 >> 8) & 0xff00u) | (d2 >> 24);
        uint8_t *o = out + GfxGroupOffset(fr, g0);
        GfxStoreU32(o, Base64FromTriple(n0));
        GfxStoreU32(o + 4, Base64FromTriple(n1));
        GfxStoreU32(o + 8, Base64FromTriple(n2));
        GfxStoreU32(o + 12, Base64FromTriple(n3));
    } else {
        for (uint32_t grp = g0; grp < g0 + 4u && grp < fr.n_groups; ++grp)
            GfxStoreU32(out + GfxGroupOffset(fr, grp), Base64Quad(png, g.png_n, grp));
    }
}
// separator in front of kitty chunk c >= 1: ESC \ ESC _ G q = 2 , m = <more> ;
TIMG_HD uint32_t KittySeparatorOffset(const GfxFraming &f, uint32_t c) {
    return f.header_len + c * (4 * (kKittyChunk / 3) + kKittySepLen) - kKittySepLen;
}
TIMG_HD void KittySeparator(const GfxFraming &f, uint32_t c, uint8_t sep[13]) {
    const char *s = "\033\\\033_Gq=2,m=";
    for (int i = 0; i < 11; ++i) sep[i] = (uint8_t)s[i];
    sep[11] = c + 1 < f.n_kitty_chunks ? '1' : '0';
    sep[12] = ';';
}
TIMG_HD void GfxTrailer(const GfxFraming &f, uint8_t *frame_out) {
    uint8_t *p = frame_out + f.total - (f.kind == kGfxKitty ? 3 : 2);
    if (f.kind == kGfxKitty) {
        p[0] = 033;
        p[1] = '\\';
        p[2] = '\n';
    } else {
        p[0] = 007;
        p[1] = '\n';
    }
}

// (host) what stands in front of the base64 data: src/kitty-canvas.cc:180-186 (no tmux),
// src/iterm2-canvas.cc:63-65.  Returns its length (< kGfxHeaderCap).
inline uint32_t FormatGfxHeader(int kind, const PngGeom &g, uint32_t image_id, char *buf) {
    int n = 0;
    if (kind == kGfxKitty)
        n = snprintf(buf, kGfxHeaderCap, "\033_Ga=T,i=%u,q=2,f=100,m=%d;", image_id, g.png_n > kKittyChunk ? 1 : 0);
    else if (kind == kGfxIterm2)
        n = snprintf(buf, kGfxHeaderCap, "\033]1337;File=size=%d;width=%dpx;height=%dpx;inline=1:", (int)g.png_n, g.w,
                     g.h);
    return (uint32_t)n;
}

}  // namespace timg_amd

#endif  // TIMG_AMD_GFX_LAYOUT_H_
