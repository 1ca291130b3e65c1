/* oracle/png.c -- TEST INFRASTRUCTURE ONLY (see timg_oracle.h).
 *
 * CPU restatement of the reference's graphics-protocol path at --compress=0:
 *   png::Encode                      src/timg-png.cc:91-153 (sub filter, IHDR/IDAT/IEND)
 *   libdeflate_zlib_compress level 0 third-party, NOT in the reference tree (libdeflate 1.8
 *                                    here); level 0 = RFC 1950 header 78 01, RFC 1951 stored
 *                                    blocks of at most 65535 bytes, Adler-32 trailer -- a
 *                                    published format with exactly one valid byte stream once
 *                                    the block size is fixed, pinned against the real library
 *                                    through oracle/_ref (tests/test_png_oracle.py)
 *   EncodeBase64                     src/timg-base64.h:28-55
 *   KittyGraphicsCanvas::Send        src/kitty-canvas.cc:126-221 (the bytes of encode_fun,
 *                                    no tmux wrapping)
 *   ITerm2GraphicsCanvas::Send       src/iterm2-canvas.cc:40-75
 * Other compression levels are libdeflate's match finder: not restated.
 */
#include <stdio.h>
#include <string.h>

#include "timg_oracle.h"

/* ---- checksums ---------------------------------------------------------- */
uint32_t oracle_crc32(uint32_t crc, const uint8_t *p, size_t n) { /* zlib / PNG CRC, crc = 0 to start */
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) {
        crc ^= p[i];
        for (int k = 0; k < 8; ++k) crc = (crc & 1u) ? (crc >> 1) ^ 0xedb88320u : crc >> 1;
    }
    return ~crc;
}

uint32_t oracle_adler32(const uint8_t *p, size_t n) {
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; ++i) {
        a = (a + p[i]) % 65521u;
        b = (b + a) % 65521u;
    }
    return (b << 16) | a;
}

/* a(x) * b(x) mod P(x) over GF(2), bit-reflected like the CRC register */
uint32_t oracle_crc32_multmodp(uint32_t a, uint32_t b) {
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

/* x^(8 * n_bytes) mod P */
uint32_t oracle_crc32_xpow_bytes(uint64_t n_bytes) {
    uint32_t sq = 1u << 30; /* x^1 */
    uint32_t r  = 1u << 31; /* x^0 */
    uint64_t e  = n_bytes * 8;
    while (e) {
        if (e & 1) r = oracle_crc32_multmodp(sq, r);
        sq = oracle_crc32_multmodp(sq, sq);
        e >>= 1;
    }
    return r;
}

/* crc(A || B) from crc(A), crc(B), |B|: what a parallel CRC is built from */
uint32_t oracle_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    return oracle_crc32_multmodp(oracle_crc32_xpow_bytes(len_b), crc_a) ^ crc_b;
}

/* ---- PNG ------------------------------------------------------------------- */
static uint8_t *put32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
    return p + 4;
}

size_t oracle_png_bytes(int w, int h, int with_alpha) {
    const size_t raw    = (size_t)h * (1 + (size_t)w * (with_alpha ? 4 : 3));
    const size_t blocks = raw ? (raw + 65534) / 65535 : 1;
    return 8 + 25 + (12 + 2 + 5 * blocks + raw + 4) + 12;
}

long oracle_png_encode(const uint8_t *fb, int w, int h, int with_alpha, char *out_c, long cap) {
    const int bpp      = with_alpha ? 4 : 3;
    const size_t row   = 1 + (size_t)w * bpp;
    const size_t raw_n = (size_t)h * row;
    if ((size_t)cap < oracle_png_bytes(w, h, with_alpha)) return -1;
    uint8_t *out = (uint8_t *)out_c, *p = out;
    static const uint8_t sig[8] = {0x89, 0x50, 0x4e, 0x47, '\r', '\n', 0x1a, '\n'};
    memcpy(p, sig, 8);
    p += 8;
    /* IHDR */
    uint8_t *chunk = p;
    p = put32(p, 13);
    memcpy(p, "IHDR", 4);
    p += 4;
    p = put32(p, (uint32_t)w);
    p = put32(p, (uint32_t)h);
    *p++ = 8;
    *p++ = with_alpha ? 6 : 2;
    *p++ = 0; *p++ = 0; *p++ = 0;
    p = put32(p, oracle_crc32(0, chunk + 4, 17));
    /* IDAT: zlib stream of stored blocks over the sub-filtered rows */
    chunk = p;
    p += 4;
    memcpy(p, "IDAT", 4);
    p += 4;
    *p++ = 0x78;
    *p++ = 0x01;
    /* the filtered bytes are produced on the fly; adler over them */
    uint32_t a = 1, b = 0;
    size_t done = 0;
    for (int y = 0; y < h; ++y) {
        const uint8_t *line = fb + (size_t)y * w * 4;
        for (size_t i = 0; i < row; ++i, ++done) {
            if (done % 65535 == 0) {  /* a new stored block */
                const size_t len = raw_n - done < 65535 ? raw_n - done : 65535;
                *p++ = (len == raw_n - done) ? 1 : 0; /* BFINAL, BTYPE=00 */
                *p++ = (uint8_t)len; *p++ = (uint8_t)(len >> 8);
                *p++ = (uint8_t)~len; *p++ = (uint8_t)(~len >> 8);
            }
            uint8_t v;
            if (i == 0) {
                v = 1; /* filter type: Sub */
            } else {
                const size_t x = (i - 1) / bpp, c = (i - 1) % bpp;
                v = x == 0 ? line[c] : (uint8_t)(line[x * 4 + c] - line[(x - 1) * 4 + c]);
            }
            *p++ = v;
            a = (a + v) % 65521u;
            b = (b + a) % 65521u;
        }
    }
    if (raw_n == 0) { /* (not reachable with h >= 1; libdeflate emits one empty final block) */
        *p++ = 1; *p++ = 0; *p++ = 0; *p++ = 0xff; *p++ = 0xff;
    }
    p = put32(p, (b << 16) | a);
    put32(chunk, (uint32_t)(p - chunk - 8));
    p = put32(p, oracle_crc32(0, chunk + 4, (size_t)(p - chunk - 4)));
    /* IEND */
    chunk = p;
    p = put32(p, 0);
    memcpy(p, "IEND", 4);
    p += 4;
    p = put32(p, oracle_crc32(0, chunk + 4, 4));
    return (long)(p - out);
}

/* ---- base64, kitty, iTerm2 -------------------------------------------------------- */
long oracle_base64(const uint8_t *in, long n, char *out) {
    static const char b64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    char *o = out;
    for (; n >= 3; n -= 3, in += 3) {
        *o++ = b64[in[0] >> 2];
        *o++ = b64[((in[0] & 3) << 4) | (in[1] >> 4)];
        *o++ = b64[((in[1] & 15) << 2) | (in[2] >> 6)];
        *o++ = b64[in[2] & 63];
    }
    if (n > 0) {
        const uint8_t b1 = n > 1 ? in[1] : 0;
        *o++ = b64[in[0] >> 2];
        *o++ = b64[((in[0] & 3) << 4) | (b1 >> 4)];
        *o++ = n > 1 ? b64[(b1 & 15) << 2] : '=';
        *o++ = '=';
    }
    return (long)(o - out);
}

size_t oracle_kitty_max_bytes(int w, int h) {
    const size_t png = oracle_png_bytes(w, h, 1);
    return 64 + (png + 2) / 3 * 4 + (png / 3072 + 1) * 16 + 8;
}

/* scratch: at least oracle_png_bytes(w, h, with_alpha) bytes */
long oracle_kitty_encode(const uint8_t *fb, int w, int h, int with_alpha, uint32_t id, char *scratch,
                         char *out, long cap) {
    enum { kByteChunk = 4096 / 4 * 3 };
    long png = oracle_png_encode(fb, w, h, with_alpha, scratch, (long)oracle_png_bytes(w, h, with_alpha));
    if (png < 0 || (size_t)cap < oracle_kitty_max_bytes(w, h)) return -1;
    char *pos = out;
    pos += sprintf(pos, "\033_Ga=T,i=%u,q=2,f=100,m=%d;", id, png > kByteChunk);
    const uint8_t *data = (const uint8_t *)scratch;
    while (png) {
        const long n = png < kByteChunk ? png : kByteChunk;
        pos += oracle_base64(data, n, pos);
        data += n;
        png -= n;
        if (png) pos += sprintf(pos, "\033\\\033_Gq=2,m=%d;", png > kByteChunk);
    }
    *pos++ = '\033';
    *pos++ = '\\';
    *pos++ = '\n';
    return (long)(pos - out);
}

long oracle_iterm2_encode(const uint8_t *fb, int w, int h, int with_alpha, char *scratch, char *out,
                          long cap) {
    const long png = oracle_png_encode(fb, w, h, with_alpha, scratch, (long)oracle_png_bytes(w, h, with_alpha));
    if (png < 0 || (size_t)cap < oracle_kitty_max_bytes(w, h)) return -1;
    char *pos = out;
    pos += sprintf(pos, "\033]1337;File=size=%ld;width=%dpx;height=%dpx;inline=1:", png, w, h);
    pos += oracle_base64((const uint8_t *)scratch, png, pos);
    *pos++ = '\007';
    *pos++ = '\n';
    return (long)(pos - out);
}
