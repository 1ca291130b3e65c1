/* oracle/blend.c -- TEST INFRASTRUCTURE ONLY (see timg_oracle.h).
 *
 * Restates timg::LinearColor (src/framebuffer.h:138-174) and
 * Framebuffer::AlphaComposeBackground (src/framebuffer.cc:108-150).
 * All arithmetic is fp32 in the reference's operation order.
 */
#include <math.h>
#include <string.h>

#include "oracle_internal.h"
#include "timg_oracle.h"

lin_t lin_from_rgba(const uint8_t *p) { /* framebuffer.h:142-143: x^2.2 ~ x^2 */
    lin_t l;
    l.r = (float)(p[0] * p[0]);
    l.g = (float)(p[1] * p[1]);
    l.b = (float)(p[2] * p[2]);
    l.a = (float)p[3];
    return l;
}

static uint8_t gamma8(float v) { /* framebuffer.h:169-172 */
    const float vg = sqrtf(v);
    return (vg > 255) ? 255 : (uint8_t)vg;
}

void lin_repack(const lin_t *l, uint8_t *out) { /* framebuffer.h:150-152 */
    out[0] = gamma8(l->r);
    out[1] = gamma8(l->g);
    out[2] = gamma8(l->b);
    out[3] = (uint8_t)l->a;
}

static void alpha_blend(lin_t *c, const lin_t *bg) { /* framebuffer.h:155-161 */
    c->r = (c->r * c->a + bg->r * (0xff - c->a)) / 0xff;
    c->g = (c->g * c->a + bg->g * (0xff - c->a)) / 0xff;
    c->b = (c->b * c->a + bg->b * (0xff - c->a)) / 0xff;
    c->a = 0xff;
}

int oracle_alpha_compose(uint8_t *fb, int w, int h, int has_getter,
                         uint32_t bg, uint32_t pattern, int pw, int ph,
                         int start_row) {
    if (!has_getter) return 0; /* framebuffer.cc:111 */
    size_t total = (size_t)w * h;
    size_t pos   = (size_t)start_row * w;
    for (; pos < total; ++pos)
        if (fb[pos * 4 + 3] < 0xff) break; /* framebuffer.cc:113-116 */
    if (pos >= total) return 0;            /* getter never called */

    uint8_t bgc[4], pat[4];
    memcpy(bgc, &bg, 4);
    memcpy(pat, &pattern, 4);
    if (bgc[3] == 0x00) return 1; /* framebuffer.cc:120-121 */

    if (pat[3] == 0x00 || pattern == bg || pw <= 0 || ph <= 0) {
        const lin_t lbg = lin_from_rgba(bgc); /* framebuffer.cc:124-132 */
        for (; pos < total; ++pos) {
            uint8_t *px = fb + pos * 4;
            if (px[3] == 0xff) continue;
            lin_t c = lin_from_rgba(px);
            alpha_blend(&c, &lbg);
            lin_repack(&c, px);
        }
        return 1;
    }
    const lin_t choice[2] = {lin_from_rgba(bgc), lin_from_rgba(pat)};
    for (; pos < total; ++pos) { /* framebuffer.cc:135-149 */
        uint8_t *px = fb + pos * 4;
        if (px[3] == 0xff) continue;
        int x = (int)(pos % (size_t)w), y = (int)(pos / (size_t)w);
        const lin_t *b = &choice[((x / pw) + (y / ph)) % 2];
        lin_t c        = lin_from_rgba(px);
        alpha_blend(&c, b);
        lin_repack(&c, px);
    }
    return 1;
}

uint8_t oracle_as_256_term_color(uint32_t c) { /* framebuffer.h:37-52 */
    uint8_t p[4];
    memcpy(p, &c, 4);
    return term256(p);
}

uint8_t term256(const uint8_t *p) {
    uint8_t r = p[0], g = p[1], b = p[2];
    if (r == g && g == b) return (uint8_t)(232 + (r * 23 / 255));
    uint8_t v[3] = {r, g, b}, q[3];
    for (int i = 0; i < 3; i++) {
        uint8_t x = v[i];
        q[i] = x < 0x5f / 2            ? 0
               : x < (0x5f + 0x87) / 2 ? 1
               : x < (0x87 + 0xaf) / 2 ? 2
               : x < (0xaf + 0xd7) / 2 ? 3
               : x < (0xd7 + 0xff) / 2 ? 4
                                       : 5;
    }
    return (uint8_t)(16 + 36 * q[0] + 6 * q[1] + q[2]);
}
