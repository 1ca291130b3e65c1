/* examples/abi_demo.c -- libtimg_hip.so from plain C: the drop-in boundary has no C++ in it.
 *
 *   gcc -std=c99 -Iinclude examples/abi_demo.c -Ltimg_amd -ltimg_hip -Wl,-rpath,'$ORIGIN/../timg_amd' -o examples/abi_demo
 *   examples/abi_demo > frame.six      (prints a 320x200 test picture as sixel)
 *
 * Mirrors what timg does per image: ImageScaler::Scale -> AlphaComposeBackground ->
 * SixelCanvas::Send / UnicodeBlockCanvas::Send, here with host buffers on both sides. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "timg_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != TIMG_HIP_OK) {                                                \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, timg_hip_last_error(ctx)); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char **argv) {
    const int sw = 1280, sh = 800, dw = 320, dh = 200;
    const int quarter = argc > 1 && strcmp(argv[1], "quarter") == 0;
    timg_hip_ctx *ctx = NULL;
    if (timg_hip_init(0, &ctx) != TIMG_HIP_OK) {
        fprintf(stderr, "no usable HIP device: %s\n", timg_hip_last_error(NULL));
        return 2;
    }
    /* a test picture with a translucent disc */
    uint8_t *src = (uint8_t *)malloc((size_t)sw * sh * 4);
    for (int y = 0; y < sh; ++y)
        for (int x = 0; x < sw; ++x) {
            uint8_t *p     = src + ((size_t)y * sw + x) * 4;
            const long dx  = x - sw / 2, dy = y - sh / 2;
            p[0]           = (uint8_t)(x * 255 / sw);
            p[1]           = (uint8_t)(y * 255 / sh);
            p[2]           = (uint8_t)((x ^ y) & 0xff);
            p[3]           = dx * dx + dy * dy < 300L * 300L ? 96 : 255;
        }
    timg_hip_scaler *scaler = NULL;
    CHECK(timg_hip_scaler_create(ctx, sw, sh, TIMG_HIP_FMT_RGBA, dw, dh, TIMG_HIP_FILTER_STB_DEFAULT, &scaler));
    timg_hip_blend blend;
    memset(&blend, 0, sizeof(blend));
    blend.enabled   = 1;
    blend.bg        = 0xff2e1e1eu; /* r=0x1e g=0x1e b=0x2e a=0xff */
    blend.pattern   = 0xff646464u;
    blend.pattern_w = blend.pattern_h = 8;
    uint8_t *fb     = (uint8_t *)malloc((size_t)dw * dh * 4);
    int transparent = 0;
    CHECK(timg_hip_scale_blend(ctx, scaler, src, 0, 0, 0, fb, 0, 0, 0, 1, &blend, &transparent, NULL));

    size_t cap = quarter ? timg_hip_block_max_bytes(dw, dh) : timg_hip_sixel_max_bytes(dw, dh);
    char *out  = (char *)malloc(cap);
    size_t len = 0;
    if (quarter)
        CHECK(timg_hip_block_encode(ctx, fb, dw, dh, 0, 0, 0, 1, TIMG_HIP_BLOCK_QUARTER, 0, out, cap, 0, &len, NULL));
    else
        CHECK(timg_hip_sixel_encode(ctx, fb, dw, dh, 0, 0, 0, 1, 0, &blend, out, cap, 0, &len, NULL));
    fwrite(out, 1, len, stdout);
    fprintf(stderr, "abi_demo: %dx%d -> %dx%d, transparent=%d, %zu bytes of %s\n", sw, sh, dw, dh, transparent, len,
            quarter ? "quarter blocks" : "sixel");
    free(out);
    free(fb);
    free(src);
    timg_hip_scaler_destroy(scaler);
    timg_hip_destroy(ctx);
    return 0;
}
