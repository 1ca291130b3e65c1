/* oracle/autocrop.c -- TEST INFRASTRUCTURE ONLY (see timg_oracle.h).
 *
 * PARITY UNPINNED.  --auto-crop is GraphicsMagick's Image::trim() in the reference
 * (src/graphics-magick-source.cc:231-241: crop_border first, :232-237, then trim, :238-240, both
 * before the image is scaled); GraphicsMagick is neither vendored nor installed.  This restates
 * the published algorithm of trim() at fuzz 0 (magick/analyze.c GetImageBoundingBox): the left
 * and the top edge are measured against the top-left corner pixel, the right edge against the
 * top-right corner, the bottom edge against the bottom-left corner -- a pixel that differs from
 * the corner colour pushes that edge outwards. */
#include <string.h>

#include "timg_oracle.h"

void oracle_autocrop_bbox(const uint8_t *rgba, int w, int h, int stride, int crop_border, int xywh[4]) {
    xywh[0] = xywh[1] = xywh[2] = xywh[3] = 0;
    const int c = crop_border;
    if (c < 0 || 2 * c >= w || 2 * c >= h) return;
    const int x0 = c, y0 = c, x1 = w - c, y1 = h - c; /* src/graphics-magick-source.cc:232-237 */
    uint32_t tl, tr, bl;
    memcpy(&tl, rgba + (size_t)y0 * stride + (size_t)x0 * 4, 4);
    memcpy(&tr, rgba + (size_t)y0 * stride + (size_t)(x1 - 1) * 4, 4);
    memcpy(&bl, rgba + (size_t)(y1 - 1) * stride + (size_t)x0 * 4, 4);
    int minx = x1, miny = y1, maxx = -1, maxy = -1;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            uint32_t p;
            memcpy(&p, rgba + (size_t)y * stride + (size_t)x * 4, 4);
            if (p != tl && x < minx) minx = x;
            if (p != tr && x > maxx) maxx = x;
            if (p != tl && y < miny) miny = y;
            if (p != bl && y > maxy) maxy = y;
        }
    if (maxx < minx || maxy < miny) return; /* nothing but border */
    xywh[0] = minx;
    xywh[1] = miny;
    xywh[2] = maxx - minx + 1;
    xywh[3] = maxy - miny + 1;
}
