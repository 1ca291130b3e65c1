/* oracle/resample.c -- TEST INFRASTRUCTURE ONLY (see timg_oracle.h).
 *
 * CPU restatement of what timg::ImageScaler::Scale computes with the STB
 * back-end (src/image-scaler.cc:75-98): stb_image_resize2 v2.12 (vendored at
 * third_party/stb/stb_image_resize2.h), RGBA/BGRA uint8 -> RGBA uint8, edge
 * CLAMP, default filters (Mitchell down, trapezoid "box" up, point at 1:1),
 * non-premultiplied "fancy" 7-channel alpha weighting.
 *
 * This is a restatement, not a copy: the structure is ours (whole-image
 * passes instead of ring buffers, no SIMD, no splits), but every fp32
 * operation whose rounding reaches the output happens in the same order as in
 * the reference's SSE2 build (which is bit-identical to its scalar build,
 * stb_image_resize2.h:191-211).  Citations are stb_image_resize2.h lines.
 *
 * Pinned by tests/test_oracle_vs_ref.py against oracle/_ref (the real
 * reference compiled from /root/reference) and by tests/golden/.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "timg_oracle.h"

/* stb_image_resize2.h:1104 */
#define SMALL_FLOAT                                                          \
    ((float)1 / (1 << 20) / (1 << 20) / (1 << 20) / (1 << 20) / (1 << 20) / \
     (1 << 20))

enum { F_BOX = 1, F_TRIANGLE = 2, F_MITCHELL = 5, F_POINT = 6 };

typedef struct {
    int n0, n1;
} contrib_t;

typedef struct {
    /* scale_info (stb:7551-7598) */
    float scale, inv_scale, pixel_shift;
    int is_rational;
    uint32_t numer, denom;
    int in_size, out_size;
    /* sampler (stb:6496-6576) */
    int filter;
    int is_gather; /* 1 upsample gather, 2 downsample gather, 0 scatter */
    int pixel_width, pixel_margin;
    int coeff_width;
    int ncontrib;          /* number of gather contributors == out_size */
    contrib_t *contrib;    /* per OUTPUT pixel, even for scatter */
    float *coeff;          /* ncontrib * coeff_width */
    int lowest, highest, widest;
} sampler_t;

/* ---- filter kernels (stb:2845-2959) ------------------------------------ */
static float kernel_eval(int filter, float x, float s) {
    switch (filter) {
    case F_BOX: { /* trapezoid, stb:2845-2864 */
        float halfscale = s / 2;
        float t         = 0.5f + halfscale;
        if (x < 0.0f) x = -x;
        if (x >= t) return 0.0f;
        float r = 0.5f - halfscale;
        if (x <= r) return 1.0f;
        return (t - x) / s;
    }
    case F_TRIANGLE: /* stb:2872-2883 */
        if (x < 0.0f) x = -x;
        if (x <= 1.0f) return 1.0f - x;
        return 0.0f;
    case F_MITCHELL: /* stb:2924-2937 */
        if (x < 0.0f) x = -x;
        if (x < 1.0f) return (16.0f + x * x * (21.0f * x - 36.0f)) / 18.0f;
        if (x < 2.0f)
            return (32.0f + x * (-60.0f + x * (36.0f - 7.0f * x))) / 18.0f;
        return 0.0f;
    default: /* point, stb:2885-2892 */
        return 1.0f;
    }
}

static float support_eval(int filter, float s) {
    switch (filter) {
    case F_BOX: return 0.5f + s / 2.0f; /* stb:2866-2870 */
    case F_TRIANGLE: return 1.0f;
    case F_MITCHELL: return 2.0f;
    default: return 0.5f;
    }
}

/* ---- scale_info -------------------------------------------------------- */
/* stb:7474-7549, continued-fraction approximation of the scale */
static int to_rational(double f, uint32_t limit, uint32_t *numer,
                       uint32_t *denom, int limit_denom) {
    double err;
    uint64_t top, bot;
    uint64_t numer_last = 0, denom_last = 1, numer_est = 1, denom_est = 0;
    top = (uint64_t)(f * (double)(1 << 25));
    bot = 1 << 25;
    for (;;) {
        uint64_t est, temp;
        if ((limit_denom ? denom_est : numer_est) >= limit) break;
        if (denom_est) {
            err = ((double)numer_est / (double)denom_est) - f;
            if (err < 0.0) err = -err;
            if (err < (1.0 / (double)(1 << 24))) {
                *numer = (uint32_t)numer_est;
                *denom = (uint32_t)denom_est;
                return 1;
            }
        }
        if (bot == 0) break;
        est  = top / bot;
        temp = top % bot;
        top  = bot;
        bot  = temp;
        temp       = est * denom_est + denom_last;
        denom_last = denom_est;
        denom_est  = temp;
        temp       = est * numer_est + numer_last;
        numer_last = numer_est;
        numer_est  = temp;
    }
    if (limit_denom) {
        numer_est = (uint64_t)(f * (double)limit + 0.5);
        denom_est = limit;
    } else {
        numer_est = limit;
        denom_est = (uint64_t)(((double)limit / f) + 0.5);
    }
    *numer = (uint32_t)numer_est;
    *denom = (uint32_t)denom_est;
    err    = denom_est ? (((double)(uint32_t)numer_est /
                        (double)(uint32_t)denom_est) -
                       f)
                       : 1.0;
    if (err < 0.0) err = -err;
    return (err < (1.0 / (double)(1 << 24))) ? 1 : 0;
}

/* stb:7551-7598 with the full-image regions timg uses (s0=0,s1=1, no clip) */
static void scale_info_init(sampler_t *s, int out_full, int in_full) {
    double output_range = (double)out_full, input_range = (double)in_full;
    double output_s = ((double)out_full) / output_range;
    double ratio    = output_s / 1.0;
    double scale    = (output_range / input_range) * ratio;
    s->scale        = (float)scale;
    s->inv_scale    = (float)(1.0 / scale);
    s->pixel_shift  = (float)(0.0 * ratio * output_range);
    s->is_rational =
        to_rational(scale, (scale <= 1.0) ? (uint32_t)out_full : (uint32_t)in_full,
                    &s->numer, &s->denom, scale >= 1.0);
    s->in_size  = in_full;
    s->out_size = out_full;
}

/* stb:2962-2991 */
static int filter_pixel_width(int filter, float scale) {
    if (scale >= (1.0f - SMALL_FLOAT))
        return (int)ceilf(support_eval(filter, 1.0f / scale) * 2.0f);
    return (int)ceilf(support_eval(filter, scale) * 2.0f / scale);
}

/* stb:6496-6576 */
static void sampler_init(sampler_t *s, int filter, int always_gather) {
    if (filter == 0) {
        filter = F_MITCHELL;
        if (s->scale >= (1.0f - SMALL_FLOAT)) {
            if ((s->scale <= (1.0f + SMALL_FLOAT)) &&
                (ceilf(s->pixel_shift) == s->pixel_shift))
                filter = F_POINT;
            else
                filter = F_BOX; /* STBIR_DEFAULT_FILTER_UPSAMPLE override,
                                   src/image-scaler.cc:32 */
        }
    }
    s->filter      = filter;
    s->pixel_width = filter_pixel_width(filter, s->scale);
    s->is_gather   = 0;
    if (s->scale >= (1.0f - SMALL_FLOAT))
        s->is_gather = 1;
    else if (always_gather || s->pixel_width <= 32) /* stb:1201 */
        s->is_gather = 2;
    s->pixel_margin = s->pixel_width / 2;
    /* Coefficients are always generated in gather form; for a scatter sampler
     * the reference generates "gather_prescatter" coefficients of width
     * filter_pixel_width and pivots them (stb:3907-3925, 6569-6575). */
    switch (s->is_gather) {
    case 1:
        s->coeff_width =
            (int)ceilf(support_eval(filter, 1.0f / s->scale) * 2.0f);
        break;
    case 2:
        s->coeff_width =
            (int)ceilf(support_eval(filter, s->scale) * 2.0f / s->scale);
        break;
    default: s->coeff_width = s->pixel_width; break;
    }
    s->ncontrib = s->out_size;
    s->contrib  = (contrib_t *)calloc((size_t)s->ncontrib, sizeof(contrib_t));
    s->coeff    = (float *)calloc((size_t)s->ncontrib * s->coeff_width + 1,
                                  sizeof(float));
}

static void sampler_free(sampler_t *s) {
    free(s->contrib);
    free(s->coeff);
}

/* ---- coefficient generation ------------------------------------------- */
/* stb:3242-3265 (CLAMP edge) */
static void in_pixel_range(int *first_pixel, int *last_pixel,
                           float out_pixel_center, float out_filter_radius,
                           float inv_scale, float out_shift) {
    float lo  = out_pixel_center - out_filter_radius;
    float hi  = out_pixel_center + out_filter_radius;
    float ilo = (lo + out_shift) * inv_scale;
    float ihi = (hi + out_shift) * inv_scale;
    int first = (int)(floorf(ilo + 0.5f));
    int last  = (int)(floorf(ihi - 0.5f));
    if (last < first) last = first;
    *first_pixel = first;
    *last_pixel  = last;
}

/* stb:3365-3380 */
static void out_pixel_range(int *first_pixel, int *last_pixel,
                            float in_pixel_center, float in_pixels_radius,
                            float scale, float out_shift, int out_size) {
    float ilo   = in_pixel_center - in_pixels_radius;
    float ihi   = in_pixel_center + in_pixels_radius;
    float olo   = ilo * scale - out_shift;
    float ohi   = ihi * scale - out_shift;
    int first   = (int)(floorf(olo + 0.5f));
    int last    = (int)(floorf(ohi - 0.5f));
    if (first < 0) first = 0;
    if (last >= out_size) last = out_size - 1;
    *first_pixel = first;
    *last_pixel  = last;
}

/* stb:3267-3327 */
static void coeffs_gather_upsample(sampler_t *s) {
    float out_filter_radius = support_eval(s->filter, s->inv_scale) * s->scale;
    int polyphase = s->is_rational && ((int)s->numer < s->ncontrib);
    int end       = polyphase ? (int)s->numer : s->ncontrib;
    for (int n = 0; n < end; n++) {
        float *cg              = s->coeff + (size_t)n * s->coeff_width;
        float out_pixel_center = (float)n + 0.5f;
        float in_center_of_out =
            (out_pixel_center + s->pixel_shift) * s->inv_scale;
        int first, last;
        in_pixel_range(&first, &last, out_pixel_center, out_filter_radius,
                       s->inv_scale, s->pixel_shift);
        if ((last - first + 1) > s->coeff_width)
            last = first + s->coeff_width - 1;
        int last_non_zero = -1;
        for (int i = 0; i <= last - first; i++) {
            float in_pixel_center = (float)(i + first) + 0.5f;
            float c = kernel_eval(s->filter, in_center_of_out - in_pixel_center,
                                  s->inv_scale);
            if ((c < SMALL_FLOAT) && (c > -SMALL_FLOAT)) {
                if (i == 0) { /* eat leading zero contributors */
                    ++first;
                    i--;
                    continue;
                }
                c = 0;
            } else
                last_non_zero = i;
            cg[i] = c;
        }
        last            = last_non_zero + first;
        s->contrib[n].n0 = first;
        s->contrib[n].n1 = last;
    }
}

/* stb:3382-3458; start..end are input pixels incl. the filter margin */
static void coeffs_gather_downsample(sampler_t *s) {
    float in_pixels_radius = support_eval(s->filter, s->scale) * s->inv_scale;
    int start = -s->pixel_margin, end = s->in_size + s->pixel_margin;
    int first_out_inited = -1;
    int polyphase = s->is_rational && ((int)s->numer < s->out_size);
    for (int in_pixel = start; in_pixel < end; in_pixel++) {
        float in_pixel_center  = (float)in_pixel + 0.5f;
        float out_center_of_in = in_pixel_center * s->scale - s->pixel_shift;
        int ofirst, olast;
        out_pixel_range(&ofirst, &olast, in_pixel_center, in_pixels_radius,
                        s->scale, s->pixel_shift, s->out_size);
        if (ofirst > olast) continue;
        if (polyphase) {
            if (ofirst == (int)s->numer) break;
            if (olast >= (int)s->numer) olast = (int)s->numer - 1;
        }
        for (int i = 0; i <= olast - ofirst; i++) {
            float out_pixel_center = (float)(i + ofirst) + 0.5f;
            float x                = out_pixel_center - out_center_of_in;
            float c = kernel_eval(s->filter, x, s->scale) * s->scale;
            if ((c < SMALL_FLOAT) && (c > -SMALL_FLOAT)) c = 0.0f;
            int out        = i + ofirst;
            float *coeffs  = s->coeff + (size_t)out * s->coeff_width;
            contrib_t *cb  = s->contrib + out;
            if (out > first_out_inited) {
                first_out_inited = out;
                cb->n0 = cb->n1 = in_pixel;
                coeffs[0]       = c;
            } else {
                if (coeffs[0] == 0.0f) cb->n0 = in_pixel; /* zap leading 0 */
                cb->n1 = in_pixel;
                if ((in_pixel - cb->n0) < s->coeff_width)
                    coeffs[in_pixel - cb->n0] = c;
            }
        }
    }
}

/* stb:3329-3363 (faithful, including its loop quirks) */
static void insert_coeff(contrib_t *cb, float *coeffs, int new_pixel,
                         float new_coeff, int max_width) {
    if (new_pixel <= cb->n1) {
        if (new_pixel < cb->n0) {
            if ((cb->n1 - new_pixel + 1) <= max_width) {
                int j, o = cb->n0 - new_pixel;
                for (j = cb->n1 - cb->n0; j <= 0; j--)
                    coeffs[j + o] = coeffs[j];
                for (j = 1; j < o; j--) coeffs[j] = coeffs[0];
                coeffs[0] = new_coeff;
                cb->n0    = new_pixel;
            }
        } else
            coeffs[new_pixel - cb->n0] += new_coeff;
    } else {
        if ((new_pixel - cb->n0 + 1) <= max_width) {
            int j, e = new_pixel - cb->n0;
            for (j = (cb->n1 - cb->n0) + 1; j < e; j++) coeffs[j] = 0;
            coeffs[e] = new_coeff;
            cb->n1    = new_pixel;
        }
    }
}

static int clamp_idx(int n, int max) { /* stb:3008-3017 */
    if (n < 0) return 0;
    if (n >= max) return max - 1;
    return n;
}

/* stb:3466-3635, CLAMP edge.  Renormalisation in double (stb:3460-3464). */
static void cleanup_gathered(sampler_t *s) {
    int input_size = s->in_size, input_last_n1 = input_size - 1;
    int lowest = 0x7fffffff, highest = -0x7fffffff, widest = -1;
    int cw        = s->coeff_width;
    int polyphase = s->is_rational && ((int)s->numer < s->ncontrib);
    int end       = polyphase ? (int)s->numer : s->ncontrib;
    for (int n = 0; n < end; n++) {
        float *coeffs  = s->coeff + (size_t)n * cw;
        contrib_t *cb  = s->contrib + n;
        double total   = 0;
        int e          = cb->n1 - cb->n0;
        for (int i = 0; i <= e; i++) total += (double)coeffs[i];
        if ((total < SMALL_FLOAT) && (total > -SMALL_FLOAT)) {
            cb->n1    = cb->n0;
            coeffs[0] = 0.0f;
        } else if ((total < (1.0f - SMALL_FLOAT)) ||
                   (total > (1.0f + SMALL_FLOAT))) {
            double filter_scale = ((double)1.0) / total;
            for (int i = 0; i <= e; i++)
                coeffs[i] = (float)(coeffs[i] * filter_scale);
        }
    }
    if (polyphase) { /* stb:3523-3537 */
        for (int n = (int)s->numer; n < s->ncontrib; n++) {
            s->contrib[n].n0 = s->contrib[n - s->numer].n0 + (int)s->denom;
            s->contrib[n].n1 = s->contrib[n - s->numer].n1 + (int)s->denom;
        }
        /* the reference's overlapping *forward* copy (stb:2658-2686)
         * replicates the first `numer` rows periodically */
        for (int n = (int)s->numer; n < s->ncontrib; n++)
            memcpy(s->coeff + (size_t)n * cw,
                   s->coeff + (size_t)(n - (int)s->numer) * cw,
                   (size_t)cw * sizeof(float));
    }
    for (int n = 0; n < s->ncontrib; n++) {
        float *coeffs = s->coeff + (size_t)n * cw;
        contrib_t *cb = s->contrib + n;
        /* fold out-of-range taps onto the clamped pixel, right side first */
        if (cb->n1 > input_last_n1) {
            int start = cb->n0, endi = cb->n1;
            cb->n1 = input_last_n1;
            for (int i = input_size; i <= endi; i++)
                insert_coeff(cb, coeffs, clamp_idx(i, input_size),
                             coeffs[i - start], cw);
        }
        if (cb->n0 < 0) {
            int save_n0;
            float save_n0_coeff;
            float *c = coeffs - (cb->n0 + 1);
            for (int i = -1; i > cb->n0; i--)
                insert_coeff(cb, coeffs, clamp_idx(i, input_size), *c--, cw);
            save_n0       = cb->n0;
            save_n0_coeff = c[0];
            cb->n0        = 0;
            for (int i = 0; i <= cb->n1; i++) coeffs[i] = coeffs[i - save_n0];
            insert_coeff(cb, coeffs, clamp_idx(save_n0, input_size),
                         save_n0_coeff, cw);
        }
        if (cb->n0 <= cb->n1) {
            int diff = cb->n1 - cb->n0 + 1;
            while (diff && (coeffs[diff - 1] == 0.0f)) --diff;
            cb->n1 = cb->n0 + diff - 1;
            if (cb->n0 <= cb->n1) {
                if (cb->n0 < lowest) lowest = cb->n0;
                if (cb->n1 > highest) highest = cb->n1;
                if (diff > widest) widest = diff;
            }
            for (int i = diff; i < cw; i++) coeffs[i] = 0.0f;
        }
    }
    s->lowest  = lowest;
    s->highest = highest;
    s->widest  = widest;
}

/* stb:3874-3898 + the gather half of 3900-3935 */
static void calculate_filters(sampler_t *s) {
    if (s->is_gather == 1)
        coeffs_gather_upsample(s);
    else
        coeffs_gather_downsample(s);
    cleanup_gathered(s);
}

/* stb:6578-6668 (non-WRAP edges): conservative decoded range of a scanline */
static void conservative_extents(const sampler_t *s, contrib_t *range) {
    int first, last;
    if (s->is_gather == 1) {
        float radius = support_eval(s->filter, s->inv_scale) * s->scale;
        in_pixel_range(&first, &last, 0.5f, radius, s->inv_scale,
                       s->pixel_shift);
        range->n0 = first;
        in_pixel_range(&first, &last, ((float)(s->out_size - 1)) + 0.5f, radius,
                       s->inv_scale, s->pixel_shift);
        range->n1 = last;
    } else {
        float in_radius = support_eval(s->filter, s->scale) * s->inv_scale;
        int n, input_end, of, ol;
        in_pixel_range(&first, &last, 0, 0, s->inv_scale, s->pixel_shift);
        range->n0 = first;
        in_pixel_range(&first, &last, (float)s->out_size, 0, s->inv_scale,
                       s->pixel_shift);
        range->n1 = last;
        n         = range->n0 + 1;
        input_end = -s->pixel_margin;
        while (n >= input_end) {
            out_pixel_range(&of, &ol, ((float)n) + 0.5f, in_radius, s->scale,
                            s->pixel_shift, s->out_size);
            if (of > ol) break;
            if ((of < s->out_size) || (ol >= 0)) range->n0 = n;
            --n;
        }
        n         = range->n1 - 1;
        input_end = n + 1 + s->pixel_margin;
        while (n <= input_end) {
            out_pixel_range(&of, &ol, ((float)n) + 0.5f, in_radius, s->scale,
                            s->pixel_shift, s->out_size);
            if (of > ol) break;
            if ((of < s->out_size) || (ol >= 0)) range->n1 = n;
            ++n;
        }
    }
    if (range->n0 < 0) range->n0 = 0;
    if (range->n1 >= s->in_size) range->n1 = s->in_size - 1;
}

/* stb:3639-3870.  Packs rows to `widest` floats and, at the right edge, moves
 * n0 back (zero-filling) so that a fixed-width unrolled loop never reads past
 * the decoded scanline.  The move matters for parity: it changes which taps
 * fall in the even / odd accumulation chain of the horizontal gather. */
static void pack_coefficients(sampler_t *s, int row1) {
    int widest = s->widest, cw = s->coeff_width, row_end = row1 + 1;
    if (cw != widest) {
        for (int n = 0; n < s->ncontrib; n++)
            memmove(s->coeff + (size_t)n * widest, s->coeff + (size_t)n * cw,
                    (size_t)widest * sizeof(float));
    }
    s->coeff_width = widest;
    contrib_t *cb  = s->contrib + s->ncontrib - 1;
    float *coeffs  = s->coeff + (size_t)widest * (s->ncontrib - 1);
    while ((cb >= s->contrib) && ((cb->n0 + widest * 2) >= row_end)) {
        if ((cb->n0 + widest) > row_end) {
            int stop_range = widest;
            if (widest > 12) {
                int mod    = widest & 3;
                stop_range = (((cb->n1 - cb->n0 + 1) - mod + 3) & ~3) + mod;
                if (stop_range < (8 + mod)) stop_range = 8 + mod;
            }
            if ((cb->n0 + stop_range) > row_end) {
                int new_n0     = row_end - stop_range;
                int num        = cb->n1 - cb->n0 + 1;
                int backup     = cb->n0 - new_n0;
                float *from_co = coeffs + num - 1;
                float *to_co   = from_co + backup;
                while (num) {
                    *to_co-- = *from_co--;
                    --num;
                }
                while (to_co >= coeffs) *to_co-- = 0;
                cb->n0 = new_n0;
            }
        }
        --cb;
        coeffs -= widest;
    }
}

/* stb:6859-6906 with the 7-channel weights row (stb:6818-6827) */
static int should_do_vertical_first(int hpw, float hscale, int hout, int vpw,
                                    float vscale, int vout, int is_gather) {
    static const float w7[8][4] = {
        {0.00000f, 0.59375f, 0.00000f, 0.96875f},
        {0.06250f, 0.81250f, 0.06250f, 0.59375f},
        {0.75000f, 0.43750f, 0.12500f, 0.96875f},
        {0.87500f, 0.06250f, 0.18750f, 0.43750f},
        {1.00000f, 1.00000f, 1.00000f, 1.00000f},
        {0.15625f, 0.12500f, 1.00000f, 1.00000f},
        {0.06250f, 0.12500f, 0.00000f, 1.00000f},
        {0.00000f, 1.00000f, 0.03125f, 0.34375f},
    };
    int cls;
    if ((vout <= 4) || (hout <= 4))
        cls = (vout < hout) ? 6 : 7;
    else if (vscale <= 1.0f)
        cls = is_gather ? 1 : 0;
    else if (vscale <= 2.0f)
        cls = 2;
    else if (vscale <= 3.0f)
        cls = 3;
    else if (vscale <= 4.0f)
        cls = 5;
    else
        cls = 6;
    const float *w = w7[cls];
    double h_cost  = (float)hpw * w[0] + hscale * (float)vpw * w[1];
    double v_cost  = (float)vpw * w[2] + vscale * (float)hpw * w[3];
    return (v_cost <= h_cost) ? 1 : 0;
}

/* ---- the plan ----------------------------------------------------------- */
typedef struct {
    sampler_t h, v;
    contrib_t conservative;
    int vertical_first;
    int both_point;
    /* scatter view of the vertical sampler (stb:3937-4003): per input row
     * y+margin, the output rows it feeds and the pivoted coefficients */
    contrib_t *v_scatter;
    float *v_scatter_coeff;
    int v_scatter_width;
} plan_t;

/* stb:3937-4003: pivot gather coefficients into per-input-row scatter lists */
static void pivot_to_scatter(plan_t *p) {
    sampler_t *s = &p->v;
    int margin   = s->pixel_margin;
    int nrows    = s->in_size + margin * 2;
    /* is_gather==0 coefficient width: ceil(support(scale)*2), stb:2984-2985 */
    int sw = (int)ceilf(support_eval(s->filter, s->scale) * 2.0f);
    p->v_scatter_width = sw;
    p->v_scatter       = (contrib_t *)calloc((size_t)nrows, sizeof(contrib_t));
    p->v_scatter_coeff = (float *)calloc((size_t)nrows * sw + 1, sizeof(float));
    int highest_set = (-margin) - 1;
    for (int n = 0; n < s->ncontrib; n++) {
        int gn0 = s->contrib[n].n0, gn1 = s->contrib[n].n1;
        const float *g = s->coeff + (size_t)n * s->coeff_width;
        for (int k = gn0; k <= gn1; k++) {
            float gc       = *g++;
            contrib_t *sc  = p->v_scatter + (k + margin);
            float *scoeff  = p->v_scatter_coeff + (size_t)(k + margin) * sw;
            if ((gc >= SMALL_FLOAT) || (gc <= -SMALL_FLOAT)) {
                if ((k > highest_set) || (sc->n0 > sc->n1)) {
                    for (int c = highest_set + margin + 1; c < k + margin; c++) {
                        p->v_scatter[c].n0 = 0;
                        p->v_scatter[c].n1 = -1;
                    }
                    sc->n0 = sc->n1 = n;
                    scoeff[0]       = gc;
                    highest_set     = k;
                } else
                    insert_coeff(sc, scoeff, n, gc, sw);
            }
        }
    }
    for (int c = highest_set + margin + 1; c < nrows; c++) {
        p->v_scatter[c].n0 = 0;
        p->v_scatter[c].n1 = -1;
    }
}

static int plan_build(plan_t *p, int sw, int sh, int dw, int dh, int filter) {
    memset(p, 0, sizeof(*p));
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return -1;
    scale_info_init(&p->h, dw, sw);
    scale_info_init(&p->v, dh, sh);
    sampler_init(&p->h, filter, 1);
    conservative_extents(&p->h, &p->conservative);
    sampler_init(&p->v, filter, 0);
    p->both_point = (p->h.filter == F_POINT) && (p->v.filter == F_POINT);
    p->vertical_first = should_do_vertical_first(
        p->h.pixel_width, p->h.scale, p->h.out_size, p->v.pixel_width,
        p->v.scale, p->v.out_size, p->v.is_gather);
    calculate_filters(&p->h);
    /* identical samplers: the vertical one is a copy of the *packed*
     * horizontal one (stb:7166-7183, 7218-7221) */
    int copy_horizontal = 0;
    if (p->h.filter == p->v.filter && p->h.out_size == p->v.out_size) {
        float ds = p->h.scale - p->v.scale;
        float dp = p->h.pixel_shift - p->v.pixel_shift;
        if (ds < 0.0f) ds = -ds;
        if (dp < 0.0f) dp = -dp;
        if (ds <= SMALL_FLOAT && dp <= SMALL_FLOAT &&
            p->h.is_gather == p->v.is_gather)
            copy_horizontal = 1;
    }
    pack_coefficients(&p->h, p->conservative.n1);
    if (copy_horizontal) {
        sampler_t *v = &p->v, *h = &p->h;
        free(v->contrib);
        free(v->coeff);
        *v         = *h;
        v->contrib = (contrib_t *)malloc((size_t)h->ncontrib * sizeof(contrib_t));
        v->coeff   = (float *)malloc(((size_t)h->ncontrib * h->coeff_width + 1) *
                                     sizeof(float));
        memcpy(v->contrib, h->contrib, (size_t)h->ncontrib * sizeof(contrib_t));
        memcpy(v->coeff, h->coeff,
               (size_t)h->ncontrib * h->coeff_width * sizeof(float));
    } else {
        calculate_filters(&p->v);
    }
    if (p->v.is_gather == 0) pivot_to_scatter(p);
    return 0;
}

static void plan_free(plan_t *p) {
    sampler_free(&p->h);
    sampler_free(&p->v);
    free(p->v_scatter);
    free(p->v_scatter_coeff);
}

/* ---- per-pixel arithmetic ---------------------------------------------- */
/* decode (stb:8300-8321) + fancy alpha weight (stb:4081-4175):
 * R G B A R*A G*A B*A, channels as float in [0,1] */
static void decode_row(const uint8_t *row, int in_fmt, int x0, int x1,
                       float *out /* (x1-x0+1)*7 */) {
    const float inv255 = 1.0f / 255.0f;
    for (int x = x0; x <= x1; x++) {
        const uint8_t *px = row + (size_t)x * 4;
        float r = ((float)px[in_fmt ? 2 : 0]) * inv255;
        float g = ((float)px[1]) * inv255;
        float b = ((float)px[in_fmt ? 0 : 2]) * inv255;
        float a = ((float)px[3]) * inv255;
        float *o = out + (size_t)(x - x0) * 7;
        o[0] = r;
        o[1] = g;
        o[2] = b;
        o[3] = a;
        o[4] = r * a;
        o[5] = g * a;
        o[6] = b * a;
    }
}

/* One scanline of horizontal gather (stb:5722-5793 SSE2 == 5842-6009 scalar):
 * taps alternate between two accumulation chains (even/odd position in the
 * packed row) which are added at the end; rows of <=3 taps use one chain
 * (stb:5621-5660).  `in` is indexed by absolute input pixel - base. */
static void hgather_row(const plan_t *p, const float *in, int base,
                        float *out /* out_w*7 */) {
    const sampler_t *s = &p->h;
    int widest         = s->coeff_width;
    for (int x = 0; x < s->out_size; x++) {
        const float *c = s->coeff + (size_t)x * widest;
        int n0 = s->contrib[x].n0, n1 = s->contrib[x].n1;
        float *o = out + (size_t)x * 7;
        if (s->filter == F_POINT && s->scale == 1.0f) { /* stb:6173-6174 */
            memcpy(o, in + (size_t)(x - base) * 7, 7 * sizeof(float));
            continue;
        }
        int cnt = n1 - n0 + 1;
        if (cnt < 1) cnt = 1;
        if (widest <= 3) {
            for (int ch = 0; ch < 7; ch++) {
                float tot = in[(size_t)(n0 - base) * 7 + ch] * c[0];
                for (int k = 1; k < widest; k++)
                    tot += in[(size_t)(n0 + k - base) * 7 + ch] * c[k];
                o[ch] = tot;
            }
            continue;
        }
        /* number of taps the reference's loop touches: `widest` for the
         * fixed-width routines (<=12), else 4 + 4*max(1,ceil((cnt-4-mod)/4))
         * + mod.  Taps beyond cnt carry zero coefficients; we keep them only
         * as far as they are guaranteed to exist so that -0/+0 matches. */
        int taps;
        if (widest <= 12)
            taps = widest;
        else {
            int mod = widest & 3;
            int n   = ((cnt - 4 - mod) + 3) >> 2;
            if (n < 1) n = 1;
            taps = 4 + 4 * n + mod;
        }
        for (int ch = 0; ch < 7; ch++) {
            const float *d = in + (size_t)(n0 - base) * 7 + ch;
            float e        = d[0] * c[0];
            float od       = (1 < cnt) ? d[7] * c[1] : 0.0f * 0.0f;
            for (int k = 2; k < taps && k < cnt; k++) {
                if (k & 1)
                    od += d[(size_t)k * 7] * c[k];
                else
                    e += d[(size_t)k * 7] * c[k];
            }
            o[ch] = e + od;
        }
    }
}

/* fancy unweight (stb:4247-4292) + encode (stb:8415-8432 / 1391-1403) */
static void encode_row(const float *in, int w, uint8_t *out) {
    for (int x = 0; x < w; x++) {
        const float *p = in + (size_t)x * 7;
        float alpha    = p[3];
        float e[4];
        if (alpha < SMALL_FLOAT) {
            e[0] = p[0];
            e[1] = p[1];
            e[2] = p[2];
        } else {
            float ialpha = 1.0f / alpha;
            e[0]         = p[4] * ialpha;
            e[1]         = p[5] * ialpha;
            e[2]         = p[6] * ialpha;
        }
        e[3] = alpha;
        for (int c = 0; c < 4; c++) {
            float f = e[c] * 255.0f + 0.5f;
            if (f < 0) f = 0;
            if (f > 255) f = 255;
            out[(size_t)x * 4 + c] = (unsigned char)f;
        }
    }
}

/* Vertical pass for one output row over `n` floats: a strictly sequential
 * multiply-add chain in increasing input-row order.  Gather (stb:10036-10180,
 * 6180-6213) and scatter (stb:9864-10034, 6342-6372) give the same chain; the
 * only special case is a single ~1.0 coefficient, which copies
 * (stb:10049-10056). */
typedef const float *(*row_getter)(void *ctx, int row);

static void vertical_row(const plan_t *p, int y, row_getter get, void *ctx,
                         int n, float *out) {
    const sampler_t *s = &p->v;
    if (s->is_gather) {
        int n0 = s->contrib[y].n0, n1 = s->contrib[y].n1;
        const float *c = s->coeff + (size_t)y * s->coeff_width;
        if (n1 == n0 && (c[0] >= (1.0f - 0.000001f)) &&
            (c[0] <= (1.0f + 0.000001f))) {
            memcpy(out, get(ctx, n0), (size_t)n * sizeof(float));
            return;
        }
        const float *r = get(ctx, n0);
        for (int i = 0; i < n; i++) out[i] = r[i] * c[0];
        for (int k = n0 + 1; k <= n1; k++) {
            float ck = c[k - n0];
            r        = get(ctx, k);
            for (int i = 0; i < n; i++) out[i] += r[i] * ck;
        }
    } else {
        /* scatter: walk input rows in order, using the pivoted table */
        int margin = s->pixel_margin, first = 1;
        for (int iy = -margin; iy < s->in_size + margin; iy++) {
            const contrib_t *sc = p->v_scatter + (iy + margin);
            if (sc->n1 < sc->n0 || y < sc->n0 || y > sc->n1) continue;
            float ck = p->v_scatter_coeff[(size_t)(iy + margin) *
                                              p->v_scatter_width +
                                          (y - sc->n0)];
            const float *r = get(ctx, clamp_idx(iy, s->in_size));
            if (first) {
                for (int i = 0; i < n; i++) out[i] = r[i] * ck;
                first = 0;
            } else
                for (int i = 0; i < n; i++) out[i] += r[i] * ck;
        }
        if (first) memset(out, 0, (size_t)n * sizeof(float));
    }
}

typedef struct {
    const uint8_t *src;
    int sw, in_fmt, x0, x1;
    float *cache;    /* decoded rows cache: rows x width*7 */
    int *cache_row;  /* which row sits in each slot */
    int slots, width;
    /* for horizontal-first: rows are h-resampled */
    const plan_t *plan;
    int hfirst;
    float *tmp;
} rowsrc_t;

static const float *rowsrc_get(void *vctx, int row) {
    rowsrc_t *rs = (rowsrc_t *)vctx;
    int slot     = row % rs->slots;
    float *dst   = rs->cache + (size_t)slot * rs->width;
    if (rs->cache_row[slot] == row) return dst;
    const uint8_t *srow = rs->src + (size_t)row * rs->sw * 4;
    if (rs->hfirst) {
        decode_row(srow, rs->in_fmt, rs->x0, rs->x1, rs->tmp);
        hgather_row(rs->plan, rs->tmp, rs->x0, dst);
    } else
        decode_row(srow, rs->in_fmt, rs->x0, rs->x1, dst);
    rs->cache_row[slot] = row;
    return dst;
}

int oracle_scale(const uint8_t *src, int sw, int sh, int in_fmt, uint8_t *dst,
                 int dw, int dh, int filter) {
    plan_t p;
    if (plan_build(&p, sw, sh, dw, dh, filter) != 0) return -1;
    if (p.both_point) {
        /* 1:1 on both axes: point sampling, no alpha weighting, unscaled
         * coders (stb:7028-7030, 7397-7407) == copy (+ BGRA swizzle) */
        for (size_t i = 0; i < (size_t)sw * sh; i++) {
            dst[i * 4 + 0] = src[i * 4 + (in_fmt ? 2 : 0)];
            dst[i * 4 + 1] = src[i * 4 + 1];
            dst[i * 4 + 2] = src[i * 4 + (in_fmt ? 0 : 2)];
            dst[i * 4 + 3] = src[i * 4 + 3];
        }
        plan_free(&p);
        return 0;
    }
    int x0 = p.conservative.n0, x1 = p.conservative.n1;
    int in_w = x1 - x0 + 1;
    rowsrc_t rs;
    memset(&rs, 0, sizeof(rs));
    rs.src    = src;
    rs.sw     = sw;
    rs.in_fmt = in_fmt;
    rs.x0     = x0;
    rs.x1     = x1;
    rs.plan   = &p;
    rs.hfirst = !p.vertical_first;
    rs.width  = (rs.hfirst ? dw : in_w) * 7;
    rs.slots  = sh < 512 ? sh : 512; /* plain cache; a window of rows */
    if (rs.slots < p.v.widest + 2) rs.slots = p.v.widest + 2;
    rs.cache     = (float *)malloc((size_t)rs.slots * rs.width * sizeof(float));
    rs.cache_row = (int *)malloc((size_t)rs.slots * sizeof(int));
    for (int i = 0; i < rs.slots; i++) rs.cache_row[i] = -1;
    rs.tmp     = (float *)malloc((size_t)in_w * 7 * sizeof(float) + 64);
    float *vrow = (float *)malloc((size_t)rs.width * sizeof(float) + 64);
    float *hrow = (float *)malloc((size_t)dw * 7 * sizeof(float) + 64);
    for (int y = 0; y < dh; y++) {
        vertical_row(&p, y, rowsrc_get, &rs, rs.width, vrow);
        if (p.vertical_first) {
            hgather_row(&p, vrow, x0, hrow);
            encode_row(hrow, dw, dst + (size_t)y * dw * 4);
        } else
            encode_row(vrow, dw, dst + (size_t)y * dw * 4);
    }
    free(rs.cache);
    free(rs.cache_row);
    free(rs.tmp);
    free(vrow);
    free(hrow);
    plan_free(&p);
    return 0;
}

int oracle_scale_plan_info(int sw, int sh, int dw, int dh, int filter,
                           int info[6]) {
    plan_t p;
    if (plan_build(&p, sw, sh, dw, dh, filter) != 0) return -1;
    info[0] = p.vertical_first;
    info[1] = p.h.widest;
    info[2] = p.v.is_gather;
    info[3] = p.v.widest;
    info[4] = p.h.filter;
    info[5] = p.v.filter;
    plan_free(&p);
    return 0;
}

/* Test introspection: dump the normalised tap tables of a plan so that the
 * product's independently built tables can be compared on a CPU-only box.
 * header: {vertical_first, both_point, h_sequential, h_stride}.  h_taps is
 * {n0,count} per output column, h_coeff count floats per column at stride
 * h_stride; v_cnt per output row, v_rows/v_coeff flattened in accumulation
 * order.  Returns the number of vertical contributions or -1. */
int oracle_scale_plan_dump(int sw, int sh, int dw, int dh, int filter,
                           int *header, int *h_taps, float *h_coeff,
                           int h_coeff_cap, int *v_cnt, int *v_rows,
                           float *v_coeff, int v_cap) {
    plan_t p;
    if (plan_build(&p, sw, sh, dw, dh, filter) != 0) return -1;
    header[0] = p.vertical_first;
    header[1] = p.both_point;
    header[2] = p.h.coeff_width <= 3;
    header[3] = p.h.coeff_width;
    int total = 0, ok = 1;
    if ((long)dw * p.h.coeff_width > h_coeff_cap) ok = 0;
    for (int x = 0; ok && x < dw; x++) {
        int cnt = p.h.contrib[x].n1 - p.h.contrib[x].n0 + 1;
        if (cnt < 1) cnt = 1;
        h_taps[2 * x]     = p.h.contrib[x].n0;
        h_taps[2 * x + 1] = cnt;
        memcpy(h_coeff + (size_t)x * p.h.coeff_width,
               p.h.coeff + (size_t)x * p.h.coeff_width,
               (size_t)p.h.coeff_width * sizeof(float));
    }
    for (int y = 0; ok && y < dh; y++) {
        const sampler_t *s = &p.v;
        int n              = 0;
        if (s->is_gather) {
            int n0 = s->contrib[y].n0, n1 = s->contrib[y].n1;
            const float *c = s->coeff + (size_t)y * s->coeff_width;
            int cnt        = n1 - n0 + 1;
            if (cnt < 1) cnt = 1;
            int copy = (cnt == 1) && (c[0] >= (1.0f - 0.000001f)) &&
                       (c[0] <= (1.0f + 0.000001f));
            for (int k = 0; k < cnt; k++) {
                if (total >= v_cap) { ok = 0; break; }
                v_rows[total]  = n0 + k;
                v_coeff[total] = copy ? 1.0f : c[k];
                total++;
                n++;
            }
        } else {
            int margin = s->pixel_margin;
            for (int iy = -margin; iy < s->in_size + margin; iy++) {
                const contrib_t *sc = p.v_scatter + (iy + margin);
                if (sc->n1 < sc->n0 || y < sc->n0 || y > sc->n1) continue;
                if (total >= v_cap) { ok = 0; break; }
                v_rows[total]  = clamp_idx(iy, s->in_size);
                v_coeff[total] = p.v_scatter_coeff[(size_t)(iy + margin) *
                                                       p.v_scatter_width +
                                                   (y - sc->n0)];
                total++;
                n++;
            }
        }
        v_cnt[y] = n;
    }
    plan_free(&p);
    return ok ? total : -1;
}
