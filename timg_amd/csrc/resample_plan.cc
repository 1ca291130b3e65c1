// timg_amd/csrc/resample_plan.cc
//
// Builds the per-axis tap tables the HIP scaler consumes.  The tables have to
// reproduce stb_image_resize2 v2.12 as timg drives it (RGBA8, CLAMP edges,
// Mitchell when shrinking, its trapezoid "box" when enlarging, point at 1:1;
// src/image-scaler.cc:32,85-91) to the last fp32 bit, because the reference's
// escape-sequence output is compared byte for byte.  What matters for that:
//   * tap positions/weights are computed in fp32 exactly as stb does, weights
//     renormalised in double (stb_image_resize2.h:3461-3517), out-of-image taps
//     folded onto the edge pixel in stb's order (:3569-3605);
//   * rational scales reuse the first `numerator` phases (:3267-3275,:3523-3537);
//   * right-edge windows are slid back for stb's fixed-width loops (:3786-3862),
//     which changes the even/odd chain a tap lands in;
//   * tall vertical filters go through stb's scatter pivot (:3937-4003);
//   * the H-first/V-first order follows stb's cost heuristic (:6859-6906).
// Everything here is host code; no pixel ever passes through it.
#include "resample_plan.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace timg_amd {
namespace {

// 2^-120, stb_image_resize2.h:1104
const float kTiny = std::ldexp(1.0f, -120);

inline bool NearlyZero(float v) { return v < kTiny && v > -kTiny; }

float Support(int filter, float s) {
    switch (filter) {
    case kFilterBox: return 0.5f + s / 2.0f;
    case kFilterTriangle: return 1.0f;
    case kFilterMitchell: return 2.0f;
    default: return 0.5f;
    }
}

float Kernel(int filter, float x, float s) {
    if (filter == kFilterPoint) return 1.0f;
    if (x < 0.0f) x = -x;
    if (filter == kFilterBox) {
        const float half = s / 2;
        const float top  = 0.5f + half;
        if (x >= top) return 0.0f;
        const float flat = 0.5f - half;
        if (x <= flat) return 1.0f;
        return (top - x) / s;
    }
    if (filter == kFilterTriangle) return x <= 1.0f ? 1.0f - x : 0.0f;
    // Mitchell-Netravali B=C=1/3
    if (x < 1.0f) return (16.0f + x * x * (21.0f * x - 36.0f)) / 18.0f;
    if (x < 2.0f) return (32.0f + x * (-60.0f + x * (36.0f - 7.0f * x))) / 18.0f;
    return 0.0f;
}

struct Span {
    int lo = 0, hi = -1;
    int size() const { return hi - lo + 1; }
};

// One resampling axis.  Taps are kept per OUTPUT sample ("gather" form) in a
// dense [out][stride] array, like stb does while it builds them.
class Axis {
public:
    Axis(int in_size, int out_size) : in_(in_size), out_(out_size) {
        const double scale = (double)out_size / (double)in_size;
        scale_             = (float)scale;
        inv_scale_         = (float)(1.0 / scale);
        shift_             = 0.0f;
        rational_ = ToRational(scale, scale <= 1.0 ? (uint32_t)out_size
                                                    : (uint32_t)in_size,
                               scale >= 1.0);
    }

    void Configure(int filter, bool always_gather) {
        if (filter == kFilterDefault) {
            filter = kFilterMitchell;
            if (scale_ >= 1.0f - kTiny) {
                filter = (scale_ <= 1.0f + kTiny && std::ceil(shift_) == shift_)
                             ? kFilterPoint
                             : kFilterBox;
            }
        }
        filter_ = filter;
        const bool enlarging = scale_ >= 1.0f - kTiny;
        footprint_ = enlarging
                         ? (int)std::ceil(Support(filter, 1.0f / scale_) * 2.0f)
                         : (int)std::ceil(Support(filter, scale_) * 2.0f / scale_);
        mode_   = enlarging ? 1 : ((always_gather || footprint_ <= 32) ? 2 : 0);
        margin_ = footprint_ / 2;
        stride_ = footprint_;  // all three modes end up with this row length
        taps_.assign(out_, Span());
        w_.assign((size_t)out_ * stride_ + 1, 0.0f);
    }

    void Generate() {
        if (mode_ == 1)
            FromOutputSide();
        else
            FromInputSide();
        NormaliseAndFold();
    }

    // Input range a scanline decode has to cover (stb's "conservative" range).
    Span DecodeRange() const;

    // Slide right-edge windows back so a fixed `widest`-long loop stays inside
    // [.., row_last]; compacts rows to `widest` floats.
    void PackForFixedLoops(int row_last);

    // Scatter pivot: returns, per output sample, the (input, weight) list in
    // increasing input order exactly as stb's scatter loop would apply them.
    void PivotedRuns(std::vector<VRun> *runs, std::vector<int32_t> *rows,
                     std::vector<float> *coeff) const;
    void GatherRuns(std::vector<VRun> *runs, std::vector<int32_t> *rows,
                    std::vector<float> *coeff) const;

    int in_, out_;
    float scale_, inv_scale_, shift_;
    bool rational_   = false;
    uint32_t numer_ = 0, denom_ = 0;
    int filter_ = 0, footprint_ = 0, margin_ = 0, mode_ = 0, stride_ = 0;
    int widest_ = -1;
    std::vector<Span> taps_;
    std::vector<float> w_;

private:
    bool ToRational(double f, uint32_t limit, bool limit_denominator);
    bool Polyphase() const { return rational_ && (int)numer_ < out_; }
    void FromOutputSide();
    void FromInputSide();
    void NormaliseAndFold();
    void InputWindow(float out_center, float radius, int *first, int *last) const {
        const float lo = (out_center - radius + shift_) * inv_scale_;
        const float hi = (out_center + radius + shift_) * inv_scale_;
        *first         = (int)std::floor(lo + 0.5f);
        *last          = (int)std::floor(hi - 0.5f);
        if (*last < *first) *last = *first;
    }
    void OutputWindow(float in_center, float radius, int *first, int *last) const {
        const float lo = (in_center - radius) * scale_ - shift_;
        const float hi = (in_center + radius) * scale_ - shift_;
        *first         = std::max(0, (int)std::floor(lo + 0.5f));
        *last          = std::min(out_ - 1, (int)std::floor(hi - 0.5f));
    }
};

// Continued-fraction search for a ratio within one fp32 ulp-ish (2^-24) of f.
bool Axis::ToRational(double f, uint32_t limit, bool limit_denominator) {
    uint64_t top = (uint64_t)(f * (double)(1 << 25)), bot = 1 << 25;
    uint64_t n_prev = 0, d_prev = 1, n_cur = 1, d_cur = 0;
    const double eps = 1.0 / (double)(1 << 24);
    for (;;) {
        if ((limit_denominator ? d_cur : n_cur) >= limit) break;
        if (d_cur) {
            double err = (double)n_cur / (double)d_cur - f;
            if (err < 0) err = -err;
            if (err < eps) {
                numer_ = (uint32_t)n_cur;
                denom_ = (uint32_t)d_cur;
                return true;
            }
        }
        if (bot == 0) break;
        const uint64_t q = top / bot, r = top % bot;
        top = bot;
        bot = r;
        uint64_t t = q * d_cur + d_prev;
        d_prev     = d_cur;
        d_cur      = t;
        t          = q * n_cur + n_prev;
        n_prev     = n_cur;
        n_cur      = t;
    }
    if (limit_denominator) {
        n_cur = (uint64_t)(f * (double)limit + 0.5);
        d_cur = limit;
    } else {
        n_cur = limit;
        d_cur = (uint64_t)((double)limit / f + 0.5);
    }
    numer_     = (uint32_t)n_cur;
    denom_     = (uint32_t)d_cur;
    double err = d_cur ? (double)numer_ / (double)denom_ - f : 1.0;
    if (err < 0) err = -err;
    return err < eps;
}

// Enlarging: walk output samples, evaluate the kernel at every input sample
// under the (output-space) footprint.
void Axis::FromOutputSide() {
    const float radius = Support(filter_, inv_scale_) * scale_;
    const int phases   = Polyphase() ? (int)numer_ : out_;
    for (int o = 0; o < phases; ++o) {
        float *row           = &w_[(size_t)o * stride_];
        const float center_o = (float)o + 0.5f;
        const float center_i = (center_o + shift_) * inv_scale_;
        int first, last;
        InputWindow(center_o, radius, &first, &last);
        if (last - first + 1 > stride_) last = first + stride_ - 1;
        int last_kept = -1;
        for (int i = 0; i <= last - first; ++i) {
            const float pos = (float)(i + first) + 0.5f;
            float c         = Kernel(filter_, center_i - pos, inv_scale_);
            if (NearlyZero(c)) {
                if (i == 0) {  // drop leading zero taps entirely
                    ++first;
                    --i;
                    continue;
                }
                c = 0;
            } else {
                last_kept = i;
            }
            row[i] = c;
        }
        taps_[o].lo = first;
        taps_[o].hi = first + last_kept;
    }
}

// Shrinking: walk input samples (including the filter margin outside the
// image) and append each one's weight to every output sample it reaches.
void Axis::FromInputSide() {
    const float radius = Support(filter_, scale_) * inv_scale_;
    const bool poly    = Polyphase();
    int newest_output  = -1;
    for (int i = -margin_; i < in_ + margin_; ++i) {
        const float center_i = (float)i + 0.5f;
        const float mapped   = center_i * scale_ - shift_;
        int first, last;
        OutputWindow(center_i, radius, &first, &last);
        if (first > last) continue;
        if (poly) {
            if (first == (int)numer_) break;
            if (last >= (int)numer_) last = (int)numer_ - 1;
        }
        for (int o = first; o <= last; ++o) {
            const float center_o = (float)o + 0.5f;
            float c = Kernel(filter_, center_o - mapped, scale_) * scale_;
            if (NearlyZero(c)) c = 0.0f;
            float *row = &w_[(size_t)o * stride_];
            Span &t    = taps_[o];
            if (o > newest_output) {
                newest_output = o;
                t.lo = t.hi = i;
                row[0]      = c;
            } else {
                if (row[0] == 0.0f) t.lo = i;  // a leading zero is overwritten
                t.hi = i;
                if (i - t.lo < stride_) row[i - t.lo] = c;
            }
        }
    }
}

// stb's tap insertion, including its (harmless for CLAMP) loop quirks.
void InsertTap(Span *t, float *row, int pixel, float weight, int max_width) {
    if (pixel <= t->hi) {
        if (pixel < t->lo) {
            if (t->hi - pixel + 1 <= max_width) {
                const int shift = t->lo - pixel;
                for (int j = t->hi - t->lo; j <= 0; --j) row[j + shift] = row[j];
                for (int j = 1; j < shift; --j) row[j] = row[0];
                row[0] = weight;
                t->lo  = pixel;
            }
        } else {
            row[pixel - t->lo] += weight;
        }
    } else if (pixel - t->lo + 1 <= max_width) {
        const int at = pixel - t->lo;
        for (int j = t->size(); j < at; ++j) row[j] = 0;
        row[at] = weight;
        t->hi   = pixel;
    }
}

void Axis::NormaliseAndFold() {
    const int phases = Polyphase() ? (int)numer_ : out_;
    for (int o = 0; o < phases; ++o) {
        float *row = &w_[(size_t)o * stride_];
        Span &t    = taps_[o];
        double sum = 0;
        for (int i = 0; i < t.size(); ++i) sum += (double)row[i];
        if (sum < kTiny && sum > -kTiny) {
            t.hi   = t.lo;
            row[0] = 0.0f;
        } else if (sum < 1.0f - kTiny || sum > 1.0f + kTiny) {
            const double k = 1.0 / sum;
            for (int i = 0; i < t.size(); ++i) row[i] = (float)(row[i] * k);
        }
    }
    if (Polyphase()) {
        for (int o = (int)numer_; o < out_; ++o) {
            taps_[o].lo = taps_[o - numer_].lo + (int)denom_;
            taps_[o].hi = taps_[o - numer_].hi + (int)denom_;
            std::memcpy(&w_[(size_t)o * stride_], &w_[(size_t)(o - numer_) * stride_],
                        sizeof(float) * stride_);
        }
    }
    widest_ = -1;
    for (int o = 0; o < out_; ++o) {
        float *row = &w_[(size_t)o * stride_];
        Span &t    = taps_[o];
        // CLAMP: taps hanging over an edge are added onto the edge sample --
        // right side first, then left side from -1 outwards, the outermost
        // one last (it is re-inserted after the row has been shifted).
        if (t.hi > in_ - 1) {
            const int start = t.lo, end = t.hi;
            t.hi = in_ - 1;
            for (int i = in_; i <= end; ++i)
                InsertTap(&t, row, in_ - 1, row[i - start], stride_);
        }
        if (t.lo < 0) {
            const float *c = row - (t.lo + 1);
            for (int i = -1; i > t.lo; --i) InsertTap(&t, row, 0, *c--, stride_);
            const int old_lo      = t.lo;
            const float outermost = c[0];
            t.lo                  = 0;
            for (int i = 0; i <= t.hi; ++i) row[i] = row[i - old_lo];
            InsertTap(&t, row, 0, outermost, stride_);
        }
        if (t.lo <= t.hi) {
            int n = t.size();
            while (n && row[n - 1] == 0.0f) --n;
            t.hi = t.lo + n - 1;
            if (n > widest_ && t.lo <= t.hi) widest_ = n;
            for (int i = n; i < stride_; ++i) row[i] = 0.0f;
        }
    }
}

Span Axis::DecodeRange() const {
    Span r;
    int first, last;
    if (mode_ == 1) {
        const float radius = Support(filter_, inv_scale_) * scale_;
        InputWindow(0.5f, radius, &first, &last);
        r.lo = first;
        InputWindow((float)(out_ - 1) + 0.5f, radius, &first, &last);
        r.hi = last;
    } else {
        const float radius = Support(filter_, scale_) * inv_scale_;
        InputWindow(0.0f, 0.0f, &first, &last);
        r.lo = first;
        InputWindow((float)out_, 0.0f, &first, &last);
        r.hi = last;
        // unclamped variant of OutputWindow's emptiness test
        auto reaches = [&](int n, bool *empty) {
            const float c  = (float)n + 0.5f;
            const float lo = (c - radius) * scale_ - shift_;
            const float hi = (c + radius) * scale_ - shift_;
            int f          = (int)std::floor(lo + 0.5f);
            int l          = (int)std::floor(hi - 0.5f);
            if (f < 0) f = 0;
            if (l >= out_) l = out_ - 1;
            *empty = f > l;
            return f < out_ || l >= 0;
        };
        bool empty;
        for (int n = r.lo + 1; n >= -margin_; --n) {
            const bool hit = reaches(n, &empty);
            if (empty) break;
            if (hit) r.lo = n;
        }
        const int stop = r.hi - 1 + 1 + margin_;
        for (int n = r.hi - 1; n <= stop; ++n) {
            const bool hit = reaches(n, &empty);
            if (empty) break;
            if (hit) r.hi = n;
        }
    }
    r.lo = std::max(r.lo, 0);
    r.hi = std::min(r.hi, in_ - 1);
    return r;
}

void Axis::PackForFixedLoops(int row_last) {
    const int widest = widest_, row_end = row_last + 1;
    if (widest != stride_) {
        for (int o = 0; o < out_; ++o)
            std::memmove(&w_[(size_t)o * widest], &w_[(size_t)o * stride_],
                         sizeof(float) * widest);
        stride_ = widest;
    }
    // How far stb's horizontal loop reads for a window: `widest` taps for the
    // unrolled <=12 variants, otherwise groups of four plus widest%4.
    auto reach = [&](const Span &t) {
        if (widest <= 12) return widest;
        const int mod = widest & 3;
        int r         = ((t.size() - mod + 3) & ~3) + mod;
        return std::max(r, 8 + mod);
    };
    for (int o = out_ - 1; o >= 0 && taps_[o].lo + widest * 2 >= row_end; --o) {
        Span &t    = taps_[o];
        float *row = &w_[(size_t)o * widest];
        if (t.lo + widest <= row_end) continue;
        const int stop = reach(t);
        if (t.lo + stop <= row_end) continue;
        const int new_lo = row_end - stop;
        const int n      = t.size();
        const int back   = t.lo - new_lo;
        for (int i = n - 1; i >= 0; --i) row[i + back] = row[i];
        for (int i = back - 1; i >= 0; --i) row[i] = 0.0f;
        t.lo = new_lo;
    }
}

void Axis::GatherRuns(std::vector<VRun> *runs, std::vector<int32_t> *rows,
                      std::vector<float> *coeff) const {
    runs->resize(out_);
    for (int o = 0; o < out_; ++o) {
        const Span &t    = taps_[o];
        const float *row = &w_[(size_t)o * stride_];
        (*runs)[o].first = (int32_t)rows->size();
        const int n      = std::max(1, t.size());
        (*runs)[o].count = n;
        // A lone weight that is 1 within 1e-6 is applied as a copy by stb's
        // vertical gather (stb_image_resize2.h:10049-10056).
        const bool copy = n == 1 && row[0] >= 1.0f - 0.000001f &&
                          row[0] <= 1.0f + 0.000001f;
        for (int i = 0; i < n; ++i) {
            rows->push_back(t.lo + i);
            coeff->push_back(copy ? 1.0f : row[i]);
        }
    }
}

void Axis::PivotedRuns(std::vector<VRun> *runs, std::vector<int32_t> *rows,
                       std::vector<float> *coeff) const {
    // Per input sample (offset by the margin): the outputs it feeds.
    const int scatter_width = (int)std::ceil(Support(filter_, scale_) * 2.0f);
    const int n_in          = in_ + 2 * margin_;
    std::vector<Span> feeds(n_in);
    std::vector<float> fw((size_t)n_in * scatter_width + 1, 0.0f);
    int newest = -margin_ - 1;
    for (int o = 0; o < out_; ++o) {
        const Span &t  = taps_[o];
        const float *g = &w_[(size_t)o * stride_];
        for (int k = t.lo; k <= t.hi; ++k) {
            const float c = *g++;
            if (NearlyZero(c)) continue;
            Span &s    = feeds[k + margin_];
            float *row = &fw[(size_t)(k + margin_) * scatter_width];
            if (k > newest || s.lo > s.hi) {
                for (int z = newest + margin_ + 1; z < k + margin_; ++z)
                    feeds[z] = Span();
                s.lo = s.hi = o;
                row[0]      = c;
                newest      = k;
            } else {
                InsertTap(&s, row, o, c, scatter_width);
            }
        }
    }
    for (int z = newest + margin_ + 1; z < n_in; ++z) feeds[z] = Span();
    // Transpose back into per-output lists, inputs ascending.
    std::vector<int> count(out_, 0);
    for (int z = 0; z < n_in; ++z)
        for (int o = feeds[z].lo; o <= feeds[z].hi; ++o) ++count[o];
    runs->resize(out_);
    size_t base = rows->size();
    for (int o = 0; o < out_; ++o) {
        (*runs)[o].first = (int32_t)base;
        (*runs)[o].count = 0;
        base += count[o];
    }
    rows->resize(base);
    coeff->resize(base);
    for (int z = 0; z < n_in; ++z) {
        const Span &s = feeds[z];
        for (int o = s.lo; o <= s.hi; ++o) {
            VRun &r               = (*runs)[o];
            const size_t at       = (size_t)r.first + r.count++;
            (*rows)[at]           = std::min(std::max(z - margin_, 0), in_ - 1);
            (*coeff)[at] = fw[(size_t)z * scatter_width + (o - s.lo)];
        }
    }
}

// stb's trained cost model for 7-channel (RGBA + premultiplied) data.
bool VerticalFirst(const Axis &h, const Axis &v) {
    static const float kW[8][4] = {
        {0.00000f, 0.59375f, 0.00000f, 0.96875f}, {0.06250f, 0.81250f, 0.06250f, 0.59375f},
        {0.75000f, 0.43750f, 0.12500f, 0.96875f}, {0.87500f, 0.06250f, 0.18750f, 0.43750f},
        {1.00000f, 1.00000f, 1.00000f, 1.00000f}, {0.15625f, 0.12500f, 1.00000f, 1.00000f},
        {0.06250f, 0.12500f, 0.00000f, 1.00000f}, {0.00000f, 1.00000f, 0.03125f, 0.34375f},
    };
    int bucket;
    if (v.out_ <= 4 || h.out_ <= 4)
        bucket = v.out_ < h.out_ ? 6 : 7;
    else if (v.scale_ <= 1.0f)
        bucket = v.mode_ ? 1 : 0;
    else if (v.scale_ <= 2.0f)
        bucket = 2;
    else if (v.scale_ <= 3.0f)
        bucket = 3;
    else if (v.scale_ <= 4.0f)
        bucket = 5;
    else
        bucket = 6;
    const float *w      = kW[bucket];
    const double h_cost = (float)h.footprint_ * w[0] + h.scale_ * (float)v.footprint_ * w[1];
    const double v_cost = (float)v.footprint_ * w[2] + v.scale_ * (float)h.footprint_ * w[3];
    return v_cost <= h_cost;
}

}  // namespace

bool BuildResamplePlan(int in_w, int in_h, int in_fmt, int out_w, int out_h,
                       int filter, ResamplePlan *plan) {
    if (in_w <= 0 || in_h <= 0 || out_w <= 0 || out_h <= 0 || !plan) return false;
    *plan        = ResamplePlan();
    plan->in_w   = in_w;
    plan->in_h   = in_h;
    plan->out_w  = out_w;
    plan->out_h  = out_h;
    plan->in_fmt = in_fmt;

    Axis h(in_w, out_w), v(in_h, out_h);
    h.Configure(filter, /*always_gather=*/true);
    const Span decode = h.DecodeRange();
    v.Configure(filter, /*always_gather=*/false);
    plan->h_filter    = h.filter_;
    plan->v_filter    = v.filter_;
    plan->v_is_gather = v.mode_;
    plan->identity    = h.filter_ == kFilterPoint && v.filter_ == kFilterPoint;
    plan->vertical_first = VerticalFirst(h, v);

    h.Generate();
    // When both axes resample identically stb reuses the horizontal tables
    // *after* packing for the vertical axis.
    bool share = false;
    if (h.filter_ == v.filter_ && h.out_ == v.out_ && h.mode_ == v.mode_) {
        const float ds = std::fabs(h.scale_ - v.scale_);
        const float dp = std::fabs(h.shift_ - v.shift_);
        share          = ds <= kTiny && dp <= kTiny;
    }
    h.PackForFixedLoops(decode.hi);
    if (share) {
        const int keep_in = v.in_;
        v                 = h;
        v.in_             = keep_in;
    } else {
        v.Generate();
    }
    plan->v_widest = v.widest_;

    plan->h_width      = h.stride_;
    plan->h_sequential = h.stride_ <= 3;
    plan->h_taps.resize(out_w);
    plan->x_lo = in_w;
    plan->x_hi = -1;
    for (int x = 0; x < out_w; ++x) {
        const int n          = std::max(1, h.taps_[x].size());
        plan->h_taps[x].n0   = h.taps_[x].lo;
        plan->h_taps[x].count = n;
        plan->x_lo            = std::min(plan->x_lo, h.taps_[x].lo);
        plan->x_hi            = std::max(plan->x_hi, h.taps_[x].lo + n - 1);
        plan->max_h_count     = std::max(plan->max_h_count, n);
    }
    plan->h_coeff.assign(h.w_.begin(), h.w_.begin() + (size_t)out_w * h.stride_);

    if (v.mode_ == 0)
        v.PivotedRuns(&plan->v_runs, &plan->v_rows, &plan->v_coeff);
    else
        v.GatherRuns(&plan->v_runs, &plan->v_rows, &plan->v_coeff);

    std::vector<int> per_row(in_h + 1, 0);
    for (int y = 0; y < out_h; ++y) {
        const VRun &r     = plan->v_runs[y];
        plan->max_v_count = std::max(plan->max_v_count, r.count);
        if (r.count > 0) {
            // rows are ascending; treat the run as the interval it spans
            const int lo = plan->v_rows[r.first];
            const int hi = plan->v_rows[r.first + r.count - 1];
            per_row[lo] += 1;
            per_row[hi + 1] -= 1;
        }
    }
    int active = 0;
    for (int r = 0; r < in_h; ++r) {
        active += per_row[r];
        plan->max_active_rows = std::max(plan->max_active_rows, active);
    }
    return true;
}

}  // namespace timg_amd
