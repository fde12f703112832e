// Random-variate algorithms of the sampler, restated so that they run on the device.
//
// The reference draws through libstdc++ <random> (GCC 11 bits/random.tcc) on a 32-bit engine:
//   std::discrete_distribution<size_t>   src/Trellis.hpp:61-66, src/StateSequence/Mixture.hpp:111
//   std::gamma_distribution<float>       src/Distribution.hpp:80,127   (Marsaglia-Tsang, random.tcc:2338-2391)
//   std::normal_distribution<float>      src/Distribution.hpp:83       (Marsaglia polar,  random.tcc:1802-1835)
//   std::generate_canonical              random.tcc:3348-3380
// Each function below follows the same algorithm in the same arithmetic types; `Src` is any
// source of 32-bit words with `uint32_t next()`.  tests/test_dist.py feeds both these and the
// libstdc++ originals from the same engine and requires identical draws.
//
// Math policy `M` supplies logf/powf: hml_math.h on the device and in the checker's device mode,
// libm in the checker's reference mode.
#ifndef HML_DIST_H
#define HML_DIST_H

#include "hml_common.h"
#include "hml_math.h"
#include "hml_philox.h"

struct hml_devmath {
    static HML_HDM float logf_(float x) { return hml_logf(x); }
    static HML_HDM float powf_(float u, float p) { return hml_powf_unit(u, p); }
    static HML_HDM float sqrtf_(float x) { return HML_SQRTF(x); }
};

// generate_canonical<float,24> on a 32-bit engine: one word, (float)r / 2^32, clamped below 1.
template <class Src>
HML_HD float hml_canonical_f32(Src& src) {
    const uint32_t r = src.next();
    float ret = (float)r / 4294967296.0f;
    if (ret >= 1.0f) ret = 0.99999994f;  // nextafter(1.0f, 0.0f)
    return ret;
}

// generate_canonical<double,53> on a 32-bit engine: two words, r0 first.
HML_HD double hml_canonical_f64(uint32_t r0, uint32_t r1) {
    const double sum = (double)r0 + (double)r1 * 4294967296.0;
    double ret = sum / 18446744073709551616.0;
    if (ret >= 1.0) ret = 0.99999999999999989;  // nextafter(1.0, 0.0)
    return ret;
}

// The uniforms of the backward draws (sub-stream kind CAT).  Row t (1-based; block b = t - 1) is addressed by b: blocks
// 2m and 2m + 1 share Philox block m of the sweep - words (0, 1) make the uniform of the even block, words (2, 3) that of
// the odd one - so a lane that owns two consecutive rows pays for one block of ten rounds instead of two.
HML_HD void hml_cat_uniform_pair(hml_key key, uint64_t epoch, uint32_t pair, double& u_even, double& u_odd) {
    const hml_u32x4 o = hml_stream4(key, HML_KIND_CAT, epoch, pair, 0);
    u_even = hml_canonical_f64(o.v[0], o.v[1]);
    u_odd = hml_canonical_f64(o.v[2], o.v[3]);
}
HML_HD double hml_cat_uniform(hml_key key, uint64_t epoch, uint32_t t) {
    const uint32_t b = t - 1u;
    const hml_u32x4 o = hml_stream4(key, HML_KIND_CAT, epoch, b >> 1, 0);
    return (b & 1u) ? hml_canonical_f64(o.v[2], o.v[3]) : hml_canonical_f64(o.v[0], o.v[1]);
}

// std::discrete_distribution over K float weights: normalise in double, cumulative sums with the
// last forced to 1, lower_bound.  An all-zero row makes every probability NaN and libstdc++'s
// lower_bound then returns index 0 (observed in practice) - reproduced here.
// K == 1 consumes no random words in libstdc++; callers handle that case.
HML_HD int hml_categorical(const float* w, int K, double u) {
    double sum = 0.0;
    for (int i = 0; i < K; ++i) sum += (double)w[i];
    double cp = 0.0;
    int first = 0, count = K;
    // lower_bound(cp, cp+K, u) on cp_i, evaluated linearly: first i with !(cp_i < u)
    // std::lower_bound is a binary search; with NaNs it ends at index 0 because every
    // comparison `cp[mid] < u` is false.  A linear scan gives the same answer for a
    // non-decreasing sequence and for the all-NaN one.
    (void)first; (void)count;
    for (int i = 0; i < K; ++i) {
        cp += (double)w[i] / sum;
        const double c = (i == K - 1) ? 1.0 : cp;
        if (!(c < u)) return i;
    }
    return K - 1;
}

// Marsaglia polar normal with libstdc++'s saved-value behaviour (mean 0, stddev 1 object that
// lives inside a gamma_distribution, or a fresh object for the NIG mean).
template <class M>
struct hml_normal_f32 {
    float saved;
    bool have;
    HML_HDM hml_normal_f32() : saved(0.0f), have(false) {}
    template <class Src>
    HML_HDM float draw(Src& src, float mean, float stddev) { return draw_std(src) * stddev + mean; }
    // the variate before `* stddev + mean` (callers that learn mean and stddev later apply them themselves: same operations)
    template <class Src>
    HML_HDM float draw_std(Src& src) {
        float ret;
        if (have) {
            have = false;
            ret = saved;
        } else {
            float x, y, r2;
            do {
                x = (float)((double)(2.0f * hml_canonical_f32(src)) - 1.0);
                y = (float)((double)(2.0f * hml_canonical_f32(src)) - 1.0);
                r2 = x * x + y * y;
            } while (r2 > 1.0f || r2 == 0.0f);
            const float mult = M::sqrtf_(-2 * M::logf_(r2) / r2);
            saved = x * mult;
            have = true;
            ret = y * mult;
        }
        return ret;
    }
};

// gamma_distribution<float>(alpha, beta)(urng) with a fresh distribution object.
// libstdc++ nests three do-while loops (polar pair / v <= 0 / squeeze-and-log rejection) around a
// normal_distribution member that caches the pair's second variate.  The same sequence of draws and
// tests is written here as ONE loop with explicit state: hipcc (ROCm 7.2, -O3) mis-executes the
// nested form on gfx950 when the lanes of a wavefront leave the loops at different trip counts
// (tests/test_gpu_parity.py::test_gamma_lanes_independent keeps watch).
// hml_gamma_core_f32: everything but the final `* beta` - the rejection loop depends on alpha only, so a caller that learns
// beta later (the parameter kernel: beta needs the sweep's sums, alpha only its counts) can run it ahead.
template <class M, class Src>
HML_HD float hml_gamma_core_f32(Src& src, float alpha) {
    const float malpha = alpha < 1.0f ? alpha + 1.0f : alpha;
    const float a1 = malpha - 1.0f / 3.0f;
    const float a2 = 1.0f / M::sqrtf_(9.0f * a1);
    float saved = 0.0f;     // normal_distribution::_M_saved
    bool have = false;      // normal_distribution::_M_saved_available
    float v = 0.0f;
    bool accepted = false;
    while (!accepted) {
        float n;
        if (have) {
            have = false;
            n = saved;
        } else {
            float x = 0.0f, y = 0.0f, r2 = 2.0f;
            while (r2 > 1.0f || r2 == 0.0f) {
                x = (float)((double)(2.0f * hml_canonical_f32(src)) - 1.0);
                y = (float)((double)(2.0f * hml_canonical_f32(src)) - 1.0);
                r2 = x * x + y * y;
            }
            const float mult = M::sqrtf_(-2 * M::logf_(r2) / r2);
            saved = x * mult;
            have = true;
            n = y * mult;
        }
        n = n * 1.0f + 0.0f;                 // __ret * stddev() + mean() of the (0,1) member
        const float vv = 1.0f + a2 * n;
        if (!(vv <= 0.0f)) {                 // `while (__v <= 0.0)` draws another n
            v = vv * vv * vv;
            const float u = hml_canonical_f32(src);
            const bool c1 = (double)u > (double)1.0f - 0.0331 * (double)n * (double)n * (double)n * (double)n;
            const bool c2 = (double)M::logf_(u) >
                            (0.5 * (double)n * (double)n + (double)a1 * ((1.0 - (double)v) + (double)M::logf_(v)));
            accepted = !(c1 && c2);
        }
    }
    if (alpha == malpha) return a1 * v;
    float u = hml_canonical_f32(src);
    while (u == 0.0f) u = hml_canonical_f32(src);
    return M::powf_(u, 1.0f / alpha) * a1 * v;
}

template <class M, class Src>
HML_HD float hml_gamma_f32(Src& src, float alpha, float beta) {
    return hml_gamma_core_f32<M>(src, alpha) * beta;   // (a1 * v * beta and powf * a1 * v * beta associate from the left)
}

#endif
