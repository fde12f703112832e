// Synthetic piecewise-constant Gaussian traces: hidden Markov path with
// geometric dwell times (mean `dwell`), uniform jump to a different level, x_t = mu[s_t] + sigma*z_t.
// Every position draws from its own Philox counter, so the trace is a pure function of
// (seed, t) and can be generated in parallel, on any machine, bit-identically.
#ifndef HML_SYNTH_H
#define HML_SYNTH_H

#include "hml_common.h"
#include "hml_math.h"
#include "hml_philox.h"

// the four random words of position t
HML_HD hml_u32x4 hml_synth_words(uint64_t seed, uint64_t t) {
    return hml_philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), 0u, (uint32_t)HML_KIND_DATA << 24,
                             (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5EEDu);
}

// does the hidden path jump when entering position t (t >= 1)?
HML_HD bool hml_synth_jumps(hml_u32x4 w, uint32_t jump_thresh) { return w.v[2] < jump_thresh; }

// next level after a jump away from s (uniform over the other K-1 levels)
HML_HD int hml_synth_target(hml_u32x4 w, int s, int K) {
    return K <= 1 ? 0 : (int)(((uint32_t)s + 1u + w.v[3] % (uint32_t)(K - 1)) % (uint32_t)K);
}

// standard normal from the first two words (Box-Muller, cosine branch)
HML_HD double hml_synth_normal(hml_u32x4 w) {
    const double u1 = ((double)w.v[0] + 1.0) / 4294967296.0;  // (0,1]
    const double u2 = (double)w.v[1] / 4294967296.0;          // [0,1)
    double c, s;
    hml_sincos2pi(u2, &c, &s);
    return HML_SQRT(-2.0 * hml_log(u1)) * c;
}

// x_t = mu + sigma * (float)z, evaluated in float (one multiply, one add)
HML_HD float hml_synth_gauss_value(float z, float mu, float sigma) {
    const float sz = sigma * z;
    return mu + sz;
}


// ---- simulated read-depth trace: copy-number segments, Poisson-lognormal
// counts fed to the sampler as floats.  rate = max(0.5, depth * cn) * exp(ln_sigma * z - ln_sigma^2 / 2).
// second word block of a position (independent of hml_synth_words)
HML_HD hml_u32x4 hml_synth_words2(uint64_t seed, uint64_t t) {
    return hml_philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), 1u, (uint32_t)HML_KIND_DATA << 24,
                             (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5EEDu);
}

// Poisson(rate) from two uniforms' worth of randomness: exact inversion below 30, rounded normal above
HML_HD float hml_synth_poisson(double rate, hml_u32x4 w2) {
    if (rate < 30.0) {
        const double u = ((double)w2.v[0] + (double)w2.v[1] * 4294967296.0) / 18446744073709551616.0;
        double p = hml_exp_nonpos(-rate), cdf = p;
        int k = 0;
        while (u >= cdf && k < 200) { ++k; p = p * rate / (double)k; cdf += p; }
        return (float)k;
    }
    hml_u32x4 wz; wz.v[0] = w2.v[2]; wz.v[1] = w2.v[3]; wz.v[2] = 0; wz.v[3] = 0;
    const double z = hml_synth_normal(wz);
    const double v = rate + HML_SQRT(rate) * z + 0.5;
    const double f = v < 0.0 ? 0.0 : v;
    return (float)(double)(long long)f;   // floor for non-negative values
}

HML_HD float hml_synth_depth_value(uint64_t seed, uint64_t t, int cn, double depth, double ln_sigma) {
    const double z = hml_synth_normal(hml_synth_words(seed, t));
    double rate = depth * (double)cn;
    if (rate < 0.5) rate = 0.5;
    // exp(ln_sigma*z - ln_sigma^2/2): hml_exp_nonpos needs a non-positive argument, so split the sign
    const double a = ln_sigma * z - 0.5 * ln_sigma * ln_sigma;
    const double f = a <= 0.0 ? hml_exp_nonpos(a) : 1.0 / hml_exp_nonpos(-a);
    return hml_synth_poisson(rate * f, hml_synth_words2(seed, t));
}

#endif
