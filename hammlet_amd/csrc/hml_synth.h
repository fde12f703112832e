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

#endif
