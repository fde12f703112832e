// Host-side driver of the synthetic trace generator (hml_synth.h): multi-threaded, deterministic.
#ifndef HML_SYNTH_HOST_HPP
#define HML_SYNTH_HOST_HPP

#include <algorithm>
#include <thread>
#include <vector>

#include "hml_synth.h"

// Fills x[0..T) with a K-level piecewise-constant Gaussian trace; optionally writes the hidden
// level of every position to states[0..T).
inline void hml_synth_gauss_trace(float* x, int16_t* states, uint64_t T, int K, const float* mu, float sigma,
                                  double dwell, uint64_t seed, int nthreads) {
    if (T == 0) return;
    if (nthreads < 1) nthreads = 1;
    const uint32_t jump_thresh = dwell <= 1.0 ? 0xffffffffu : (uint32_t)(4294967296.0 / dwell);
    struct Jump { uint64_t t; hml_u32x4 w; };
    std::vector<std::vector<Jump>> jumps(nthreads);
    const uint64_t per = (T + nthreads - 1) / nthreads;
    auto pass1 = [&](int th) {
        const uint64_t a = (uint64_t)th * per, b = std::min(T, a + per);
        for (uint64_t t = a; t < b; ++t) {
            const hml_u32x4 w = hml_synth_words(seed, t);
            x[t] = (float)hml_synth_normal(w);
            if (t > 0 && hml_synth_jumps(w, jump_thresh)) jumps[th].push_back({t, w});
        }
    };
    {
        std::vector<std::thread> ths;
        for (int th = 0; th < nthreads; ++th) ths.emplace_back(pass1, th);
        for (auto& t : ths) t.join();
    }
    // resolve the hidden path sequentially over the (few) jump positions
    std::vector<uint64_t> jt;
    std::vector<int16_t> js;
    int s = K <= 1 ? 0 : (int)(hml_synth_words(seed, 0).v[3] % (uint32_t)K);
    jt.push_back(0); js.push_back((int16_t)s);
    for (int th = 0; th < nthreads; ++th)
        for (auto& j : jumps[th]) { s = hml_synth_target(j.w, s, K); jt.push_back(j.t); js.push_back((int16_t)s); }
    jt.push_back(T);
    auto pass3 = [&](int th) {
        const uint64_t a = (uint64_t)th * per, b = std::min(T, a + per);
        if (a >= b) return;
        size_t seg = (size_t)(std::upper_bound(jt.begin(), jt.end(), a) - jt.begin()) - 1;
        for (uint64_t t = a; t < b; ++t) {
            while (jt[seg + 1] <= t) ++seg;
            const int st = js[seg];
            x[t] = hml_synth_gauss_value(x[t], mu[st], sigma);
            if (states) states[t] = (int16_t)st;
        }
    };
    {
        std::vector<std::thread> ths;
        for (int th = 0; th < nthreads; ++th) ths.emplace_back(pass3, th);
        for (auto& t : ths) t.join();
    }
}


// Copy-number segments: diploid (cn = 2) runs with log-uniform length in [1e4, 1e6] alternate with CNV runs
// (cn in {0, 1, 3, 4}) of log-uniform length in [1e3, 1e5]; counts are Poisson-lognormal around depth * cn.
inline void hml_synth_depth_trace(float* x, int16_t* states, uint64_t T, double depth, double ln_sigma, uint64_t seed,
                                  int nthreads) {
    if (T == 0) return;
    if (nthreads < 1) nthreads = 1;
    std::vector<uint64_t> jt;
    std::vector<int16_t> js;
    uint64_t t = 0, seg = 0;
    bool diploid = true;
    while (t < T) {
        const hml_u32x4 w = hml_philox4x32_10((uint32_t)seg, (uint32_t)(seg >> 32), 2u, (uint32_t)HML_KIND_DATA << 24,
                                              (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5EEDu);
        const double u = ((double)w.v[0] + 0.5) / 4294967296.0;
        const double lo = diploid ? 1e4 : 1e3, hi = diploid ? 1e6 : 1e5;
        // log-uniform length: lo * (hi/lo)^u = lo * exp(-u' * ln(lo/hi)) with a non-positive exponent
        const double len = hi * hml_exp_nonpos(-(1.0 - u) * hml_log(hi / lo));
        const int cn = diploid ? 2 : (int)((const int[]){0, 1, 3, 4}[w.v[1] & 3u]);
        jt.push_back(t);
        js.push_back((int16_t)cn);
        t += (uint64_t)(len < 1.0 ? 1.0 : len);
        diploid = !diploid;
        ++seg;
    }
    jt.push_back(T);
    const uint64_t per = (T + nthreads - 1) / nthreads;
    auto fill = [&](int th) {
        const uint64_t a = (uint64_t)th * per, b = std::min(T, a + per);
        if (a >= b) return;
        size_t s = (size_t)(std::upper_bound(jt.begin(), jt.end(), a) - jt.begin()) - 1;
        for (uint64_t p = a; p < b; ++p) {
            while (jt[s + 1] <= p) ++s;
            x[p] = hml_synth_depth_value(seed, p, js[s], depth, ln_sigma);
            if (states) states[p] = js[s];
        }
    };
    std::vector<std::thread> ths;
    for (int th = 0; th < nthreads; ++th) ths.emplace_back(fill, th);
    for (auto& th : ths) th.join();
}

#endif
