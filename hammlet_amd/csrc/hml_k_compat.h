// The reference-compatible mode (hml_set_option "compat", `hammlet -compat`): one Gibbs sweep exactly as the reference's
// single thread computes it, on the GPU - so that a run with the reference's seed writes the reference's files.
// BASELINE.json's north star asks for the reference's state marginals "at a fixed RNG seed"; the default path cannot give
// them literally (D1 counter-based Philox instead of one sequential engine, D2 own logf / powf, D3
// tree sums, D4 exact integer counts), this mode removes all four:
//   D1  std::mt19937 (src/Distribution.hpp:15, src/main.cpp:107-108) restated on the device - the engine's state lives
//       in device memory, one lane draws from it in the reference's order: B categorical draws from the last block to the
//       first (two 32-bit outputs each, ForwardBackward.hpp:133-162 / Trellis.hpp:61-66; in block order for a mixture
//       sweep, Mixture.hpp:111), then theta_0 .. theta_{K-1} (gamma, normal), pi, the rows of A (HMM.hpp:110-115) with
//       libstdc++'s variate algorithms (hml_dist.h);
//   D2  expf / logf / powf of the reference's libm (hml_math_glibc.h: glibc 2.35's algorithms in its FMA build);
//   D3  per-state sums of the block statistics by one float Kahan aggregator in block order (ForwardBackward.hpp:189-192,
//       KahanAggregator.hpp:26-45);
//   D4  transition and occupancy counts as `size_t += float` (ForwardBackward.hpp:183-187: they round above 2^24).
// The order-dependent part - filter, backward draws, count pass, conjugate updates, parameter draws - is one lane of one
// wavefront walking the blocks (this mode is for traces up to ~10^6 positions; the default path is the fast one); block
// enumeration, block statistics and the marginals use the same kernels as the default path (integer-exact there).
// Univariate models only (D = 1).
#ifndef HML_K_COMPAT_H
#define HML_K_COMPAT_H

#include "hml_dist.h"
#include "hml_k_forward.h"
#include "hml_math_glibc.h"
#include "hml_state.h"

struct hml_glibcmath {
    static __device__ __forceinline__ float logf_(float x) { return hml_glibc_logf(x); }
    static __device__ __forceinline__ float powf_(float u, float p) { return hml_glibc_powf_unit(u, p); }
    static __device__ __forceinline__ float sqrtf_(float x) { return HML_SQRTF(x); }   // correctly rounded, like glibc's
};

// std::mt19937: 624 words of state and the index of the next output
#define HML_MT_N 624
struct hml_mt_state {
    uint32_t mt[HML_MT_N];
    uint32_t idx;
};
// seeding of mersenne_twister_engine(value) (bits/random.tcc: _M_x[0] = value mod 2^32, the Knuth recurrence behind it)
static inline void hml_mt_seed(hml_mt_state* s, uint64_t seed) {
    s->mt[0] = (uint32_t)seed;
    for (uint32_t i = 1; i < HML_MT_N; ++i) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + i;
    s->idx = HML_MT_N;
}

#if defined(__HIPCC__)
// the engine of ONE lane, state in LDS (copied in and out by the kernel)
struct hml_mt_src {
    uint32_t* mt;
    uint32_t idx;
    __device__ __forceinline__ uint32_t next() {
        if (idx >= HML_MT_N) {
            for (int k = 0; k < HML_MT_N; ++k) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % HML_MT_N] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % HML_MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0u;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};

// Trellis::sample / std::discrete_distribution on the sequential engine: two outputs, r0 first (hml_dist.h)
__device__ __forceinline__ int hml_compat_categorical(hml_mt_src& src, const float* w, int K) {
    const uint32_t r0 = src.next(), r1 = src.next();
    return hml_categorical(w, K, hml_canonical_f64(r0, r1));
}

// theta_k from its posterior (Distribution<NormalInverseGamma>::resample, Distribution.hpp:77-87), derived values as the
// sweep needs them, posterior back to the prior (Theta.hpp:203-211)
__device__ __forceinline__ void hml_compat_draw_theta(hml_model* mdl, hml_mt_src& src, int P) {
    for (int k = 0; k < P; ++k) {
        const float alpha = mdl->nig_post[k][0], beta = mdl->nig_post[k][1], mu0 = mdl->nig_post[k][2], nu = mdl->nig_post[k][3];
        const float g = hml_gamma_f32<hml_glibcmath>(src, alpha, (float)(1.0 / (double)beta));
        const float v = (float)(1.0 / (double)g);
        hml_normal_f32<hml_glibcmath> nd;
        const float m = nd.draw(src, mu0, HML_SQRTF(v / nu));
        if (!hml_isfinite(m)) hml_raise(mdl, HML_DEVERR_MEAN_NOT_FINITE, m);
        if (!hml_isfinite(v)) hml_raise(mdl, HML_DEVERR_VAR_NOT_FINITE, v);
        else if (v <= 0.0f) hml_raise(mdl, HML_DEVERR_VAR_NOT_POSITIVE, v);
        mdl->mu[k] = m; mdl->var[k] = v; mdl->sd[k] = HML_SQRTF(v);
        mdl->rvar2[k] = 1.0 / (2.0 * (double)v);
        for (int i = 0; i < 4; ++i) mdl->nig_post[k][i] = mdl->nig_prior[i];
    }
}
// dirichlet_sample (Distribution.hpp:116-139): gammas in index order, float running sum, then the quotients
__device__ __forceinline__ void hml_compat_dirichlet(hml_mt_src& src, const float* alphas, float* probs, int n) {
    float sum = 0.0f;
    for (int d = 0; d < n; ++d) { const float r = hml_gamma_f32<hml_glibcmath>(src, alphas[d], 1.0f); probs[d] = r; sum += r; }
    for (int d = 0; d < n; ++d) probs[d] = probs[d] / sum;
}
__device__ __forceinline__ void hml_compat_draw_pi_A(hml_model* mdl, hml_mt_src& src, int K) {
    hml_compat_dirichlet(src, mdl->dirPi, mdl->pi, K);
    for (int k = 0; k < K; ++k) mdl->dirPi[k] = mdl->pi_alpha;
    for (int i = 0; i < K; ++i) hml_compat_dirichlet(src, mdl->dirA + i * K, mdl->A + i * K, K);
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) mdl->dirA[i * K + j] = (i == j) ? mdl->a_diag : mdl->a_off;
}
// logNormalizer, log A_ss and the threshold of the current parameters with the reference's logf (hml_derive's values)
__device__ __forceinline__ void hml_compat_derive(hml_model* mdl, int K) {
    float mv = HML_INF_F;
    const int P = mdl->P, D = mdl->D;
    for (int k = 0; k < P; ++k) {   // per emission parameter (EFD.hpp:35-38; threshold: Theta.hpp:227-234)
        const float m = mdl->mu[k], v = mdl->var[k];
        mdl->logN[k] = hml_glibc_logf(mdl->sd[k]) + m * m / (2 * v);
        mv = (v < mv) ? v : mv;
    }
    for (int k = 0; k < K; ++k) {
        // theta.logNormalizer(state): float sum over the state's parameters from 0, in dimension order (Theta.hpp:148-158)
        float r = 0.0f;
        for (int d = 0; d < D; ++d) r += mdl->logN[mdl->map[k][d]];
        mdl->logNs[k] = r;
        mdl->logA[k] = hml_glibc_logf(mdl->A[k * K + k]);
    }
    const float l = hml_glibc_logf((float)mdl->T);
    const float arg = 2 * l * mv;
    const float t = HML_SQRTF(arg);
    mdl->thr_theta = t;
    if (mdl->dynamic) mdl->thr = t;
}

// after hml_set_parameters: the derived values of injected parameters with the mode's own logf
template <int K>
HML_KERNEL __launch_bounds__(64) void hml_k_compat_derive(hml_model* __restrict__ mdl) {
    if (threadIdx.x == 0) hml_compat_derive(mdl, K);
}

// mode 1: theta, pi, A from the (reset) priors (main.cpp:393-401); mode 2: Theta's constructor draw (Theta.hpp:126-127)
template <int K>
HML_KERNEL __launch_bounds__(64) void hml_k_compat_draw(hml_model* __restrict__ mdl, hml_mt_state* __restrict__ mts, int mode) {
    __shared__ uint32_t lmt[HML_MT_N];
    for (int i = threadIdx.x; i < HML_MT_N; i += 64) lmt[i] = mts->mt[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    hml_mt_src src{lmt, mts->idx};
    hml_compat_draw_theta(mdl, src, mdl->P);
    if (mode != 2) hml_compat_draw_pi_A(mdl, src, K);
    hml_compat_derive(mdl, K);
    mdl->epoch += 1ull;
    for (int i = 0; i < HML_MT_N; ++i) mts->mt[i] = lmt[i];
    mts->idx = src.idx;
}

// One sweep (sampleHMM's body, HMM.hpp:99-121) over the blocks the launches before it enumerated: method 0 =
// StateSequence<ForwardBackward>::sample (ForwardBackward.hpp:16-213), 1 = StateSequence<Mixture>::sample
// (Mixture.hpp:31-144).  rows: (B + 1) x K floats, row 0 = pi.
template <int K>
HML_KERNEL __launch_bounds__(64) void hml_k_compat_sweep(hml_model* __restrict__ mdl, hml_mt_state* __restrict__ mts,
                                                         const uint32_t* __restrict__ starts, const float2* __restrict__ bstat,
                                                         float* __restrict__ rows, int16_t* __restrict__ q, int method,
                                                         float* __restrict__ eprobe, float* __restrict__ aprobe) {
    __shared__ uint32_t lmt[HML_MT_N];
    __shared__ unsigned long long s_trans[K * K], s_occ[K], s_n[K];
    __shared__ float s_ps[K], s_pq[K], s_es[K], s_eq[K];   // KahanAggregator per state: positive sums and their error terms
    __shared__ float s_w[K], s_logA[K], s_logN[K];
    if (mdl->halted != 0u) return;   // (hml_state.h: the sweep's blocks did not fit the chain's buffers; the host grows them and sweeps again)
    for (int i = threadIdx.x; i < HML_MT_N; i += 64) lmt[i] = mts->mt[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    hml_mt_src src{lmt, mts->idx};
    const uint32_t B = mdl->B;
    const bool self = mdl->self_trans != 0;
    const int P = mdl->P, D = mdl->D;                 // "-s C P D": P emission parameters over D data dimensions (Mapping.hpp:53-137)
    const uint64_t dstride = mdl->stat_stride;        // block statistics: one plane per dimension
    // innerProduct(y, theta.value(), theta.mapping(s)) (EFD.hpp:83-93): float sum over the dimensions from 0, every term
    // the univariate product (EFD.hpp:23-32: double inside)
    auto ip_state = [&](uint32_t b, int s) -> float {
        float r = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float2 st = bstat[(uint64_t)d * dstride + b];
            const int pp = mdl->map[s][d];
            const float ip = (float)((2.0 * (double)mdl->mu[pp] * (double)st.x - (double)st.y) / (2.0 * (double)mdl->var[pp]));
            if (!hml_isfinite(ip)) hml_raise(mdl, HML_DEVERR_IP_NOT_FINITE, ip);
            r += ip;
        }
        return r;
    };
    for (int s = 0; s < K; ++s) {
        s_logA[s] = self ? mdl->logA[s] : 0.0f;
        s_logN[s] = mdl->logNs[s];
        s_occ[s] = 0ull; s_n[s] = 0ull; s_ps[s] = 0.0f; s_pq[s] = 0.0f; s_es[s] = 0.0f; s_eq[s] = 0.0f;
        for (int j = 0; j < K; ++j) s_trans[s * K + j] = 0ull;
    }
    unsigned long long nfb = 0ull;
    if (method == 0) {
        // ---- forward (ForwardBackward.hpp:57-123)
        float prev[K], fwd[K];
        for (int s = 0; s < K; ++s) { prev[s] = mdl->pi[s]; rows[s] = prev[s]; if (aprobe) aprobe[s] = prev[s]; }
        float prevN = 1.0f;
        for (uint32_t t = 1; t <= B; ++t) {
            const uint32_t b = t - 1u;
            const float N = (float)(starts[t] - starts[b]);
            float maxE = -3.40282346638528859812e+38f;
            for (int s = 0; s < K; ++s) {
                float E = ip_state(b, s) - N * s_logN[s];
                if (self) E += (N - 1.0f) * s_logA[s];
                fwd[s] = E;
                maxE = (E < maxE) ? maxE : E;
                if (eprobe) eprobe[(uint64_t)b * K + s] = E;
            }
            for (int s = 0; s < K; ++s) fwd[s] = hml_glibc_expf(fwd[s] - maxE);
            float Z = 0.0f;
            for (int j = 0; j < K; ++j) {
                float tt = 0.0f;
                for (int i = 0; i < K; ++i) tt += prev[i] * mdl->A[i * K + j];
                fwd[j] *= tt;
                Z += fwd[j];
            }
            if (Z != 0.0f) { for (int j = 0; j < K; ++j) fwd[j] = fwd[j] / Z; }
            else { nfb++; for (int j = 0; j < K; ++j) fwd[j] = (float)(1.0 / (double)(float)K); }
            if (aprobe) for (int s = 0; s < K; ++s) aprobe[(uint64_t)t * K + s] = fwd[s];
            if (self) for (int s = 0; s < K; ++s) rows[(uint64_t)(t - 1u) * K + s] = prev[s] * hml_glibc_expf((prevN - 1.0f) * s_logA[s]);   // :115-119
            for (int s = 0; s < K; ++s) { rows[(uint64_t)t * K + s] = fwd[s]; prev[s] = fwd[s]; }
            prevN = N;
        }
        // ---- backward (ForwardBackward.hpp:133-162)
        for (int s = 0; s < K; ++s) s_w[s] = rows[(uint64_t)B * K + s];
        int j = hml_compat_categorical(src, s_w, K);
        q[B - 1u] = (int16_t)j;
        for (uint32_t tt = B - 1u; tt > 0u; --tt) {
            for (int i = 0; i < K; ++i) {
                const float r = rows[(uint64_t)tt * K + i] * mdl->A[i * K + j];
                if (r < 0.0f) hml_raise(mdl, HML_DEVERR_NEG_BACKWARD, r);
                s_w[i] = r;
            }
            j = hml_compat_categorical(src, s_w, K);
            q[tt - 1u] = (int16_t)j;
        }
    } else {
        // ---- mixture (Mixture.hpp:54-112): one draw per block in block order, no transitions
        for (uint32_t b = 0; b < B; ++b) {
            const float N = (float)(starts[b + 1u] - starts[b]);   // (size_t N, converted where it meets a float)
            float maxE = -3.40282346638528859812e+38f;
            for (int s = 0; s < K; ++s) {
                const float E = ip_state(b, s) - N * s_logN[s];
                s_w[s] = E;
                maxE = (E < maxE) ? maxE : E;
                if (eprobe) eprobe[(uint64_t)b * K + s] = E;
            }
            for (int s = 0; s < K; ++s) s_w[s] = hml_glibc_expf(s_w[s] - maxE);
            q[b] = (int16_t)hml_compat_categorical(src, s_w, K);
        }
    }
    // ---- count pass in block order (ForwardBackward.hpp:170-200 / Mixture.hpp:113-141)
    int prevs = 0;
    for (uint32_t b = 0; b < B; ++b) {
        const uint32_t n = starts[b + 1u] - starts[b];
        const int s = q[b];
        if (method == 1) {
            s_occ[s] += n;
            s_trans[s * K + s] += n - 1u;
        } else {
            const float N = (float)n;   // size_t += float: the sum goes through a float
            s_trans[s * K + s] = (unsigned long long)((float)s_trans[s * K + s] + (N - 1.0f));
            s_occ[s] = (unsigned long long)((float)s_occ[s] + N);
        }
        s_trans[prevs * K + s] += 1ull;
        // stats[mapping[state][d]].add(y.suffStat(d), N) for every dimension in order (ForwardBackward.hpp:189-192)
        for (int d = 0; d < D; ++d) {
            const int pp = mdl->map[s][d];
            const float2 st = bstat[(uint64_t)d * dstride + b];
            { const float y = st.x - s_es[pp], t = s_ps[pp] + y; s_es[pp] = (t - s_ps[pp]) - y; s_ps[pp] = t; }
            { const float y = st.y - s_eq[pp], t = s_pq[pp] + y; s_eq[pp] = (t - s_pq[pp]) - y; s_pq[pp] = t; }
            s_n[pp] += n;
        }
        prevs = s;
    }
    // ---- conjugate updates (Conjugate.hpp:121-168,178-205)
    for (int k = 0; k < K; ++k) mdl->last_occ[k] = s_occ[k];
    for (int k = 0; k < P; ++k) {   // tau_theta.addObservation per parameter (ForwardBackward.hpp:202-207)
        const float sum = s_ps[k] - 0.0f, sumSq = s_pq[k] - 0.0f;   // KahanAggregator::sum(): positive part minus the (empty) negative part
        mdl->last_sum[k] = sum; mdl->last_sumsq[k] = sumSq;
        if (s_n[k] > 0ull) {
            if (sumSq < 0.0f) hml_raise(mdl, HML_DEVERR_NEG_SUMSQ, sumSq);
            const double N = (double)s_n[k];
            const float alpha = mdl->nig_post[k][0], beta = mdl->nig_post[k][1], mu0 = mdl->nig_post[k][2], nu = mdl->nig_post[k][3];
            const float xbar = (float)((double)sum / N);
            float ssN = (float)((double)(sum * sum) / N);
            if (ssN > sumSq) ssN = sumSq;
            const float na = (float)((double)alpha + N / 2.0);
            const float dxm = (xbar - mu0) * (xbar - mu0);
            const float nb = (float)((double)beta + (((double)sumSq + (N * (double)nu / (N + (double)nu)) * (double)dxm) - (double)ssN) / 2.0);
            const float nm = (float)((double)(nu * mu0 + sum) / ((double)nu + N));
            const float nn = (float)((double)nu + N);
            if (na <= 0.0f) hml_raise(mdl, HML_DEVERR_NIG_ALPHA, na);
            if (nb <= 0.0f) hml_raise(mdl, HML_DEVERR_NIG_BETA, nb);
            if (nn <= 0.0f) hml_raise(mdl, HML_DEVERR_NIG_NU, nn);
            if (!hml_isfinite(nm)) hml_raise(mdl, HML_DEVERR_NIG_MU0, nm);
            mdl->nig_post[k][0] = na; mdl->nig_post[k][1] = nb; mdl->nig_post[k][2] = nm; mdl->nig_post[k][3] = nn;
        }
    }
    for (int i = 0; i < K; ++i) {
        for (int j = 0; j < K; ++j) { mdl->dirA[i * K + j] += (float)s_trans[i * K + j]; mdl->last_trans[i * K + j] = s_trans[i * K + j]; }
        mdl->dirPi[i] += (float)s_occ[i];
    }
    // ---- theta, pi, A (HMM.hpp:111-115), derived values
    hml_compat_draw_theta(mdl, src, P);
    hml_compat_draw_pi_A(mdl, src, K);
    hml_compat_derive(mdl, K);
    mdl->uniform_fallbacks += nfb;
    mdl->epoch += 1ull;
    mdl->sweeps += 1ull;
    mdl->block_updates += (unsigned long long)B;
    for (int i = 0; i < HML_MT_N; ++i) mts->mt[i] = lmt[i];
    mts->idx = src.idx;
}
#endif

#endif
