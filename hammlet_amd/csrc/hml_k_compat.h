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
// Models over several data dimensions ("-s C P D") since round 4.
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
HML_KERNEL __launch_bounds__(64) void hml_k_compat_derive(hml_model* __restrict__ mdl) {
    if (threadIdx.x == 0) hml_compat_derive(mdl, mdl->K);
}

// mode 1: theta, pi, A from the (reset) priors (main.cpp:393-401); mode 2: Theta's constructor draw (Theta.hpp:126-127)
HML_KERNEL __launch_bounds__(64) void hml_k_compat_draw(hml_model* __restrict__ mdl, hml_mt_state* __restrict__ mts, int mode) {
    __shared__ uint32_t lmt[HML_MT_N];
    for (int i = threadIdx.x; i < HML_MT_N; i += 64) lmt[i] = mts->mt[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int K = mdl->K;
    hml_mt_src src{lmt, mts->idx};
    hml_compat_draw_theta(mdl, src, mdl->P);
    if (mode != 2) hml_compat_draw_pi_A(mdl, src, K);
    hml_compat_derive(mdl, K);
    mdl->epoch += 1ull;
    for (int i = 0; i < HML_MT_N; ++i) mts->mt[i] = lmt[i];
    mts->idx = src.idx;
}

// ------------------------------------------------------------------------------------------------------------------------
// One sweep (sampleHMM's body, HMM.hpp:99-121) over the blocks the launches before it enumerated, in four launches (round 4;
// rounds 3 had one lane walk the whole sweep: 3.9 us per block at config 3 - a chain of dependent memory round trips -
// seven times slower than the reference on one CPU core).  The number of states is a run-time value here (up to
// HML_CAP_K = 64: one lane per state), so the mode also serves models the default path's register-resident kernels do not
// instantiate (K > 16).  Everything is the reference's arithmetic in the reference's order:
//   hml_k_compat_emission   the blocks' emission terms, a lane per block (independent between blocks: EFD.hpp:23-38,83-93,
//                           ForwardBackward.hpp:67-84 / Mixture.hpp:54-77);
//   hml_k_compat_forward / _backward   filter and backward draws by ONE wavefront, lane j = state j: the K sums over the predecessors run
//                           side by side (each lane its own, i = 0 .. K-1 in order), the row sum Z and the categorical's double
//                           sums are taken serially over the lanes in index order (v_readlane), blocks staged through LDS 64 at a
//                           time; the engine's outputs are tempered a tile ahead (they do not depend on the data) and its twist
//                           runs on all lanes;
//   hml_k_compat_mixture    Mixture.hpp:90-112: a lane per block, the draws in block order;
//   hml_k_compat_update     count pass in block order (float Kahan sums, `size_t += float` counts), conjugate updates, parameter
//                           draws: one lane, its blocks staged through LDS by all.
// ------------------------------------------------------------------------------------------------------------------------
#define HML_COMPAT_TILE 1024   // floats staged per tile: min(64, 1024 / K) blocks

__device__ __forceinline__ float hml_lane_f32(float v, int i) {   // (i wave-uniform)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
}
__device__ __forceinline__ double hml_lane_f64(double v, int i) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, i), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), i);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void hml_compat_fence() {   // LDS operations of one wavefront complete in order; keep the compiler's order too
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// mersenne_twister_engine::_M_gen_rand on 64 lanes: the in-place recurrence reads mt[k + 1] and mt[k + 397] before they are
// replaced for k < 227 and the replaced mt[k - 227] (mt[0] for k = 623) afterwards - chunks of 64 in rising order, every
// chunk's reads before its writes, see exactly those values
__device__ __forceinline__ void hml_mt_twist_wave(uint32_t* mt, int lane) {
    for (int k0 = 0; k0 < HML_MT_N; k0 += 64) {
        const int k = k0 + lane;
        uint32_t nv = 0u;
        if (k < HML_MT_N) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % HML_MT_N] & 0x7fffffffu);
            nv = mt[(k + 397) % HML_MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        hml_compat_fence();
        if (k < HML_MT_N) mt[k] = nv;
        hml_compat_fence();
    }
}
// the engine's next n outputs -> out[0 .. n) (n, idx wave-uniform)
__device__ __forceinline__ void hml_mt_fill_wave(uint32_t* mt, uint32_t& idx, uint32_t* out, uint32_t n, int lane) {
    uint32_t produced = 0u;
    while (produced < n) {
        if (idx >= HML_MT_N) { hml_mt_twist_wave(mt, lane); idx = 0u; }
        const uint32_t m = (n - produced < HML_MT_N - idx) ? n - produced : HML_MT_N - idx;
        for (uint32_t k = (uint32_t)lane; k < m; k += 64u) {
            uint32_t y = mt[idx + k];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            out[produced + k] = y;
        }
        idx += m; produced += m;
    }
    hml_compat_fence();
}

// hml_categorical (hml_dist.h: std::discrete_distribution + lower_bound) with weight i in lane i: the same double sums in
// the same order, the K quotients side by side
__device__ __forceinline__ int hml_compat_categorical_exact(float w, int K, double u) {
    const double wd = (double)w;
    double sum = 0.0;
    for (int i = 0; i < K; ++i) sum += hml_lane_f64(wd, i);
    const double p = wd / sum;
    double cp = 0.0;
    for (int i = 0; i < K; ++i) {
        cp += hml_lane_f64(p, i);
        const double c = (i == K - 1) ? 1.0 : cp;
        if (!(c < u)) return i;
    }
    return K - 1;
}
// ... behind a screen that settles all but one draw in 10^6 without the quotients: the cumulative probability cp_i differs
// from t_i / sum (t_i = w_0 + ... + w_i in double, sum = t_{K-1}: the reference's own sum, same order) by at most (K + 1) 2^-52
// relatively - K rounded quotients and their K rounded additions - so wherever t_i and u sum are further apart than 2^-30 sum,
// `cp_i < u` is `t_i < u sum`.  A draw closer than that to a boundary, weights that are negative or not finite, and an all-zero
// row (every probability NaN: index 0 in libstdc++) take the literal form.  KC as in the kernels.
template <int KC>
__device__ __forceinline__ int hml_compat_categorical_wave(float w, int K, double u, int lane) {
    const double wd = (double)w;
    double t = 0.0;   // lane i: t_i
    if (KC) {
#pragma unroll
        for (int i = 0; i < (KC ? KC : 1); ++i) { const double x = hml_lane_f64(wd, i); t = (lane >= i) ? t + x : t; }
    } else {
        for (int i = 0; i < K; ++i) { const double x = hml_lane_f64(wd, i); t = (lane >= i) ? t + x : t; }
    }
    const double sum = hml_lane_f64(t, K - 1);
    const double us = u * sum, margin = sum * 9.31322574615478515625e-10;   // 2^-30
    const bool inner = lane < K - 1;
    const double d = t - us;
    const bool unclear = inner && !(d > margin || d < -margin);
    const bool ok = sum > 0.0 && sum < 1.7976931348623157e308 && !(w < 0.0f);
    if (__builtin_expect(__ballot(unclear || (lane < K && !ok)) != 0ull, 0)) return hml_compat_categorical_exact(w, K, u);
    const unsigned long long reached = __ballot(inner && d > 0.0);   // !(cp_i < u)
    return reached ? __ffsll((long long)reached) - 1 : K - 1;
}

// emission terms: em[b * K + s] = expf(E_s - max E) (method 0: with the self-transition term of a block, ForwardBackward.hpp:74-84);
// g[b * K + s] = expf((N_b - 1) log A(s, s)), the factor the filter rescales the block's row with once the next row exists (:115-119)
HML_KERNEL __launch_bounds__(256) void hml_k_compat_emission(hml_model* __restrict__ mdl, const uint32_t* __restrict__ starts,
                                                             const float2* __restrict__ bstat, float* __restrict__ em, float* __restrict__ g,
                                                             int method, float* __restrict__ eprobe) {
    if (mdl->halted != 0u) return;
    const uint32_t B = mdl->B;
    const int K = mdl->K, D = mdl->D;
    const bool self = mdl->self_trans != 0 && method == 0;
    const uint64_t dstride = mdl->stat_stride;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (uint64_t)gridDim.x * blockDim.x) {
        const float N = (float)(starts[b + 1u] - starts[b]);   // (size_t N, converted where it meets a float)
        float* const e = em + b * (uint64_t)K;
        float maxE = -3.40282346638528859812e+38f;
        for (int s = 0; s < K; ++s) {
            // innerProduct(y, theta.value(), theta.mapping(s)) (EFD.hpp:83-93): float sum over the dimensions from 0, every
            // term the univariate product (EFD.hpp:23-32: double inside)
            float r = 0.0f;
            for (int d = 0; d < D; ++d) {
                const float2 st = bstat[(uint64_t)d * dstride + b];
                const int pp = mdl->map[s][d];
                const float ip = (float)((2.0 * (double)mdl->mu[pp] * (double)st.x - (double)st.y) / (2.0 * (double)mdl->var[pp]));
                if (!hml_isfinite(ip)) hml_raise(mdl, HML_DEVERR_IP_NOT_FINITE, ip);
                r += ip;
            }
            float E = r - N * mdl->logNs[s];
            if (self) E += (N - 1.0f) * mdl->logA[s];
            e[s] = E;
            maxE = (E < maxE) ? maxE : E;
            if (eprobe) eprobe[b * (uint64_t)K + s] = E;
        }
        for (int s = 0; s < K; ++s) e[s] = hml_glibc_expf(e[s] - maxE);
        if (self) for (int s = 0; s < K; ++s) g[b * (uint64_t)K + s] = hml_glibc_expf((N - 1.0f) * mdl->logA[s]);
    }
}

// StateSequence<ForwardBackward>::sample's filter (ForwardBackward.hpp:86-123).  rows: (B + 1) x K floats, row 0 = pi; row t
// as the backward pass reads it - rescaled by A(s, s)^(N_t - 1) (:115-119) - except the last.  KC: the number of states as
// a compile-time value (2 .. 16: A's column in registers, loops unrolled) or 0 = the model's value (A in LDS).
template <int KC>
HML_KERNEL __launch_bounds__(64) void hml_k_compat_forward(hml_model* __restrict__ mdl, const uint32_t* __restrict__ starts,
                                                           const float* __restrict__ em, const float* __restrict__ g,
                                                           float* __restrict__ rows, float* __restrict__ aprobe) {
    __shared__ float sA[KC ? 1 : HML_CAP_K * HML_CAP_K];   // row-major: lane j reads A(i, j)
    __shared__ float tile[HML_COMPAT_TILE], tile_g[HML_COMPAT_TILE];
    if (mdl->halted != 0u) return;
    const int lane = threadIdx.x;
    const int K = KC ? KC : mdl->K;
    const uint32_t B = mdl->B;
    const bool self = mdl->self_trans != 0;
    const bool act = lane < K;
    float acol[KC ? KC : 1];
    if (KC) {
#pragma unroll
        for (int i = 0; i < (KC ? KC : 1); ++i) acol[i] = act ? mdl->A[i * K + lane] : 0.0f;
    } else {
        for (int i = lane; i < K * K; i += 64) sA[i] = mdl->A[i];
    }
    float prev = act ? mdl->pi[lane] : 0.0f;
    if (act) { rows[lane] = prev; if (aprobe) aprobe[lane] = prev; }
    // the factor of row 0: a "block" of size 1 before the first (prevN = 1, ForwardBackward.hpp:107)
    float gprev = (act && self) ? hml_glibc_expf((1.0f - 1.0f) * mdl->logA[lane]) : 1.0f;
    const uint32_t TB = (uint32_t)((HML_COMPAT_TILE / K) < 64 ? (HML_COMPAT_TILE / K) : 64);
    hml_compat_fence();
    unsigned long long nfb = 0ull;
    constexpr int NREG = HML_COMPAT_TILE / 64;
    float nxt[NREG], nxt_g[NREG];
    auto fetch = [&](uint32_t b0) {   // a tile's emission terms and factors into registers (they travel during the tile before)
        const uint32_t nb = (b0 < B) ? ((B - b0 < TB) ? B - b0 : TB) : 0u;
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            const uint32_t i = (uint32_t)r * 64u + (uint32_t)lane;
            const bool in = i < nb * (uint32_t)K;
            nxt[r] = in ? em[(uint64_t)b0 * K + i] : 0.0f;
            nxt_g[r] = (in && self) ? g[(uint64_t)b0 * K + i] : 1.0f;
        }
    };
    fetch(0u);
    for (uint32_t b0 = 0; b0 < B; b0 += TB) {
        const uint32_t nb = (B - b0 < TB) ? B - b0 : TB;
#pragma unroll
        for (int r = 0; r < NREG; ++r) { tile[r * 64 + lane] = nxt[r]; tile_g[r * 64 + lane] = nxt_g[r]; }
        hml_compat_fence();
        fetch(b0 + TB);
        float e_nx = act ? tile[lane] : 0.0f, g_nx = act ? tile_g[lane] : 1.0f;   // (one block ahead of the filter)
        for (uint32_t r = 0; r < nb; ++r) {
            const uint64_t t = (uint64_t)b0 + r + 1u;
            float f = e_nx;
            const float gcur = g_nx;
            if (r + 1u < nb) { e_nx = act ? tile[(r + 1u) * (uint32_t)K + (uint32_t)lane] : 0.0f; g_nx = act ? tile_g[(r + 1u) * (uint32_t)K + (uint32_t)lane] : 1.0f; }
            float tt = 0.0f;
            if (KC) {
#pragma unroll
                for (int i = 0; i < (KC ? KC : 1); ++i) tt += hml_lane_f32(prev, i) * acol[i];
            } else {
                for (int i = 0; i < K; ++i) tt += hml_lane_f32(prev, i) * sA[i * K + lane];   // (lanes >= K read inside the array; their values are not used)
            }
            f *= tt;
            float Z = 0.0f;
            if (KC) {
#pragma unroll
                for (int j = 0; j < (KC ? KC : 1); ++j) Z += hml_lane_f32(f, j);
            } else {
                for (int j = 0; j < K; ++j) Z += hml_lane_f32(f, j);
            }
            float fw;
            if (Z != 0.0f) fw = f / Z;
            else { nfb++; fw = (float)(1.0 / (double)(float)K); }
            if (act) {
                if (aprobe) aprobe[t * K + lane] = fw;
                if (self) rows[(t - 1u) * K + lane] = prev * gprev;   // :115-119
                rows[t * K + lane] = fw;
            }
            prev = act ? fw : 0.0f;
            gprev = gcur;
        }
        hml_compat_fence();
    }
    if (lane == 0) mdl->uniform_fallbacks += nfb;
}

// ... and its backward draws (ForwardBackward.hpp:133-162; Trellis::sample, Trellis.hpp:61-66): the last row, then rows
// B-1 .. 1 with weights rows[t][i] * A(i, q_{t+1}), two engine outputs per draw
template <int KC>
HML_KERNEL __launch_bounds__(64) void hml_k_compat_backward(hml_model* __restrict__ mdl, hml_mt_state* __restrict__ mts,
                                                            const float* __restrict__ rows, int16_t* __restrict__ q) {
    __shared__ uint32_t lmt[HML_MT_N];
    __shared__ float sAT[KC ? 1 : HML_CAP_K * HML_CAP_K];   // transposed: lane i reads A(i, j) at [j * K + i]
    __shared__ float tile[HML_COMPAT_TILE];
    __shared__ uint32_t draws[128];
    __shared__ int16_t tile_q[64];
    if (mdl->halted != 0u) return;
    const int lane = threadIdx.x;
    const int K = KC ? KC : mdl->K;
    const uint32_t B = mdl->B;
    const bool act = lane < K;
    for (int i = lane; i < HML_MT_N; i += 64) lmt[i] = mts->mt[i];
    float arow[KC ? KC : 1];   // A(lane, j)
    if (KC) {
#pragma unroll
        for (int j = 0; j < (KC ? KC : 1); ++j) arow[j] = act ? mdl->A[lane * K + j] : 0.0f;
    } else {
        for (int idx = lane; idx < K * K; idx += 64) { const int i = idx / K, j = idx - i * K; sAT[j * K + i] = mdl->A[idx]; }
    }
    const uint32_t TB = (uint32_t)((HML_COMPAT_TILE / K) < 64 ? (HML_COMPAT_TILE / K) : 64);
    uint32_t idx = mts->idx;
    hml_compat_fence();
    int j = 0;
    {
        hml_mt_fill_wave(lmt, idx, draws, 2u, lane);
        const float w = act ? rows[(uint64_t)B * K + lane] : 0.0f;
        j = hml_compat_categorical_wave<KC>(w, K, hml_canonical_f64(draws[0], draws[1]), lane);
        if (lane == 0) q[B - 1u] = (int16_t)j;
        hml_compat_fence();
    }
    constexpr int NREG = HML_COMPAT_TILE / 64;
    float nxt[NREG];
    auto fetch = [&](uint32_t hi) {   // rows lo + 1 .. hi into registers
        const uint32_t nb = hi < TB ? hi : TB;
        const uint32_t lo = hi - nb;
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            const uint32_t i = (uint32_t)r * 64u + (uint32_t)lane;
            nxt[r] = (i < nb * (uint32_t)K) ? rows[(uint64_t)(lo + 1u) * K + i] : 0.0f;
        }
    };
    fetch(B - 1u);
    for (uint32_t hi = B - 1u; hi > 0u; ) {   // rows lo + 1 .. hi, taken downwards
        const uint32_t nb = hi < TB ? hi : TB;
        const uint32_t lo = hi - nb;
#pragma unroll
        for (int r = 0; r < NREG; ++r) tile[r * 64 + lane] = nxt[r];
        hml_mt_fill_wave(lmt, idx, draws, 2u * nb, lane);   // (fences)
        fetch(lo);
        float row_nx = act ? tile[(nb - 1u) * (uint32_t)K + (uint32_t)lane] : 0.0f;   // (one row ahead of the draws)
        uint32_t d0 = draws[0], d1 = draws[1];
        for (uint32_t r = 0; r < nb; ++r) {
            const float row = row_nx;
            const double u = hml_canonical_f64(d0, d1);
            if (r + 1u < nb) { row_nx = act ? tile[(nb - 2u - r) * (uint32_t)K + (uint32_t)lane] : 0.0f; d0 = draws[2u * r + 2u]; d1 = draws[2u * r + 3u]; }
            float a;
            if (KC) {
                a = arow[0];
#pragma unroll
                for (int jj = 1; jj < (KC ? KC : 1); ++jj) a = (j == jj) ? arow[jj] : a;
            } else a = sAT[j * K + lane];
            const float w = act ? row * a : 0.0f;
            const unsigned long long neg = __ballot(w < 0.0f);
            if (neg != 0ull) {   // ForwardBackward.hpp:147-149 (the first negative weight in state order is the one reported)
                const int first = __ffsll((long long)neg) - 1;
                if (lane == first) hml_raise(mdl, HML_DEVERR_NEG_BACKWARD, w);
                hml_compat_fence();
                if (w < 0.0f && lane != first) hml_raise(mdl, HML_DEVERR_NEG_BACKWARD, w);
            }
            j = hml_compat_categorical_wave<KC>(w, K, u, lane);
            if (lane == 0) tile_q[r] = (int16_t)j;
        }
        hml_compat_fence();
        if ((uint32_t)lane < nb) q[hi - 1u - (uint32_t)lane] = tile_q[lane];   // q[t - 1] of row t = hi - lane
        hml_compat_fence();
        hi = lo;
    }
    for (int i = lane; i < HML_MT_N; i += 64) mts->mt[i] = lmt[i];
    if (lane == 0) mts->idx = idx;
}

// StateSequence<Mixture>::sample's draws (Mixture.hpp:90-112): one per block in block order, no transitions
HML_KERNEL __launch_bounds__(64) void hml_k_compat_mixture(hml_model* __restrict__ mdl, hml_mt_state* __restrict__ mts,
                                                           const float* __restrict__ em, int16_t* __restrict__ q) {
    __shared__ uint32_t lmt[HML_MT_N];
    __shared__ float tile[HML_COMPAT_TILE];
    __shared__ uint32_t draws[128];
    if (mdl->halted != 0u) return;
    const int lane = threadIdx.x;
    const int K = mdl->K;
    const uint32_t B = mdl->B;
    for (int i = lane; i < HML_MT_N; i += 64) lmt[i] = mts->mt[i];
    uint32_t idx = mts->idx;
    const uint32_t TB = (uint32_t)((HML_COMPAT_TILE / K) < 64 ? (HML_COMPAT_TILE / K) : 64);
    hml_compat_fence();
    for (uint32_t b0 = 0; b0 < B; b0 += TB) {
        const uint32_t nb = (B - b0 < TB) ? B - b0 : TB;
        for (uint32_t i = (uint32_t)lane; i < nb * (uint32_t)K; i += 64u) tile[i] = em[(uint64_t)b0 * K + i];
        hml_mt_fill_wave(lmt, idx, draws, 2u * nb, lane);   // (fences)
        if ((uint32_t)lane < nb) q[b0 + lane] = (int16_t)hml_categorical(tile + (uint32_t)lane * (uint32_t)K, K, hml_canonical_f64(draws[2 * lane], draws[2 * lane + 1]));
        hml_compat_fence();
    }
    for (int i = lane; i < HML_MT_N; i += 64) mts->mt[i] = lmt[i];
    if (lane == 0) mts->idx = idx;
}

// count pass in block order (ForwardBackward.hpp:170-200 / Mixture.hpp:113-141), conjugate updates (Conjugate.hpp:121-168,
// 178-205), theta, pi, A (HMM.hpp:111-115), derived values.  The aggregators live in registers, lane = parameter (Kahan sums,
// term counts) or state (occupancy, the diagonal transition count - the two that go through a float); a block's state, size and
// statistics reach all lanes as scalars (v_readlane from the lane that loaded them), the lane concerned takes them.  The
// off-diagonal transition counts are plain integer increments in LDS.
HML_KERNEL __launch_bounds__(64) void hml_k_compat_update(hml_model* __restrict__ mdl, hml_mt_state* __restrict__ mts,
                                                          const uint32_t* __restrict__ starts, const float2* __restrict__ bstat,
                                                          const int16_t* __restrict__ q, int method) {
    __shared__ uint32_t lmt[HML_MT_N];
    __shared__ unsigned long long s_trans[HML_CAP_K * HML_CAP_K], s_occ[HML_CAP_K], s_n[HML_CAP_K];
    __shared__ float s_ps[HML_CAP_K], s_pq[HML_CAP_K];
    __shared__ uint8_t s_map[HML_CAP_K][HML_MAX_D];
    if (mdl->halted != 0u) return;
    const int lane = threadIdx.x;
    const int K = mdl->K, P = mdl->P, D = mdl->D;
    const uint32_t B = mdl->B;
    const uint64_t dstride = mdl->stat_stride;        // block statistics: one plane per dimension
    for (int i = lane; i < HML_MT_N; i += 64) lmt[i] = mts->mt[i];
    for (int i = lane; i < K * K; i += 64) s_trans[i] = 0ull;
    if (lane < K) for (int d = 0; d < HML_MAX_D; ++d) s_map[lane][d] = mdl->map[lane][d];
    hml_compat_fence();
    // lane = parameter: KahanAggregator (positive sum and its error term) of Sx and Sxx, number of terms; lane = state: occupancy, A(s, s)'s count
    float ps = 0.0f, pq = 0.0f, es = 0.0f, eq = 0.0f;
    // (occupancy and diagonal count as doubles: integers below 2^53 are exact there, and float <-> double conversions are one
    // instruction each where float <-> 64-bit integer ones are twenty: (float)(double)x == (float)x for such x)
    unsigned long long n_terms = 0ull;
    double occ = 0.0, diag = 0.0;
    int prevs = 0;
    uint32_t r_n = 0u; int r_q = 0; float2 r_st[HML_MAX_D];
    auto fetch = [&](uint32_t b0) {
        const uint32_t b = b0 + (uint32_t)lane;
        const bool in = b < B;
        r_n = in ? starts[b + 1u] - starts[b] : 0u;
        r_q = in ? (int)q[b] : 0;
#pragma unroll
        for (int d = 0; d < HML_MAX_D; ++d) r_st[d] = (in && d < D) ? bstat[(uint64_t)d * dstride + b] : make_float2(0.0f, 0.0f);
    };
    fetch(0u);
    for (uint32_t b0 = 0; b0 < B; b0 += 64u) {
        const uint32_t nb = (B - b0 < 64u) ? B - b0 : 64u;
        const uint32_t c_n = r_n; const int c_q = r_q;
        float2 c_st[HML_MAX_D];
#pragma unroll
        for (int d = 0; d < HML_MAX_D; ++d) c_st[d] = r_st[d];
        fetch(b0 + 64u);   // (travels while this tile is walked)
        for (uint32_t r = 0; r < nb; ++r) {
            const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)c_n, (int)r);
            const int s = __builtin_amdgcn_readlane(c_q, (int)r);
            if (method == 1) {
                if (lane == s) { occ += (double)n; diag += (double)(n - 1u); }
            } else if (lane == s) {
                const float N = (float)n;   // size_t += float: the sum goes through a float
                diag = (double)((float)diag + (N - 1.0f));
                occ = (double)((float)occ + N);
            }
            if (prevs == s) { if (lane == s) diag += 1.0; }
            else if (lane == 0) atomicAdd(&s_trans[prevs * K + s], 1ull);   // (no return value: the increment does not hold up the walk)
            // stats[mapping[state][d]].add(y.suffStat(d), N) for every dimension in order (ForwardBackward.hpp:189-192)
            if (D == 1) {
                const float sx = hml_lane_f32(c_st[0].x, (int)r), sq = hml_lane_f32(c_st[0].y, (int)r);
                if (lane == s) {
                    { const float y = sx - es, t = ps + y; es = (t - ps) - y; ps = t; }
                    { const float y = sq - eq, t = pq + y; eq = (t - pq) - y; pq = t; }
                    n_terms += n;
                }
            } else {
#pragma unroll
                for (int d = 0; d < HML_MAX_D; ++d) {
                    if (d < D) {
                        const int pp = s_map[s][d];
                        const float sx = hml_lane_f32(c_st[d].x, (int)r), sq = hml_lane_f32(c_st[d].y, (int)r);
                        if (lane == pp) {
                            { const float y = sx - es, t = ps + y; es = (t - ps) - y; ps = t; }
                            { const float y = sq - eq, t = pq + y; eq = (t - pq) - y; pq = t; }
                            n_terms += n;
                        }
                    }
                }
            }
            prevs = s;
        }
    }
    hml_compat_fence();
    if (lane < K) { s_trans[lane * K + lane] = (unsigned long long)diag; s_occ[lane] = (unsigned long long)occ; s_n[lane] = n_terms; s_ps[lane] = ps; s_pq[lane] = pq; }
    hml_compat_fence();
    if (lane != 0) return;
    hml_mt_src src{lmt, mts->idx};
    // ---- conjugate updates
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
    mdl->epoch += 1ull;
    mdl->sweeps += 1ull;
    mdl->block_updates += (unsigned long long)B;
    for (int i = 0; i < HML_MT_N; ++i) mts->mt[i] = lmt[i];
    mts->idx = src.idx;
}
#endif

#endif
