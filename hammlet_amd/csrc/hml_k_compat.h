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
// The order-dependent part - filter, backward draws, count pass, conjugate updates, parameter draws - keeps the reference's
// arithmetic and order but not its one thread (round 4, below: a lane per state, chunks that are checked against each other,
// the count pass by state); block enumeration, block statistics and the marginals use the same kernels as the default path
// (integer-exact there).  Models over several data dimensions ("-s C P D") and 2 .. 64 states.
#ifndef HML_K_COMPAT_H
#define HML_K_COMPAT_H

#include "hml_dist.h"
#include "hml_k_forward.h"
#include "hml_math_glibc.h"
#include "hml_state.h"

// expf of the two modes that share the kernels below: the reference's libm (this mode) or hml_math.h's (the default path's
// arithmetic: models of more than 16 states run these kernels with it, hml_k_wide.h)
struct hml_glibc_exp { static __device__ __forceinline__ float expf_(float x) { return hml_glibc_expf(x); } };
struct hml_dev_exp { static __device__ __forceinline__ float expf_(float x) { return hml_expf(x); } };

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
// One sweep (sampleHMM's body, HMM.hpp:99-121) over the blocks the launches before it enumerated (round 4; round 3 had one
// lane walk the whole sweep: 3.9 us per block at config 3 - a chain of dependent memory round trips - seven times slower than
// the reference on one CPU core).  The number of states is a run-time value here (up to HML_CAP_K = 64: one lane per state), so
// the mode also serves models the default path's register-resident kernels do not instantiate (K > 16).  Everything is the
// reference's arithmetic in the reference's order:
//   hml_k_compat_emission   the blocks' emission terms, a lane per block (independent between blocks: EFD.hpp:23-38,83-93,
//                           ForwardBackward.hpp:67-84 / Mixture.hpp:54-77);
//   hml_k_compat_draws      the engine's outputs of the sweep's categorical draws, ahead of them (they do not depend on the data);
//   hml_k_compat_forward / _backward (+ _check)   filter and backward draws, a wavefront per CHUNK of blocks, lane j = state j: the K
//                           sums over the predecessors run side by side (each lane its own, i = 0 .. K-1 in order), the row sum Z
//                           and the categorical's double sums are taken serially over the lanes in index order (v_readlane); chunks
//                           start from a guess some blocks early and are checked against each other (see "in CHUNKS" below);
//   hml_k_compat_mixture    Mixture.hpp:90-112: a lane per block, the draws in block order;
//   hml_k_compat_part_*, hml_k_compat_update   count pass (float Kahan sums, `size_t += float` counts) by state over stably
//                           partitioned lists - or in block order: one lane, its blocks staged by all - conjugate updates, parameter
//                           draws.
// ------------------------------------------------------------------------------------------------------------------------
#define HML_COMPAT_TILE 1024   // floats staged per tile: min(64, 1024 / K) blocks
#define HML_COMPAT_MAX_CHUNKS 16384   // (round 5: 2048 before - a chunk is one wavefront, and the machine holds 8000 of them)

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
// The running sums t_i = (..(w_0 + w_1) + ..) + w_i, lane i its own, by SHIFTS (round 5): every lane repeats t <- t_of_the_lane_below
// + w.  After step s the lanes up to s hold their final value and keep recomputing it from the same operands, so no lane needs
// to be told when to stop: two DPP moves and an addition per step where the broadcast form took two v_readlane, the addition,
// a comparison and two selects.  (t_0 = w_0 where the loop from zero had 0.0 + w_0: the same value but for a weight of -0.0,
// and a row whose sum is a zero of either sign takes the literal form below.)
// PAD: KC is an upper bound of the model's K (a multiple of four): the loop is unrolled over KC in groups of four that are skipped
// from K on.
__device__ __forceinline__ double hml_wave_shr1_f64(double v) {   // lane i <- lane i - 1, lane 0 <- 0.0 (DPP wave_shr:1)
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, 0x138, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), 0x138, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo);
}
template <int KC, bool PAD = false>
__device__ __forceinline__ int hml_compat_categorical_wave(float w, int K, double u, int lane) {
    const double wd = (double)w;
    double t = wd;   // lane i: t_i
    if (KC && PAD) {
#pragma unroll
        for (int i0 = 0; i0 < (KC ? KC : 1); i0 += 4) {
            if (i0 < K) {   // wave-uniform (steps beyond K - 1 recompute final values)
#pragma unroll
                for (int i = i0; i < i0 + 4; ++i) if (i > 0) t = hml_wave_shr1_f64(t) + wd;
            }
        }
    } else if (KC) {
#pragma unroll
        for (int i = 1; i < (KC ? KC : 1); ++i) t = hml_wave_shr1_f64(t) + wd;
    } else {
        for (int i = 1; i < K; ++i) t = hml_wave_shr1_f64(t) + wd;
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
template <class M>
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
        for (int s = 0; s < K; ++s) e[s] = M::expf_(e[s] - maxE);
        if (self) for (int s = 0; s < K; ++s) g[b * (uint64_t)K + s] = M::expf_((N - 1.0f) * mdl->logA[s]);
    }
}

// ---- filter and backward draws in CHUNKS (round 4).  Both recursions forget where they started: a chunk of L blocks is run by
// its own wavefront from a guessed start W blocks earlier (the filter: a flat row; the backward draws: state 0 - all chains of
// draws share the uniforms, so paths from different states coalesce), and is RIGHT exactly if the value it reached at its first
// block equals, bit for bit, what the chunk before it left there.  A second launch of one wavefront walks the chunks in order,
// compares, and runs the rare chunk that was wrong again from the right value (then the next comparison uses what it left).
// The first chunk starts from the true value, so by induction every stored row / state is the sequential one.  One chunk
// (gridDim.x = 1) is the sequential form: probes, short sweeps.
struct hml_compat_chunks {
    float* entry;        // [C][K] filter: the row a chunk reached at its first block (from its warm-up)
    float* exitv;        // [C][K] ... and the unscaled row it left behind its last block
    uint32_t* bad;       // [C] filter: the chunk's entry row is not what the chunk before it left (hml_k_compat_forward_verify)
    uint32_t* nfb;       // [C] uniform fallbacks inside the chunk's own blocks
    int32_t* in_state;   // [C] backward draws: the state above the chunk's first row (from its warm-up)
    int32_t* out_state;  // [C] ... and the state of its last row
    uint32_t W;          // warm-up, blocks; HML_CHUNK_W_ADAPTIVE: the model's own (mdl->fwd_W: it follows the chunks that had to run again)
    unsigned long long* tot;   // [3] (hml_k_wide_lanes.h) wrong chunks of the filter, the sum of nfb, wrong chunks of the backward draws
};
#define HML_CHUNK_W_ADAPTIVE 0xffffffffu
// The warm-up of the chunks (results never depend on it: a chunk is accepted only on bit equality with what the chunk before it
// left).  Adaptive (round 5): it starts at 64 blocks, doubles when a sweep had chunks that ran again and falls by a quarter after
// sixteen sweeps without one, down to a floor of 32 (mdl->fwd_W / fwd_W0 / fwd_quiet, which the kernels of this family own in
// their modes) - models of 20-64 states on config 3's trace run no chunk again at 32 blocks, and 128 (round 4's constant beyond 16
// states) was 58 % of a chunk's steps.
__device__ __forceinline__ uint32_t hml_chunk_warmup(const hml_model* mdl, const uint32_t W) { return W == HML_CHUNK_W_ADAPTIVE ? mdl->fwd_W : W; }
__device__ __forceinline__ void hml_chunk_warmup_adapt(hml_model* mdl, const uint32_t W, unsigned long long ran_again, bool may_shrink) {
    if (W != HML_CHUNK_W_ADAPTIVE) return;
    uint32_t w = mdl->fwd_W;
    if (ran_again != 0ull) { w = (2u * w < 1024u) ? 2u * w : 1024u; mdl->fwd_quiet = 0u; }
    else if (may_shrink && ++mdl->fwd_quiet >= 16u) {
        const uint32_t lower = (w - w / 4u) & ~7u;
        w = lower > mdl->fwd_W0 ? lower : mdl->fwd_W0;
        mdl->fwd_quiet = 0u;
    }
    mdl->fwd_W = w;
}
// blocks per chunk of a launch of C chunks over B blocks
__device__ __forceinline__ uint32_t hml_compat_chunk_len(uint32_t B, uint32_t C) { const uint32_t L = (B + C - 1u) / C; return L < 64u ? 64u : L; }

// StateSequence<ForwardBackward>::sample's filter (ForwardBackward.hpp:86-123) over blocks [b_begin, b_end), rows stored from
// block b_store on.  rows: (B + 1) x K floats, row 0 = pi; row t = b + 1 in the form the backward pass reads it - rescaled by
// A(s, s)^(N_b - 1) (:115-119: the reference rescales a row once the next one exists; the same product) - except the last.
// prev: the row before block b_begin (in), behind block b_end - 1 (out, unscaled); entry_out: the row before block b_store.
// KC: the number of states as a compile-time value (2 .. 16: A's column in registers, loops unrolled) or 0 = the model's
// value (A in LDS).
template <int KC, bool PAD = false>
__device__ __forceinline__ void hml_compat_forward_range(int K, uint32_t B, bool self, const float* __restrict__ em, const float* __restrict__ g,
                                                         float* __restrict__ rows, float* __restrict__ aprobe, const float (&acol)[KC ? KC : 1],
                                                         const float* sA, float* tile, float* tile_g, uint32_t b_begin, uint32_t b_store, uint32_t b_end,
                                                         float& prev, float* entry_out, uint32_t& nfb, int lane) {
    const bool act = lane < K;
    const uint32_t TB = (uint32_t)((HML_COMPAT_TILE / K) < 64 ? (HML_COMPAT_TILE / K) : 64);
    constexpr int NREG = HML_COMPAT_TILE / 64;
    float nxt[NREG], nxt_g[NREG];
    auto fetch = [&](uint32_t b0) {   // a tile's emission terms and factors into registers (they travel during the tile before)
        const uint32_t nb = (b0 < b_end) ? ((b_end - b0 < TB) ? b_end - b0 : TB) : 0u;
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            const uint32_t i = (uint32_t)r * 64u + (uint32_t)lane;
            const bool in = i < nb * (uint32_t)K;
            nxt[r] = in ? em[(uint64_t)b0 * K + i] : 0.0f;
            nxt_g[r] = (in && self) ? g[(uint64_t)b0 * K + i] : 1.0f;
        }
    };
    fetch(b_begin);
    for (uint32_t b0 = b_begin; b0 < b_end; b0 += TB) {
        const uint32_t nb = (b_end - b0 < TB) ? b_end - b0 : TB;
        hml_compat_fence();
#pragma unroll
        for (int r = 0; r < NREG; ++r) { tile[r * 64 + lane] = nxt[r]; tile_g[r * 64 + lane] = nxt_g[r]; }
        hml_compat_fence();
        fetch(b0 + TB);
        float e_nx = act ? tile[lane] : 0.0f, g_nx = act ? tile_g[lane] : 1.0f;   // (one block ahead of the filter)
        for (uint32_t r = 0; r < nb; ++r) {
            const uint32_t b = b0 + r;
            const uint64_t t = (uint64_t)b + 1u;
            float f = e_nx;
            const float gcur = g_nx;
            if (r + 1u < nb) { e_nx = act ? tile[(r + 1u) * (uint32_t)K + (uint32_t)lane] : 0.0f; g_nx = act ? tile_g[(r + 1u) * (uint32_t)K + (uint32_t)lane] : 1.0f; }
            if (b == b_store && entry_out && act) entry_out[lane] = prev;
            float tt = 0.0f;
            if (KC && PAD) {   // (acol[i] = 0 and prev = 0 from K on: the padded terms are +0.0)
#pragma unroll
                for (int i0 = 0; i0 < (KC ? KC : 1); i0 += 4) {
                    if (i0 < K) {
#pragma unroll
                        for (int i = i0; i < i0 + 4; ++i) tt += hml_lane_f32(prev, i) * acol[i];
                    }
                }
            } else if (KC) {
#pragma unroll
                for (int i = 0; i < (KC ? KC : 1); ++i) tt += hml_lane_f32(prev, i) * acol[i];
            } else {
                for (int i = 0; i < K; ++i) tt += hml_lane_f32(prev, i) * sA[i * K + lane];   // (lanes >= K read inside the array; their values are not used)
            }
            f *= tt;
            float Z = 0.0f;
            if (KC && PAD) {
#pragma unroll
                for (int j0 = 0; j0 < (KC ? KC : 1); j0 += 4) {
                    if (j0 < K) {
#pragma unroll
                        for (int j = j0; j < j0 + 4; ++j) Z += hml_lane_f32(f, j);
                    }
                }
            } else if (KC) {
#pragma unroll
                for (int j = 0; j < (KC ? KC : 1); ++j) Z += hml_lane_f32(f, j);
            } else {
                for (int j = 0; j < K; ++j) Z += hml_lane_f32(f, j);
            }
            float fw;
            if (Z != 0.0f) fw = f / Z;
            else { if (b >= b_store) nfb++; fw = (float)(1.0 / (double)(float)K); }
            if (act && b >= b_store) {
                if (aprobe) aprobe[t * K + lane] = fw;
                rows[t * K + lane] = (self && t < (uint64_t)B) ? fw * gcur : fw;
            }
            prev = act ? fw : 0.0f;
        }
    }
}

template <int KC, class M = hml_glibc_exp, bool PAD = false>
HML_KERNEL __launch_bounds__(64) void hml_k_compat_forward(hml_model* __restrict__ mdl, const float* __restrict__ em, const float* __restrict__ g,
                                                           float* __restrict__ rows, float* __restrict__ aprobe, const hml_compat_chunks ch) {
    __shared__ float sA[KC ? 1 : HML_CAP_K * HML_CAP_K];   // row-major: lane j reads A(i, j)
    __shared__ float tile[HML_COMPAT_TILE], tile_g[HML_COMPAT_TILE];
    if (mdl->halted != 0u) return;
    const int lane = threadIdx.x;
    const int K = (KC && !PAD) ? KC : mdl->K;
    const uint32_t B = mdl->B;
    const uint32_t C = gridDim.x, c = blockIdx.x, L = hml_compat_chunk_len(B, C);
    const uint32_t lo = c * L;
    if (lane == 0) ch.bad[c] = 0u;
    if (lo >= B) return;
    const uint32_t hi = (lo + L < B) ? lo + L : B;
    const bool self = mdl->self_trans != 0;
    const bool act = lane < K;
    float acol[KC ? KC : 1];
    if (KC) {
#pragma unroll
        for (int i = 0; i < (KC ? KC : 1); ++i) acol[i] = (act && i < K) ? mdl->A[i * K + lane] : 0.0f;
    } else {
        for (int i = lane; i < K * K; i += 64) sA[i] = mdl->A[i];
    }
    const uint32_t W = hml_chunk_warmup(mdl, ch.W);
    const uint32_t ws = (lo > W) ? lo - W : 0u;
    float prev = act ? ((ws == 0u) ? mdl->pi[lane] : (float)(1.0 / (double)(float)K)) : 0.0f;
    if (c == 0u && act) {
        if (aprobe) aprobe[lane] = prev;
        // row 0 with the factor of a "block" of size 1 before the first (prevN = 1, ForwardBackward.hpp:107)
        rows[lane] = self ? prev * M::expf_((1.0f - 1.0f) * mdl->logA[lane]) : prev;
    }
    uint32_t nfb = 0u;
    hml_compat_forward_range<KC, PAD>(K, B, self, em, g, rows, aprobe, acol, sA, tile, tile_g, ws, lo, hi, prev, ch.entry + (uint64_t)c * K, nfb, lane);
    if (act) ch.exitv[(uint64_t)c * K + lane] = prev;
    if (lane == 0) ch.nfb[c] = nfb;
}

// which chunks started from another row than the chunk before them left: every element of every chunk's entry row against the
// exit row before it, bit for bit, by as many threads as there are elements (round 5: one wavefront used to walk the chunks)
HML_KERNEL __launch_bounds__(256) void hml_k_compat_forward_verify(const hml_model* __restrict__ mdl, const hml_compat_chunks ch, uint32_t C) {
    if (mdl->halted != 0u) return;
    const uint32_t B = mdl->B, K = (uint32_t)mdl->K;
    const uint32_t L = hml_compat_chunk_len(B, C);
    const uint32_t n_chunks = (B + L - 1u) / L;
    const uint64_t n = (uint64_t)n_chunks * K;
    const uint32_t W = hml_chunk_warmup(mdl, ch.W);
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)(e / K);
        // (a chunk whose warm-up reaches block 0 started from pi: exact)
        if (c > 0u && (uint64_t)c * L > W && hml_f2u(ch.entry[e]) != hml_f2u(ch.exitv[e - K])) ch.bad[c] = 1u;
    }
}

// the chunks' flags of a check kernel as a bit map in LDS (bit c of word c / 64), by all of its 256 threads; returns with a barrier
#define HML_COMPAT_MAP_WORDS (HML_COMPAT_MAX_CHUNKS / 64)
template <class F>
__device__ __forceinline__ void hml_compat_flag_map(unsigned long long* map, uint32_t n_chunks, int tid, F flag) {
    const int lane = tid & 63, wave = tid >> 6;
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += 256u) {   // workgroup-uniform
        const uint32_t cl = c0 + (uint32_t)tid;
        const unsigned long long m = __ballot(cl < n_chunks && flag(cl));
        if (lane == 0 && c0 + 64u * (uint32_t)wave < n_chunks) map[c0 / 64u + (uint32_t)wave] = m;
    }
    __syncthreads();
}
// bit of chunk c: cleared (the chunk was just compared with what its predecessor really left)
__device__ __forceinline__ void hml_compat_flag_clear(unsigned long long* map, uint32_t c, int lane) {
    if (lane == 0) map[c >> 6] &= ~(1ull << (c & 63u));
    hml_compat_fence();
}

// the filter's chunks in order: a chunk whose first row is not what the chunk before it left runs again from that row
template <int KC, bool PAD = false>
HML_KERNEL __launch_bounds__(256) void hml_k_compat_forward_check(hml_model* __restrict__ mdl, const float* __restrict__ em, const float* __restrict__ g,
                                                                  float* __restrict__ rows, const hml_compat_chunks ch, uint32_t C) {
    __shared__ float sA[KC ? 1 : HML_CAP_K * HML_CAP_K];
    __shared__ float tile[HML_COMPAT_TILE], tile_g[HML_COMPAT_TILE];
    __shared__ unsigned long long map[HML_COMPAT_MAP_WORDS];
    __shared__ unsigned long long s_nfb[4];
    if (mdl->halted != 0u) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = (KC && !PAD) ? KC : mdl->K;
    const uint32_t B = mdl->B;
    const uint32_t L = hml_compat_chunk_len(B, C);
    const uint32_t n_chunks = (B + L - 1u) / L;
    // the chunks' flags (hml_k_compat_forward_verify) and the sum of their uniform fallbacks, by all four wavefronts
    {
        unsigned long long mine = 0ull;
        for (uint32_t cl = (uint32_t)tid; cl < n_chunks; cl += 256u) mine += (unsigned long long)ch.nfb[cl];
        for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m);
        if (lane == 0) s_nfb[tid >> 6] = mine;
    }
    hml_compat_flag_map(map, n_chunks, tid, [&](uint32_t cl) { return ch.bad[cl] != 0u; });
    if (tid >= 64) return;
    const bool self = mdl->self_trans != 0;
    const bool act = lane < K;
    float acol[KC ? KC : 1];
    if (KC) {
#pragma unroll
        for (int i = 0; i < (KC ? KC : 1); ++i) acol[i] = (act && i < K) ? mdl->A[i * K + lane] : 0.0f;
    } else {
        for (int i = lane; i < K * K; i += 64) sA[i] = mdl->A[i];
    }
    hml_compat_fence();
    const uint32_t W = hml_chunk_warmup(mdl, ch.W);   // (the warm-up this sweep's chunks ran with: adapted at the very end)
    unsigned long long total_nfb = s_nfb[0] + s_nfb[1] + s_nfb[2] + s_nfb[3], redone = 0ull;
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64u) {
        while (true) {   // wave-uniform
            const unsigned long long todo = map[c0 >> 6];
            if (todo == 0ull) break;
            const uint32_t c = c0 + (uint32_t)(__ffsll((long long)todo) - 1);
            hml_compat_flag_clear(map, c, lane);
            // chunks c, c + 1, ... until one leaves what its successor started from
            for (uint32_t cc = c; cc < n_chunks; ++cc) {
                const uint32_t lo = cc * L, hi = (lo + L < B) ? lo + L : B;
                float prev = act ? ch.exitv[(uint64_t)(cc - 1u) * K + lane] : 0.0f;
                uint32_t nfb = 0u;
                const uint32_t old_nfb = ch.nfb[cc];
                hml_compat_forward_range<KC, PAD>(K, B, self, em, g, rows, nullptr, acol, sA, tile, tile_g, lo, lo, hi, prev, nullptr, nfb, lane);
                if (act) ch.exitv[(uint64_t)cc * K + lane] = prev;
                if (lane == 0) ch.nfb[cc] = nfb;
                total_nfb += (unsigned long long)nfb - (unsigned long long)old_nfb;
                redone++;
                if (cc + 1u >= n_chunks) break;
                hml_compat_flag_clear(map, cc + 1u, lane);   // (the successor is compared right here)
                const float nx = act ? ch.entry[(uint64_t)(cc + 1u) * K + lane] : 0.0f;
                const bool exact_next = (uint64_t)(cc + 1u) * L <= W;
                if (exact_next || __ballot(act && hml_f2u(nx) != hml_f2u(prev)) == 0ull) break;
            }
        }
    }
    if (lane == 0) { mdl->uniform_fallbacks += total_nfb; mdl->forward_refits += redone; hml_chunk_warmup_adapt(mdl, ch.W, redone, true); }
}

// the engine's next n outputs -> out[0 .. n): the uniforms of a sweep's categorical draws do not depend on the data.
// Round 5: one workgroup of FOUR wavefronts and a twist in three steps.  mersenne_twister_engine::_M_gen_rand replaces the
// state in place, k = 0 .. 623: x_k <- x_{k+397 mod 624} ^ f(x_k, x_{k+1 mod 624}).  For k < 227 all three operands are
// old values; for 227 <= k < 454 the first is the NEW x_{k-227} of the first range; for 454 <= k < 624 the new x_{k-227} of
// the second range (and x_623 also reads the new x_0).  So a twist is three ranges of at most 227 independent elements:
// one element per thread, a barrier between the ranges, old and new state in two LDS arrays (no read-before-write hazards
// inside a range) - 3 barriers per 624 outputs where round 4's single wavefront went through ten chunks of 64 with two
// fences each (1.4 us per twist, 1.1 ms of config 3's sweep).
#define HML_MT_M 397
__device__ __forceinline__ uint32_t hml_mt_mix(uint32_t hi, uint32_t lo, uint32_t far) {
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t hml_mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
// old -> nw (two arrays of 624 words in LDS), 256 threads; ends with a barrier.  The first `m` words of the new state are also
// the engine's next outputs: the thread that makes word k tempers it and stores it to out[k] (out may be null: m = 0).
// a barrier that waits for the workgroup's LDS traffic only: the tempered outputs go to memory behind it without being waited
// for (__syncthreads also waits for every outstanding store: a memory round trip per range of a twist)
#define HML_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__device__ __forceinline__ void hml_mt_twist_wg(const uint32_t* old, uint32_t* nw, int tid, uint32_t* __restrict__ out, uint32_t m) {
    constexpr int R = HML_MT_N - HML_MT_M;   // 227
    if (tid < R) {
        const uint32_t v = hml_mt_mix(old[tid], old[tid + 1], old[tid + HML_MT_M]);
        nw[tid] = v;
        if ((uint32_t)tid < m) out[tid] = hml_mt_temper(v);
    }
    HML_LDS_BARRIER();
    if (tid < R) {
        const int k = R + tid;
        const uint32_t v = hml_mt_mix(old[k], old[k + 1], nw[k - R]);
        nw[k] = v;
        if ((uint32_t)k < m) out[k] = hml_mt_temper(v);
    }
    HML_LDS_BARRIER();
    if (tid < HML_MT_N - 2 * R) {   // 170 elements
        const int k = 2 * R + tid;
        const uint32_t v = hml_mt_mix(old[k], (k == HML_MT_N - 1) ? nw[0] : old[k + 1], nw[k - R]);
        nw[k] = v;
        if ((uint32_t)k < m) out[k] = hml_mt_temper(v);
    }
    HML_LDS_BARRIER();
}
HML_KERNEL __launch_bounds__(256) void hml_k_compat_draws(hml_model* __restrict__ mdl, hml_mt_state* __restrict__ mts, uint32_t* __restrict__ out, uint32_t per_block) {
    __shared__ uint32_t st[2][HML_MT_N];
    if (mdl->halted != 0u) return;
    const int tid = threadIdx.x;
    const uint64_t n = (uint64_t)mdl->B * per_block;
    for (int i = tid; i < HML_MT_N; i += 256) st[0][i] = mts->mt[i];
    uint32_t idx = mts->idx;
    int cur = 0;
    __syncthreads();
    uint64_t o = 0;
    if (idx < HML_MT_N && n > 0) {   // what the current state still holds
        const uint32_t m = (n < (uint64_t)(HML_MT_N - idx)) ? (uint32_t)n : (uint32_t)(HML_MT_N - idx);
        for (uint32_t k = (uint32_t)tid; k < m; k += 256u) out[k] = hml_mt_temper(st[0][idx + k]);
        idx += m; o = m;
    }
    while (o < n) {   // workgroup-uniform: one twist per 624 outputs, tempered and stored as they are made
        const uint32_t m = (n - o < (uint64_t)HML_MT_N) ? (uint32_t)(n - o) : (uint32_t)HML_MT_N;
        hml_mt_twist_wg(st[cur], st[cur ^ 1], tid, out + o, m);
        cur ^= 1; idx = m; o += m;
    }
    __syncthreads();
    for (int i = tid; i < HML_MT_N; i += 256) mts->mt[i] = st[cur][i];
    if (tid == 0) mts->idx = idx;
}

// StateSequence<ForwardBackward>::sample's backward draws (ForwardBackward.hpp:133-162; Trellis::sample, Trellis.hpp:61-66) over
// rows t_begin, t_begin - 1, ..., t_end + 1 (row B: the weights are the row itself; below: rows[t][i] * A(i, q_{t+1})), the states
// stored from row t_store down.  j: the state above row t_begin (in), of row t_end + 1 (out).  draws: two engine outputs per row, row
// t at 2 (B - t).  Returns false if a weight was negative (the caller runs the rows again where it may raise: ForwardBackward.hpp:147-149).
template <int KC, bool RAISE, bool PAD = false>
__device__ __forceinline__ bool hml_compat_backward_range(hml_model* mdl, int K, uint32_t B, const float* __restrict__ rows, const uint32_t* __restrict__ draws,
                                                          int16_t* __restrict__ q, const float (&arow)[(KC && !PAD) ? KC : 1], const float* sAT, float* tile, uint32_t* tile_d,
                                                          int16_t* tile_q, uint32_t t_begin, uint32_t t_store, uint32_t t_end, int& j, int* in_out, int lane) {
    const bool act = lane < K;
    const uint32_t TB = (uint32_t)((HML_COMPAT_TILE / K) < 64 ? (HML_COMPAT_TILE / K) : 64);
    constexpr int NREG = HML_COMPAT_TILE / 64;
    bool clean = true;
    float nxt[NREG];
    uint32_t nxt_d[2];
    auto fetch = [&](uint32_t hi) {   // rows lo + 1 .. hi and their outputs into registers
        const uint32_t span = hi > t_end ? hi - t_end : 0u;
        const uint32_t nb = span < TB ? span : TB;
        const uint32_t lo = hi - nb;
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            const uint32_t i = (uint32_t)r * 64u + (uint32_t)lane;
            nxt[r] = (i < nb * (uint32_t)K) ? rows[(uint64_t)(lo + 1u) * K + i] : 0.0f;
        }
        // outputs of rows hi, hi - 1, ...: 2 (B - hi) + k, k < 2 nb (lane: k = lane, lane + 64)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t k = (uint32_t)r * 64u + (uint32_t)lane;
            nxt_d[r] = (k < 2u * nb) ? draws[2ull * (B - hi) + k] : 0u;
        }
    };
    fetch(t_begin);
    for (uint32_t hi = t_begin; hi > t_end; ) {
        const uint32_t nb = (hi - t_end) < TB ? (hi - t_end) : TB;
        const uint32_t lo = hi - nb;
        hml_compat_fence();
#pragma unroll
        for (int r = 0; r < NREG; ++r) tile[r * 64 + lane] = nxt[r];
        tile_d[lane] = nxt_d[0]; tile_d[64 + lane] = nxt_d[1];
        hml_compat_fence();
        fetch(lo);
        float row_nx = act ? tile[(nb - 1u) * (uint32_t)K + (uint32_t)lane] : 0.0f;   // (one row ahead of the draws)
        uint32_t d0 = tile_d[0], d1 = tile_d[1];
        for (uint32_t r = 0; r < nb; ++r) {
            const uint32_t t = hi - r;
            const float row = row_nx;
            const double u = hml_canonical_f64(d0, d1);
            if (r + 1u < nb) { row_nx = act ? tile[(nb - 2u - r) * (uint32_t)K + (uint32_t)lane] : 0.0f; d0 = tile_d[2u * r + 2u]; d1 = tile_d[2u * r + 3u]; }
            if (t == t_store && in_out && lane == 0) *in_out = j;
            float w;
            if (t == B) w = act ? row : 0.0f;   // (the last row: no check, like Trellis::sample)
            else {
                float a;
                if (KC && !PAD) {
                    a = arow[0];
#pragma unroll
                    for (int jj = 1; jj < (KC ? KC : 1); ++jj) a = (j == jj) ? arow[jj] : a;
                } else a = sAT[j * K + lane];
                w = act ? row * a : 0.0f;
                const unsigned long long neg = __ballot(w < 0.0f);
                if (neg != 0ull) {   // ForwardBackward.hpp:147-149 (the first negative weight in state order is the one reported)
                    clean = false;
                    if (RAISE) {
                        const int first = __ffsll((long long)neg) - 1;
                        if (lane == first) hml_raise(mdl, HML_DEVERR_NEG_BACKWARD, w);
                        hml_compat_fence();
                        if (w < 0.0f && lane != first) hml_raise(mdl, HML_DEVERR_NEG_BACKWARD, w);
                    }
                }
            }
            j = hml_compat_categorical_wave<KC, PAD>(w, K, u, lane);
            if (lane == 0) tile_q[r] = (int16_t)j;
        }
        hml_compat_fence();
        // q[t - 1] of row t = hi - lane, rows from t_store down
        if ((uint32_t)lane < nb && hi - (uint32_t)lane <= t_store) q[hi - 1u - (uint32_t)lane] = tile_q[lane];
        hi = lo;
    }
    return clean;
}

template <int KC, bool PAD = false>
HML_KERNEL __launch_bounds__(64) void hml_k_compat_backward(hml_model* __restrict__ mdl, const float* __restrict__ rows, const uint32_t* __restrict__ draws,
                                                            int16_t* __restrict__ q, const hml_compat_chunks ch) {
    constexpr bool REG = KC && !PAD;                        // A's row in registers (a cascade of selects finds A(lane, j): up to 16 states)
    __shared__ float sAT[REG ? 1 : HML_CAP_K * HML_CAP_K];   // transposed: lane i reads A(i, j) at [j * K + i]
    __shared__ float tile[HML_COMPAT_TILE];
    __shared__ uint32_t tile_d[128];
    __shared__ int16_t tile_q[64];
    if (mdl->halted != 0u) return;
    const int lane = threadIdx.x;
    const int K = REG ? KC : mdl->K;
    const uint32_t B = mdl->B;
    const uint32_t C = gridDim.x, c = blockIdx.x, L = hml_compat_chunk_len(B, C);
    if (c * L >= B) return;
    const uint32_t hi = B - c * L;                                   // rows hi .. lo + 1 (chunk 0 holds row B)
    const uint32_t lo = (hi > L) ? hi - L : 0u;
    const bool act = lane < K;
    float arow[REG ? KC : 1];   // A(lane, j)
    if (REG) {
#pragma unroll
        for (int j = 0; j < (REG ? KC : 1); ++j) arow[j] = act ? mdl->A[lane * K + j] : 0.0f;
    } else {
        for (int idx = lane; idx < K * K; idx += 64) { const int i = idx / K, j = idx - i * K; sAT[j * K + i] = mdl->A[idx]; }
    }
    const uint32_t W = hml_chunk_warmup(mdl, ch.W);
    const uint32_t tw = (hi + W < B) ? hi + W : B;   // (tw = B: the chain of draws from its true start)
    int j = 0, in_state = -1;
    const bool single = (C == 1u);
    bool clean;
    if (single) clean = hml_compat_backward_range<KC, true, PAD>(mdl, K, B, rows, draws, q, arow, sAT, tile, tile_d, tile_q, tw, hi, lo, j, &in_state, lane);
    else clean = hml_compat_backward_range<KC, false, PAD>(mdl, K, B, rows, draws, q, arow, sAT, tile, tile_d, tile_q, tw, hi, lo, j, &in_state, lane);
    if (lane == 0) {
        ch.out_state[c] = j;
        // (a chunk that met a negative weight is run again by the checking launch, which raises; -2 never equals a state)
        ch.in_state[c] = clean ? ((tw == B) ? -1 : in_state) : -2;
    }
}
// the chunks of the backward draws from the top: a chunk that started from another state than the chunk above it ended in (or met
// a negative weight) runs again from that state
template <int KC, bool PAD = false>
HML_KERNEL __launch_bounds__(256) void hml_k_compat_backward_check(hml_model* __restrict__ mdl, const float* __restrict__ rows, const uint32_t* __restrict__ draws,
                                                                   int16_t* __restrict__ q, const hml_compat_chunks ch, uint32_t C) {
    constexpr bool REG = KC && !PAD;
    __shared__ float sAT[REG ? 1 : HML_CAP_K * HML_CAP_K];
    __shared__ float tile[HML_COMPAT_TILE];
    __shared__ uint32_t tile_d[128];
    __shared__ int16_t tile_q[64];
    __shared__ unsigned long long map[HML_COMPAT_MAP_WORDS];
    if (mdl->halted != 0u) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = REG ? KC : mdl->K;
    const uint32_t B = mdl->B;
    const uint32_t L = hml_compat_chunk_len(B, C);
    const uint32_t n_chunks = (B + L - 1u) / L;
    hml_compat_flag_map(map, n_chunks, tid, [&](uint32_t cl) {
        if (cl == 0u) return false;
        const int in = ch.in_state[cl];
        return (in == -2) || (in >= 0 && in != ch.out_state[cl - 1u]);
    });
    if (tid >= 64) return;
    const bool act = lane < K;
    float arow[REG ? KC : 1];
    if (REG) {
#pragma unroll
        for (int j = 0; j < (REG ? KC : 1); ++j) arow[j] = act ? mdl->A[lane * K + j] : 0.0f;
    } else {
        for (int idx = lane; idx < K * K; idx += 64) { const int i = idx / K, j = idx - i * K; sAT[j * K + i] = mdl->A[idx]; }
    }
    hml_compat_fence();
    unsigned long long redone = 0ull;
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64u) {
        while (true) {   // wave-uniform
            const unsigned long long todo = map[c0 >> 6];
            if (todo == 0ull) break;
            const uint32_t c = c0 + (uint32_t)(__ffsll((long long)todo) - 1);
            hml_compat_flag_clear(map, c, lane);
            for (uint32_t cc = c; cc < n_chunks; ++cc) {
                const uint32_t hi = B - cc * L, lo = (hi > L) ? hi - L : 0u;
                int j = ch.out_state[cc - 1u];
                hml_compat_backward_range<KC, true, PAD>(mdl, K, B, rows, draws, q, arow, sAT, tile, tile_d, tile_q, hi, hi, lo, j, nullptr, lane);
                hml_compat_fence();
                if (lane == 0) ch.out_state[cc] = j;
                hml_compat_fence();
                redone++;
                if (cc + 1u >= n_chunks) break;
                hml_compat_flag_clear(map, cc + 1u, lane);   // (the successor is compared right here)
                const int in = ch.in_state[cc + 1u];
                if (in == -1 || in == j) break;
            }
        }
    }
    if (lane == 0) { mdl->forward_refits += redone; hml_chunk_warmup_adapt(mdl, ch.W, redone, false); }   // (the statistic counts chunks of either pass that ran again)
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

// ---- the count pass by STATE (round 4; univariate models).  What the reference accumulates in block order is, per state, a chain
// of its own - the Kahan sums of the state's parameter, its occupancy and its diagonal transition count (the two that go through a
// float) only ever see the blocks of that state, in block order; the off-diagonal transition counts are exact increments.  So the
// blocks are PARTITIONED by state, stably (per tile of 4096 blocks a histogram, a scan over the tiles, a ranked scatter), and one
// wavefront walks all lists at once, lane s its own: max_s(blocks of state s) steps instead of B.
#define HML_COMPAT_PART_TILE 4096
struct hml_compat_lists {
    uint32_t* tile_count;   // [tiles][K] blocks of state s in the tile, then their exclusive prefix over the tiles
    uint32_t* state_off;    // [2 (K + 1)] first list position of state s (a multiple of four), then from K + 1 on the length of its list
    unsigned long long* offdiag;   // [K * K] transitions between different states
    // per list position, one array per quantity (round 5: each of the four wavefronts of the walk reads only what its chain adds -
    // sixteen entries are one 64-byte run per lane and array where they were sixteen 16-byte structures):
    float* sx;              // the block's Sx
    float* sq;              // the block's Sxx
    uint32_t* n;            // its size (below 2^31: the host takes the walk in block order for longer traces) and, in bit 31, whether
                            // the block before it has the same state (prev_0 = 0, ForwardBackward.hpp:172)
    // (every state's list starts at a multiple of four entries and the arrays are 16-byte aligned: four entries are one load;
    // the arrays hold B + 4 K entries)
};
HML_KERNEL __launch_bounds__(256) void hml_k_compat_part_count(const hml_model* __restrict__ mdl, const int16_t* __restrict__ q, const hml_compat_lists pl) {
    __shared__ uint32_t h[HML_CAP_K];
    if (mdl->halted != 0u) return;
    const int K = mdl->K;
    const uint32_t B = mdl->B;
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < K * K; i += 256) pl.offdiag[i] = 0ull;
    // (the grid is sized from an earlier sweep's block count: tiles are dealt round robin, whatever B is)
    for (uint32_t tile = blockIdx.x; (uint64_t)tile * HML_COMPAT_PART_TILE < B; tile += gridDim.x) {   // workgroup-uniform
        const uint32_t b0 = tile * (uint32_t)HML_COMPAT_PART_TILE;
        if (threadIdx.x < (unsigned)K) h[threadIdx.x] = 0u;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < (uint32_t)HML_COMPAT_PART_TILE && b0 + i < B; i += 256u) atomicAdd(&h[q[b0 + i]], 1u);
        __syncthreads();
        if (threadIdx.x < (unsigned)K) pl.tile_count[(uint64_t)tile * K + threadIdx.x] = h[threadIdx.x];
        __syncthreads();
    }
}
HML_KERNEL __launch_bounds__(64) void hml_k_compat_part_scan(const hml_model* __restrict__ mdl, const hml_compat_lists pl) {
    if (mdl->halted != 0u) return;
    const int K = mdl->K, lane = threadIdx.x;
    const uint32_t B = mdl->B;
    const uint32_t tiles = (B + HML_COMPAT_PART_TILE - 1u) / HML_COMPAT_PART_TILE;
    uint32_t run = 0u;
    if (lane < K) {
        uint32_t nxt = pl.tile_count[lane];
        for (uint32_t t = 0; t < tiles; ++t) {
            const uint32_t v = nxt;
            if (t + 1u < tiles) nxt = pl.tile_count[(uint64_t)(t + 1u) * K + lane];
            pl.tile_count[(uint64_t)t * K + lane] = run;
            run += v;
        }
    }
    // first list position of every state: exclusive prefix of the totals over the states
    uint32_t incl = run;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    // (lists start at multiples of four entries: at most 3 K entries of padding - the exclusive prefix of the rounded lengths)
    const uint32_t padded = (lane < K) ? ((run + 3u) & ~3u) : 0u;
    uint32_t pincl = padded;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(pincl, d); if (lane >= d) pincl += o; }
    (void)incl;
    if (lane < K) { pl.state_off[lane] = pincl - padded; pl.state_off[K + 1 + lane] = run; }
    if (lane == K - 1) pl.state_off[K] = pincl;
}
HML_KERNEL __launch_bounds__(256) void hml_k_compat_part_scatter(const hml_model* __restrict__ mdl, const int16_t* __restrict__ q, const uint32_t* __restrict__ starts,
                                                                 const float2* __restrict__ bstat, const hml_compat_lists pl) {
    __shared__ uint32_t cnt[HML_CAP_K], pw[4][HML_CAP_K];
    if (mdl->halted != 0u) return;
    const int K = mdl->K;
    const uint32_t B = mdl->B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t tile = blockIdx.x; (uint64_t)tile * HML_COMPAT_PART_TILE < B; tile += gridDim.x) {   // workgroup-uniform
    const uint32_t b0 = tile * (uint32_t)HML_COMPAT_PART_TILE;
    __syncthreads();
    if (threadIdx.x < (unsigned)K) cnt[threadIdx.x] = pl.state_off[threadIdx.x] + pl.tile_count[(uint64_t)tile * K + threadIdx.x];
    for (int i = threadIdx.x; i < 4 * HML_CAP_K; i += 256) (&pw[0][0])[i] = 0u;
    __syncthreads();
    for (uint32_t step = 0; step < (uint32_t)HML_COMPAT_PART_TILE / 256u; ++step) {   // workgroup-uniform trip count
        const uint32_t b = b0 + step * 256u + threadIdx.x;
        const bool in = b < B;
        const int s = in ? (int)q[b] : -1;
        const int prev = (in && b > 0u) ? (int)q[b - 1u] : 0;
        // rank among the wavefront's blocks of the same state, in block order
        uint32_t rank = 0u;
        unsigned long long left = __ballot(in);
        while (left != 0ull) {   // wave-uniform: one trip per state present
            const int leader = __builtin_amdgcn_readlane(s, __ffsll((long long)left) - 1);
            const unsigned long long m = __ballot(s == leader);
            if (s == leader) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) pw[wave][leader] = (uint32_t)__popcll(m);
            left &= ~m;
        }
        __syncthreads();
        if (in) {
            uint32_t pos = cnt[s] + rank;
            for (int w2 = 0; w2 < wave; ++w2) pos += pw[w2][s];
            const float2 st = bstat[b];
            pl.sx[pos] = st.x; pl.sq[pos] = st.y; pl.n[pos] = (starts[b + 1u] - starts[b]) | (prev == s ? 0x80000000u : 0u);
            if (prev != s) atomicAdd(&pl.offdiag[prev * K + s], 1ull);
        }
        __syncthreads();
        if (threadIdx.x < (unsigned)K) {
            cnt[threadIdx.x] += pw[0][threadIdx.x] + pw[1][threadIdx.x] + pw[2][threadIdx.x] + pw[3][threadIdx.x];
            pw[0][threadIdx.x] = 0u; pw[1][threadIdx.x] = 0u; pw[2][threadIdx.x] = 0u; pw[3][threadIdx.x] = 0u;
        }
        __syncthreads();
    }
    }
}

// A lane's list (hml_compat_lists: `cnt` entries from `off` on, `off` a multiple of four) walked in batches of 16 entries - four
// 16-byte loads - with SEVEN batches of loads in flight ahead of the one being added: step(entry) in list order.  The walk is a
// chain of dependent operations per entry, and what it used to wait for was memory (round 5): round 4 loaded one batch ahead,
// every load behind its own `entry < count` test - divergent code, at whose end the compiler waits for all loads it issued - so
// each batch paid a whole memory round trip (2.7 us per 32 entries, 5 ms per sweep of config 3); with straight-line loads of
// single words three batches ahead it still waited (96 cycles per entry: the lists come from the other end of the chip, 2 us
// away).  The loads are straight-line code: the group of four is clamped into the lane's list instead of the load being skipped
// (a lane without a list reads the array's first group), only the ADDITIONS of a lane's last batch are predicated.
// `longest`: the longest list of the wavefront (wave-uniform trip count).
#define HML_WALK_NB 16
#define HML_WALK_DEPTH 8
__device__ __forceinline__ void hml_compat_walk_fetch(const uint4* __restrict__ a4, uint32_t last4, uint32_t b0, uint4 (&v)[HML_WALK_NB / 4]) {
#pragma unroll
    for (int q = 0; q < HML_WALK_NB / 4; ++q) {
        const uint32_t g4 = b0 / 4u + (uint32_t)q;
        v[q] = a4[g4 < last4 ? g4 : last4];
    }
}
template <class Step>
__device__ __forceinline__ void hml_compat_walk_batch(uint32_t cnt, uint32_t b0, const uint4 (&v)[HML_WALK_NB / 4], Step& step) {
    if (b0 + (uint32_t)HML_WALK_NB <= cnt) {
#pragma unroll
        for (int q = 0; q < HML_WALK_NB / 4; ++q) { step(v[q].x); step(v[q].y); step(v[q].z); step(v[q].w); }
    } else if (b0 < cnt) {
#pragma unroll
        for (int q = 0; q < HML_WALK_NB / 4; ++q) {
            if (b0 + 4u * q + 0u < cnt) step(v[q].x);
            if (b0 + 4u * q + 1u < cnt) step(v[q].y);
            if (b0 + 4u * q + 2u < cnt) step(v[q].z);
            if (b0 + 4u * q + 3u < cnt) step(v[q].w);
        }
    }
}
template <class Step>
__device__ __forceinline__ void hml_compat_walk(const uint32_t* __restrict__ a, uint32_t off, uint32_t cnt, uint32_t longest, Step step) {
    constexpr uint32_t NB = HML_WALK_NB;
    constexpr int DEPTH = HML_WALK_DEPTH;
    const uint4* __restrict__ a4 = reinterpret_cast<const uint4*>(a) + off / 4u;
    const uint32_t last4 = cnt ? (cnt - 1u) / 4u : 0u;
    uint4 ring[DEPTH][NB / 4];   // (indexed by constants only once the loops below are unrolled: registers)
#pragma unroll
    for (int u = 0; u < DEPTH - 1; ++u) hml_compat_walk_fetch(a4, last4, (uint32_t)u * NB, ring[u]);
    for (uint32_t base = 0; base < longest; base += (uint32_t)DEPTH * NB) {   // wave-uniform
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            hml_compat_walk_fetch(a4, last4, base + (uint32_t)(u + DEPTH - 1) * NB, ring[(u + DEPTH - 1) % DEPTH]);
            hml_compat_walk_batch(cnt, base + (uint32_t)u * NB, ring[u], step);
        }
    }
}

// count pass in block order (ForwardBackward.hpp:170-200 / Mixture.hpp:113-141), conjugate updates (Conjugate.hpp:121-168,
// 178-205), theta, pi, A (HMM.hpp:111-115), derived values.  The aggregators live in registers, lane = parameter (Kahan sums,
// term counts) or state (occupancy, the diagonal transition count - the two that go through a float); a block's state, size and
// statistics reach all lanes as scalars (v_readlane from the lane that loaded them), the lane concerned takes them.  The
// off-diagonal transition counts are plain integer increments in LDS.
HML_KERNEL __launch_bounds__(256) void hml_k_compat_update(hml_model* __restrict__ mdl, hml_mt_state* __restrict__ mts,
                                                          const uint32_t* __restrict__ starts, const float2* __restrict__ bstat,
                                                          const int16_t* __restrict__ q, int method, const hml_compat_lists pl, int by_state) {
    __shared__ uint32_t lmt[HML_MT_N];
    __shared__ unsigned long long s_trans[HML_CAP_K * HML_CAP_K], s_occ[HML_CAP_K], s_n[HML_CAP_K];
    __shared__ float s_ps[HML_CAP_K], s_pq[HML_CAP_K];
    __shared__ uint8_t s_map[HML_CAP_K][HML_MAX_D];
    if (mdl->halted != 0u) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // (64 threads in block order, 256 by state)
    const int K = mdl->K, P = mdl->P, D = mdl->D;
    const uint32_t B = mdl->B;
    const uint64_t dstride = mdl->stat_stride;        // block statistics: one plane per dimension
    if (wave == 0) {
        for (int i = lane; i < HML_MT_N; i += 64) lmt[i] = mts->mt[i];
        for (int i = lane; i < K * K; i += 64) s_trans[i] = 0ull;
        if (lane < K) for (int d = 0; d < HML_MAX_D; ++d) s_map[lane][d] = mdl->map[lane][d];
    }
    if (by_state) __syncthreads();
    else hml_compat_fence();
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
    if (by_state) {
        // the lists of hml_k_compat_part_scatter: lane s walks the blocks of state s in block order, 16 entries in flight - and the
        // four chains that a state's blocks feed (Kahan sum of Sx, of Sxx, diagonal count, occupancy) are independent of each
        // other: FOUR wavefronts (the launch has 256 threads in this mode), each walks all lists for one of them
        const uint32_t off = (lane < K) ? pl.state_off[lane] : 0u;
        const uint32_t cnt = (lane < K) ? pl.state_off[K + 1 + lane] : 0u;
        uint32_t longest = cnt;
        for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)longest, m); longest = o > longest ? o : longest; }
        if (wave == 0) {
            hml_compat_walk(reinterpret_cast<const uint32_t*>(pl.sx), off, cnt, longest, [&](uint32_t xb) { const float x = hml_u2f(xb); const float y = x - es, t = ps + y; es = (t - ps) - y; ps = t; });
            if (lane < K) s_ps[lane] = ps;
        } else if (wave == 1) {
            hml_compat_walk(reinterpret_cast<const uint32_t*>(pl.sq), off, cnt, longest, [&](uint32_t xb) { const float x = hml_u2f(xb); const float y = x - eq, t = pq + y; eq = (t - pq) - y; pq = t; });
            if (lane < K) s_pq[lane] = pq;
        } else if (wave == 2) {
            // A(s, s)'s count: `size_t += float` (ForwardBackward.hpp:183-187) - the sum goes through a float, then the block's
            // entering transition is an exact increment (it may land between two floats once the count exceeds 2^24, hence the
            // double).  fd = (float)count throughout: four dependent operations per block, no branch.
            if (method == 1) {   // (a mixture sweep counts exactly, Mixture.hpp:113-128)
                hml_compat_walk(pl.n, off, cnt, longest, [&](uint32_t nf) { diag += (double)((nf & 0x7fffffffu) - 1u) + ((nf >> 31) ? 1.0 : 0.0); });
            } else {
                float fd = 0.0f;
                hml_compat_walk(pl.n, off, cnt, longest, [&](uint32_t nf) {
                    diag = (double)(fd + ((float)(nf & 0x7fffffffu) - 1.0f)) + ((nf >> 31) ? 1.0 : 0.0);
                    fd = (float)diag;
                });
            }
            if (lane < K) s_trans[lane * K + lane] = (unsigned long long)diag;
        } else {
            // the occupancy: `size_t += float` is one float addition per block ((float)(size_t)x == x for the integers a float sum
            // can be: the round trip through the integer changes nothing)
            if (method == 1) {
                hml_compat_walk(pl.n, off, cnt, longest, [&](uint32_t nf) { const uint32_t n = nf & 0x7fffffffu; occ += (double)n; n_terms += n; });
            } else {
                float fo = 0.0f;
                hml_compat_walk(pl.n, off, cnt, longest, [&](uint32_t nf) { const uint32_t n = nf & 0x7fffffffu; fo = fo + (float)n; n_terms += n; });
                occ = (double)fo;
            }
            if (lane < K) { s_occ[lane] = (unsigned long long)occ; s_n[lane] = n_terms; }
        }
        for (int i = tid; i < K * K; i += 256) if (i / K != i % K) s_trans[i] = pl.offdiag[i];
        __syncthreads();
        if (tid != 0) return;
    } else {
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
    }
    if (!by_state) {
        hml_compat_fence();
        if (lane < K) { s_trans[lane * K + lane] = (unsigned long long)diag; s_occ[lane] = (unsigned long long)occ; s_n[lane] = n_terms; s_ps[lane] = ps; s_pq[lane] = pq; }
        hml_compat_fence();
        if (lane != 0) return;
    }
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
