// The default path for models of MORE THAN 16 STATES (17 .. HML_CAP_K = 64; the reference takes any `-s K`, src/main.cpp:112-137).
// The kernels of hml_k_forward.h / hml_k_backward.h keep a chunk's K-vector and the transition matrix in registers and pack a
// candidate map into 4-bit fields; they are instantiated for 2 .. 16 states.  Beyond that the number of states is a RUN-TIME
// value and a state is a LANE:
//   * filter and backward draws are the chunked lane-per-state kernels of hml_k_compat.h (chunks that start from a guess some
//     blocks early and are checked against each other, bit for bit: the stored rows and states are the sequential
//     recursion's) - instantiated with THIS path's arithmetic (hml_math.h's expf, hml_dev_exp) instead of glibc's and with their
//     loops unrolled over 32 or 64 states in groups of four; the emission terms have a kernel of their own (hml_k_wide_emission);
//   * the random decisions are the default path's: every backward row's uniform from its own Philox address
//     (hml_cat_uniform: blocks 2m and 2m + 1 share Philox block m), generated ahead of the draws by all lanes of the machine
//     (hml_k_wide_uniforms); every parameter variate from its own sub-stream (hml_k_wide_params);
//   * the count pass is the default path's fixed tree (D3: hml_k_counts' accumulators (group, wavefront, lane), pairwise over
//     the lanes, the wavefronts in order, pairwise over the 1024 groups) with exact integer counts (D4) - hml_k_wide_counts,
//     sixteen parameters per pass in registers - and the conjugate updates and draws of hml_k_params.h with the model's K
//     (hml_k_wide_params).
// Same chain as the CPU checker's device mode, bit for bit (tests/test_gpu_parity.py::test_sweeps_match_checker, 17 .. 64 states).
// Reference: src/StateSequence/ForwardBackward.hpp:16-213, src/StateSequence/Mixture.hpp:31-144, src/Conjugate.hpp:121-205,
// src/Distribution.hpp:77-178, src/Theta.hpp:203-234.
#ifndef HML_K_WIDE_H
#define HML_K_WIDE_H

#include "hml_k_compat.h"
#include "hml_k_params.h"

#if defined(__HIPCC__)

// the integer counts of a sweep: transitions [K][K], occupancies [K] (one slot: a workgroup adds its totals once)
struct hml_wide_acc {
    unsigned long long trans[HML_CAP_K * HML_CAP_K];
    unsigned long long occ[HML_CAP_K];
};

// the uniforms of a sweep's backward draws, where hml_k_compat_backward reads them: row t (1 .. B) at words 2 (B - t), 2 (B - t) + 1.
// Blocks 2m (row 2m + 1) and 2m + 1 (row 2m + 2) share Philox block m: words (0, 1) and (2, 3) (hml_cat_uniform, hml_dist.h).
HML_KERNEL __launch_bounds__(256) void hml_k_wide_uniforms(const hml_model* __restrict__ mdl, uint32_t* __restrict__ draws) {
    if (mdl->halted != 0u) return;
    const uint32_t B = mdl->B;
    const hml_key key = mdl->key;
    const unsigned long long epoch = mdl->epoch;
    const uint32_t pairs = (B + 1u) / 2u;
    for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < pairs; m += gridDim.x * blockDim.x) {
        const hml_u32x4 o = hml_stream4(key, HML_KIND_CAT, epoch, m, 0);
        const uint64_t at = 2ull * (uint64_t)(B - (2u * m + 1u));      // row 2m + 1
        draws[at] = o.v[0]; draws[at + 1u] = o.v[1];
        if (2u * m + 2u <= B) { draws[at - 2u] = o.v[2]; draws[at - 1u] = o.v[3]; }   // row 2m + 2
    }
}

// StateSequence<Mixture>::sample's draws (Mixture.hpp:90-112): one per block, block b from sub-stream (MIX, epoch, b)
HML_KERNEL __launch_bounds__(256) void hml_k_wide_mixture(const hml_model* __restrict__ mdl, const float* __restrict__ em, int16_t* __restrict__ q) {
    if (mdl->halted != 0u) return;
    const uint32_t B = mdl->B;
    const int K = mdl->K;
    const hml_key key = mdl->key;
    const unsigned long long epoch = mdl->epoch;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        const hml_u32x4 o = hml_stream4(key, HML_KIND_MIX, epoch, b, 0);
        q[b] = (int16_t)hml_categorical(em + (uint64_t)b * K, K, hml_canonical_f64(o.v[0], o.v[1]));
    }
}

// Emission terms of the blocks, [b][s] as hml_k_compat_forward reads them: em = expf(E_s - max E), g = expf((N - 1) log A(s, s))
// (ForwardBackward.hpp:67-84, EFD.hpp:23-38,83-93; method 1: Mixture.hpp:54-77, no self-transition term).  hml_k_compat_emission's
// values in this path's arithmetic, by a kernel of its own: a lane per block, the model's parameters and the expf table in
// LDS, the inner product through the double reciprocal (hml_inner_product: the division only where the product could round
// differently), and the K terms of a wavefront's 64 blocks through a tile in LDS so that they reach memory in whole lines
// (a lane's K terms lie K floats from its neighbour's: 1.5 10^6 blocks of 20 states took 0.96 ms the plain way).
HML_KERNEL __launch_bounds__(256) void hml_k_wide_emission(hml_model* __restrict__ mdl, const uint32_t* __restrict__ starts, const float2* __restrict__ bstat,
                                                           float* __restrict__ em, float* __restrict__ g, int method, float* __restrict__ eprobe) {
    __shared__ float s_mu[HML_CAP_K], s_var[HML_CAP_K], s_logNs[HML_CAP_K], s_logA[HML_CAP_K];
    __shared__ double s_rvar[HML_CAP_K];
    __shared__ uint8_t s_map[HML_CAP_K][HML_MAX_D];
    __shared__ uint64_t s_tab[32];
    __shared__ float tiles[4][64 * (HML_CAP_K + 1)];
    if (mdl->halted != 0u) return;
    const uint32_t B = mdl->B;
    const int K = mdl->K, D = mdl->D, P = mdl->P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool self = mdl->self_trans != 0 && method == 0;
    const uint64_t dstride = mdl->stat_stride;
    if (tid < P) { s_mu[tid] = mdl->mu[tid]; s_var[tid] = mdl->var[tid]; s_rvar[tid] = mdl->rvar2[tid]; }
    if (tid >= 64 && tid < 64 + K) {
        const int k = tid - 64;
        s_logNs[k] = mdl->logNs[k]; s_logA[k] = mdl->logA[k];
        for (int d = 0; d < HML_MAX_D; ++d) s_map[k][d] = mdl->map[k][d];
    }
    if (tid >= 128 && tid < 160) s_tab[tid - 128] = HML_EXP2F_TAB[tid - 128];
    __syncthreads();
    float* const tile = tiles[wave];
    const int pitch = K + 1;
    const uint32_t n_waves = gridDim.x * 4u;
    for (uint32_t b0 = (blockIdx.x * 4u + (uint32_t)wave) * 64u; b0 < B; b0 += n_waves * 64u) {   // wave-uniform
        const uint32_t b = b0 + (uint32_t)lane;
        const bool in = b < B;
        float N = 1.0f;
        float sx[HML_MAX_D], sq[HML_MAX_D];
#pragma unroll
        for (int d = 0; d < HML_MAX_D; ++d) { sx[d] = 0.0f; sq[d] = 0.0f; }
        if (in) {
            N = (float)(starts[b + 1u] - starts[b]);   // (size_t N, converted where it meets a float)
#pragma unroll
            for (int d = 0; d < HML_MAX_D; ++d) if (d < D) { const float2 v = bstat[(uint64_t)d * dstride + b]; sx[d] = v.x; sq[d] = v.y; }
        }
        float maxE = -3.40282346638528859812e+38f;
        for (int st = 0; st < K; ++st) {
            float r = 0.0f;   // innerProduct(y, theta.value(), theta.mapping(s)): float sum over the dimensions from 0 (EFD.hpp:83-93)
#pragma unroll
            for (int d = 0; d < HML_MAX_D; ++d) {
                if (d < D) {
                    const int pp = s_map[st][d];
                    const float ip = hml_inner_product(s_mu[pp], s_var[pp], s_rvar[pp], sx[d], sq[d]);
                    if (in && !hml_isfinite(ip)) hml_raise(mdl, HML_DEVERR_IP_NOT_FINITE, ip);
                    r += ip;
                }
            }
            float E = r - N * s_logNs[st];
            if (self) E += (N - 1.0f) * s_logA[st];
            tile[lane * pitch + st] = E;
            maxE = (E < maxE) ? maxE : E;
            if (eprobe && in) eprobe[(uint64_t)b * K + st] = E;
        }
        for (int st = 0; st < K; ++st) tile[lane * pitch + st] = hml_expf_tab(tile[lane * pitch + st] - maxE, s_tab);
        hml_compat_fence();
        const uint32_t nb = (B - b0 < 64u) ? B - b0 : 64u;
        const uint32_t n_out = nb * (uint32_t)K;
        auto write_out = [&](float* __restrict__ dst) {   // the tile's rows one after the other: consecutive floats of dst
            uint32_t blk = (uint32_t)lane / (uint32_t)K, st = (uint32_t)lane - blk * (uint32_t)K;
            const uint32_t dq = 64u / (uint32_t)K, dr = 64u - dq * (uint32_t)K;
            for (uint32_t i = (uint32_t)lane; i < n_out; i += 64u) {
                dst[(uint64_t)b0 * K + i] = tile[blk * (uint32_t)pitch + st];
                blk += dq; st += dr;
                if (st >= (uint32_t)K) { st -= (uint32_t)K; blk += 1u; }
            }
        };
        write_out(em);
        if (self) {
            hml_compat_fence();
            for (int st = 0; st < K; ++st) tile[lane * pitch + st] = hml_expf_tab((N - 1.0f) * s_logA[st], s_tab);
            hml_compat_fence();
            write_out(g);
        }
        hml_compat_fence();
    }
}

// K8 with the model's K.  The same accumulators and the same tree as hml_b_counts (hml_k_backward.h): block b
// belongs to chunk c = b / 256 and group g = c mod 1024, thread (wavefront, lane) = b mod 256 adds its blocks' terms in block
// order, pairwise tree over the lanes, the four wavefronts in order; partial[(2 p + {0, 1}) * 1024 + g].  A thread's
// accumulators live in registers sixteen parameters at a time (one pass over the group's chunks per sixteen: strongly
// compressed sweeps have one chunk per group, and the statistics of a long sweep stay in the caches between the passes); the
// integer counts are exact and go through LDS in the first pass.
#define HML_WIDE_PASS 16
HML_KERNEL __launch_bounds__(256) void hml_k_wide_counts(const int16_t* __restrict__ q, const uint32_t* __restrict__ starts, const float2* __restrict__ bstat,
                                                         hml_model* __restrict__ mdl, double* __restrict__ partial, hml_wide_acc* __restrict__ acc) {
    __shared__ unsigned long long h_trans[HML_CAP_K * HML_CAP_K];
    __shared__ unsigned long long h_occ[HML_CAP_K];
    __shared__ double wsum[4][HML_WIDE_PASS][2];
    __shared__ uint8_t s_map[HML_CAP_K][HML_MAX_D];
    if (mdl->halted != 0u) return;
    const uint32_t B = mdl->B;
    const int K = mdl->K, P = mdl->P, D = mdl->D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t g = blockIdx.x;
    const uint32_t nchunks = (B + HML_REDUCE_CHUNK - 1u) / HML_REDUCE_CHUNK;
    if (g >= nchunks) {   // a group without a chunk: the zeros its tree would produce (workgroup-uniform, before any barrier)
        for (int i = tid; i < 2 * P; i += 256) partial[(uint64_t)i * HML_REDUCE_GROUPS + g] = 0.0;
        return;
    }
    for (int i = tid; i < K * K; i += 256) h_trans[i] = 0ull;
    if (tid < K) { h_occ[tid] = 0ull; for (int d = 0; d < HML_MAX_D; ++d) s_map[tid][d] = mdl->map[tid][d]; }
    __syncthreads();
    const uint64_t dstride = mdl->stat_stride;
    for (int p0 = 0; p0 < P; p0 += HML_WIDE_PASS) {   // workgroup-uniform
        double acc_s[HML_WIDE_PASS], acc_q[HML_WIDE_PASS];
#pragma unroll
        for (int i = 0; i < HML_WIDE_PASS; ++i) { acc_s[i] = 0.0; acc_q[i] = 0.0; }
        if (D == 1) {
            // (the group's next chunk is asked for before this one is used - a chunk was a chain of three memory round trips, five
            // or six chunks a group and pass; lanes beyond the last block read it and add nothing: no load waits in a branch)
            struct item { int st, prev; uint32_t n; float2 v; bool in; };
            auto fetch = [&](const uint32_t c) {
                item it;
                const uint64_t b = (uint64_t)c * HML_REDUCE_CHUNK + (uint32_t)tid;
                it.in = c < nchunks && b < (uint64_t)B;
                const uint32_t bl = it.in ? (uint32_t)b : B - 1u;
                it.st = q[bl];
                it.v = bstat[bl];
                it.prev = 0; it.n = 0u;
                if (p0 == 0) {   // (workgroup-uniform)
                    const int before = (int)q[bl ? bl - 1u : 0u];
                    it.prev = bl ? before : 0;
                    it.n = starts[bl + 1u] - starts[bl];
                }
                return it;
            };
            item cur = fetch(g);
            for (uint32_t c = g; c < nchunks; c += HML_REDUCE_GROUPS) {
                const item nxt = fetch(c + HML_REDUCE_GROUPS);
                if (cur.in) {
                    if (p0 == 0) {
                        atomicAdd(&h_trans[cur.st * K + cur.st], (unsigned long long)(cur.n - 1u));
                        atomicAdd(&h_trans[cur.prev * K + cur.st], 1ull);
                        atomicAdd(&h_occ[cur.st], (unsigned long long)cur.n);
                    }
                    const int i = cur.st - p0;
                    if (i >= 0 && i < HML_WIDE_PASS) {
                        const double vx = (double)cur.v.x, vq = (double)cur.v.y;
#pragma unroll
                        for (int k = 0; k < HML_WIDE_PASS; ++k) if (i == k) { acc_s[k] = acc_s[k] + vx; acc_q[k] = acc_q[k] + vq; }
                    }
                }
                cur = nxt;
            }
        } else {
        for (uint32_t c = g; c < nchunks; c += HML_REDUCE_GROUPS) {
            const uint32_t b = c * HML_REDUCE_CHUNK + (uint32_t)tid;
            if (b < B) {
                const int st = q[b];
                if (p0 == 0) {
                    const int prev = (b == 0u) ? 0 : (int)q[b - 1u];
                    const uint32_t n = starts[b + 1u] - starts[b];
                    atomicAdd(&h_trans[st * K + st], (unsigned long long)(n - 1u));
                    atomicAdd(&h_trans[prev * K + st], 1ull);
                    atomicAdd(&h_occ[st], (unsigned long long)n);
                }
                {
                    // the term for parameter p: the block's statistics of the dimensions mapped to p, added in dimension order
#pragma unroll
                    for (int k = 0; k < HML_WIDE_PASS; ++k) {
                        double ts = 0.0, tq = 0.0;
                        bool any = false;
                        for (int dd = 0; dd < D; ++dd) {
                            if ((int)s_map[st][dd] == p0 + k) {
                                const float2 v2 = bstat[(uint64_t)dd * dstride + b];
                                ts = ts + (double)v2.x; tq = tq + (double)v2.y; any = true;
                            }
                        }
                        if (any) { acc_s[k] = acc_s[k] + ts; acc_q[k] = acc_q[k] + tq; }
                    }
                }
            }
        }
        }
        __syncthreads();   // (wsum of the pass before has been read)
#pragma unroll
        for (int k = 0; k < HML_WIDE_PASS; ++k) {
            const double a = hml_wave_tree_f64(acc_s[k]), d = hml_wave_tree_f64(acc_q[k]);
            if (lane == 0) { wsum[wave][k][0] = a; wsum[wave][k][1] = d; }
        }
        __syncthreads();
        if (tid < 2 * HML_WIDE_PASS && p0 + (tid >> 1) < P) {
            const int k = tid >> 1, cc = tid & 1;
            double v = 0.0;
            for (int wv = 0; wv < 4; ++wv) v = v + wsum[wv][k][cc];
            partial[(uint64_t)((p0 + k) * 2 + cc) * HML_REDUCE_GROUPS + g] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < K * K; i += 256) if (h_trans[i]) atomicAdd(&acc->trans[i], h_trans[i]);
    if (tid < K && h_occ[tid]) atomicAdd(&acc->occ[tid], h_occ[tid]);
}

// logNormalizer, log A(s, s), threshold of the current parameters (hml_derive of hml_k_params.h with the model's K)
__device__ __forceinline__ void hml_wide_derive(hml_model* mdl, int tid, int nthreads) {
    const int K = mdl->K, P = mdl->P, D = mdl->D;
    for (int k = tid; k < P; k += nthreads) {
        const float m = mdl->mu[k], v = mdl->var[k], sd = mdl->sd[k];
        mdl->logN[k] = hml_logf(sd) + m * m / (2 * v);
        mdl->rvar2[k] = 1.0 / (2.0 * (double)v);
    }
    for (int s = tid; s < K; s += nthreads) {
        float r = 0.0f;   // theta.logNormalizer(state): float sum over the state's parameters, in dimension order (Theta.hpp:148-158)
        for (int d = 0; d < D; ++d) {
            const int pp = mdl->map[s][d];
            const float m = mdl->mu[pp], v = mdl->var[pp], sd = mdl->sd[pp];
            r += hml_logf(sd) + m * m / (2 * v);
        }
        mdl->logNs[s] = r;
        mdl->logA[s] = hml_logf(mdl->A[s * K + s]);
    }
    if (tid == 0) {
        float mv = HML_INF_F;
        for (int k = 0; k < P; ++k) { const float v = mdl->var[k]; mv = (v < mv) ? v : mv; }   // std::min(result, var)
        const float l = hml_logf((float)mdl->T);
        const float arg = 2 * l * mv;
        const float t = HML_SQRTF(arg);
        mdl->thr_theta = t;
        if (mdl->dynamic) mdl->thr = t;
    }
}
HML_KERNEL __launch_bounds__(64) void hml_k_wide_derive(hml_model* mdl) { hml_wide_derive(mdl, threadIdx.x, 64); }

// K9 with the model's K: the tree over the count pass's group partials, conjugate updates (Conjugate.hpp:121-168,178-205), theta
// (Distribution.hpp:77-87), pi and the rows of A (Distribution.hpp:116-178) - every variate from its own Philox sub-stream, as
// hml_k_params.h draws them - posteriors back to the priors, derived values.  mode 0: after a sweep; 1: from the priors
// (main.cpp:393-401); 2: Theta's constructor draw (theta only).  One workgroup of 1024 threads.
HML_KERNEL __launch_bounds__(1024) void hml_k_wide_params(hml_model* __restrict__ mdl, const double* __restrict__ partial, hml_wide_acc* __restrict__ acc, int mode) {
    __shared__ double wp[16][2 * HML_CAP_K];
    __shared__ float fin[2 * HML_CAP_K];
    __shared__ float graw[HML_CAP_K * HML_CAP_K];
    __shared__ float praw[HML_CAP_K];
    __shared__ unsigned long long s_occ[HML_CAP_K];
    if (mode == 0 && mdl->halted != 0u) return;   // (the sweep did not happen: hml_state.h)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = mdl->K, P = mdl->P, D = mdl->D;
    const unsigned long long epoch = mdl->epoch;
    const hml_key key = mdl->key;
    if (tid < K) {
        unsigned long long o = 0ull;
        if (mode == 0) { o = acc->occ[tid]; acc->occ[tid] = 0ull; mdl->last_occ[tid] = o; }
        s_occ[tid] = o;
    }
    if (mode == 0) {
        // the fixed tree over the 1024 group partials of every statistic: pairwise inside each run of 64 (wavefront w: run w) ...
        for (int i0 = 0; i0 < 2 * P; i0 += 4) {
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (i0 + j < 2 * P) ? partial[(uint64_t)(i0 + j) * HML_REDUCE_GROUPS + tid] : 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double r = hml_wave_tree_f64(v[j]);
                if (lane == 0 && i0 + j < 2 * P) wp[wave][i0 + j] = r;
            }
        }
        __syncthreads();
        // ... then pairwise over the 16 runs
        if (tid < 2 * P) {
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = wp[i][tid];
#pragma unroll
            for (int st = 1; st < 16; st <<= 1)
#pragma unroll
                for (int i = 0; i < 16; i += 2 * st) v[i] = v[i] + v[i + st];
            fin[tid] = (float)v[0];
        }
    }
    __syncthreads();
    // ---- the variates: P x theta, then (mode != 2) K x pi, K x K x A, dealt over the threads
    const int n_items = P + (mode != 2 ? K + K * K : 0);
    for (int item = tid; item < n_items; item += 1024) {
        if (item < P) {
            const int k = item;
            float alpha = mdl->nig_post[k][0], beta = mdl->nig_post[k][1], mu0 = mdl->nig_post[k][2], nu = mdl->nig_post[k][3];
            unsigned long long cnt = 0ull;   // positions of every (state, dimension) mapped to parameter k
            if (mode == 0) {
                if (D == 1) cnt = s_occ[k];
                else for (int st = 0; st < K; ++st) for (int d = 0; d < D; ++d) if (mdl->map[st][d] == k) cnt += s_occ[st];
                mdl->last_sum[k] = fin[2 * k]; mdl->last_sumsq[k] = fin[2 * k + 1];
            }
            if (cnt > 0ull) {
                // Conjugate<NormalInverseGammaParam>::addObservation (Conjugate.hpp:121-168)
                const float sum = fin[2 * k], sumSq = fin[2 * k + 1];
                if (sumSq < 0.0f) hml_raise(mdl, HML_DEVERR_NEG_SUMSQ, sumSq);
                const double N = (double)cnt;
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
                alpha = na; beta = nb; mu0 = nm; nu = nn;
            }
            // Distribution<NormalInverseGamma>::resample (Distribution.hpp:77-87): gamma(alpha, 1 / beta), then the mean
            hml_dev_src src;
            src.s = hml_stream_open(key, HML_KIND_THETA, epoch, (uint32_t)k);
            const float gv = hml_gamma_f32<hml_devmath>(src, alpha, (float)(1.0 / (double)beta));
            const float v = (float)(1.0 / (double)gv);
            hml_normal_f32<hml_devmath> nd;
            const float m = nd.draw(src, mu0, HML_SQRTF(v / nu));
            if (!hml_isfinite(m)) hml_raise(mdl, HML_DEVERR_MEAN_NOT_FINITE, m);
            if (!hml_isfinite(v)) hml_raise(mdl, HML_DEVERR_VAR_NOT_FINITE, v);
            else if (v <= 0.0f) hml_raise(mdl, HML_DEVERR_VAR_NOT_POSITIVE, v);
            mdl->mu[k] = m; mdl->var[k] = v; mdl->sd[k] = HML_SQRTF(v);
            for (int i = 0; i < 4; ++i) mdl->nig_post[k][i] = mdl->nig_prior[i];
        } else if (item < P + K) {
            const int k = item - P;
            const float al = mdl->dirPi[k] + (float)s_occ[k];
            hml_dev_src src;
            src.s = hml_stream_open(key, HML_KIND_PI, epoch, (uint32_t)k);
            praw[k] = hml_gamma_f32<hml_devmath>(src, al, 1.0f);
            mdl->dirPi[k] = mdl->pi_alpha;
        } else {
            const int e = item - P - K;
            unsigned long long t = 0ull;
            if (mode == 0) { t = acc->trans[e]; acc->trans[e] = 0ull; mdl->last_trans[e] = t; }
            const float al = mdl->dirA[e] + (float)t;
            hml_dev_src src;
            src.s = hml_stream_open(key, HML_KIND_TRANS, epoch, (uint32_t)e);
            graw[e] = hml_gamma_f32<hml_devmath>(src, al, 1.0f);
            mdl->dirA[e] = (e / K == e % K) ? mdl->a_diag : mdl->a_off;
        }
    }
    __syncthreads();
    if (mode != 2) {
        // dirichlet_sample's normalisation (Distribution.hpp:116-139): float sum in index order, then the quotients
        if (tid < K) {
            float sum = 0.0f;
            for (int d = 0; d < K; ++d) sum += graw[tid * K + d];
            for (int d = 0; d < K; ++d) mdl->A[tid * K + d] = graw[tid * K + d] / sum;
        }
        if (tid == 64) {
            float sum = 0.0f;
            for (int d = 0; d < K; ++d) sum += praw[d];
            for (int d = 0; d < K; ++d) mdl->pi[d] = praw[d] / sum;
        }
    }
    __threadfence_block();
    __syncthreads();
    hml_wide_derive(mdl, tid, 1024);
    if (tid == 1023) {
        mdl->epoch = epoch + 1ull;
        if (mode == 0) { mdl->sweeps += 1ull; mdl->block_updates += (unsigned long long)mdl->B; }
    }
}

#endif
#endif
