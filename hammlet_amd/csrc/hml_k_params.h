// Conjugate parameter resampling: the K-sized step between two sweeps.
#ifndef HML_K_PARAMS_H
#define HML_K_PARAMS_H

#include <type_traits>

#include "hml_dist.h"
#include "hml_k_backward.h"
#include "hml_state.h"

struct hml_dev_src {
    hml_stream s;
    __device__ __forceinline__ uint32_t next() { return hml_stream_next(&s); }
};

// logNormalizer / logA / threshold from the current parameters (EFD.hpp:35-38,
// ForwardBackward.hpp:47-52, BreakpointArray.hpp:196-199 + Theta.hpp:227-234)
template <int K>
__device__ __forceinline__ void hml_derive(hml_model* mdl, int tid) {
    const int P = mdl->P, D = mdl->D;
    if (tid < P) {
        const float m = mdl->mu[tid], v = mdl->var[tid], sd = mdl->sd[tid];
        mdl->logN[tid] = hml_logf(sd) + m * m / (2 * v);
        mdl->rvar2[tid] = 1.0 / (2.0 * (double)v);
    }
    if (tid < K) {
        // theta.logNormalizer(state): float sum over the state's parameters, in dimension order (Theta.hpp:148-158)
        float r = 0.0f;
        for (int d = 0; d < D; ++d) {
            const int pp = mdl->map[tid][d];
            const float m = mdl->mu[pp], v = mdl->var[pp], sd = mdl->sd[pp];
            r += hml_logf(sd) + m * m / (2 * v);
        }
        mdl->logNs[tid] = r;
        mdl->logA[tid] = hml_logf(mdl->A[tid * K + tid]);
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

template <int K>
HML_KERNEL __launch_bounds__(64) void hml_k_derive(hml_model* mdl) { hml_derive<K>(mdl, threadIdx.x); }

HML_KERNEL __launch_bounds__(64) void hml_k_set_dynamic(hml_model* mdl, int on, int take_threshold) {
    if (threadIdx.x == 0) {
        mdl->dynamic = on;
        if (take_threshold) mdl->thr = mdl->thr_theta;
    }
}

// ------------------------------------------------------------------------------------------
// K9 params_resample - posterior updates and draws (reference src/Conjugate.hpp:121-168,178-205;
// src/Distribution.hpp:77-87,116-178; Theta::sample src/Theta.hpp:203-211; Initial::sample
// src/Initial.hpp:34-40; Transitions::sample src/Transitions.hpp:75-79), then posteriors reset to
// the priors.  mode 0: after a sweep (also finishes the count pass's fixed summation tree);
// mode 1: draw from the priors (src/main.cpp:393-401); mode 2: Theta's constructor draw (theta only).
// One workgroup of 1024: wavefront 0 draws theta, wavefront 1 pi, wavefronts 2.. the K*K entries
// of A, every variate from its own Philox sub-stream.
// ------------------------------------------------------------------------------------------
// SPREAD (round 4; mode 0 only, launched with gridDim.x = HML_PARAMS_TREE_WGS = 16): the first level of the count pass's tree
// - 1024 group partials per statistic, pairwise inside each run of 64 - is spread over sixteen workgroups.  Workgroup w reduces
// groups 64 w .. 64 w + 63 of every statistic (the pairwise tree that wavefront w of the one-workgroup form runs) and leaves
// 2 K doubles behind the group partials; its last wavefront - which draws nothing - takes a ticket while the others draw, and
// the workgroup that arrives last of the sixteen is the one that goes on: it reads 16 x 2 K doubles instead of 1024 x 2 K
// through one compute unit's memory pipeline (160 KB at K = 10: the tree was 7 us of a 14 us kernel; several chains at once
// stretch it further).  Every workgroup makes the draws that depend on the counts only - nothing outside the workgroup is
// written before the ticket is known - so none of this lies on the path of the one that goes on.  Same summation tree, same
// bits (D3 fixes the order, not who adds).  Config 4's sweep (10 states) 0.0938 -> 0.0869 ms, eight chains of config 3 0.161 ->
// 0.154 ms per round; config 3's single chain gains nothing (0.0557 / 0.0564) and keeps the one-workgroup form.
template <int K, bool SPREAD, bool MANY = false>
__device__ __forceinline__ void hml_b_params(hml_model* __restrict__ mdl, typename std::conditional<SPREAD, double, const double>::type* __restrict__ partial,
                                                     int mode, int leaf, int nleaf) {
    __shared__ double wp[16][K][2];
    __shared__ uint32_t s_last;
    __shared__ float fin[K][2];
    __shared__ float graw[K * K];
    __shared__ float praw[K];
    __shared__ unsigned long long s_trans[K * K];
    __shared__ unsigned long long s_occ[K];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a halted chain (the sweep's enumeration found more blocks than its buffers hold, hml_state.h): the sweep did not happen
    if (mode == 0 && mdl->halted != 0u) return;   // workgroup-uniform, before any barrier
    if (tid == 0) mdl->dbg_t[0] = wall_clock64();
    __shared__ float s_var[K], s_logN[K];
    const int P = mdl->P, D = mdl->D;
    const unsigned long long epoch = mdl->epoch;
    const hml_key key = mdl->key;
    // hyperparameters of this thread's variate, requested before the reductions below need the memory pipeline
    float hyp0 = 0.0f, hyp1 = 0.0f, hyp2 = 0.0f, hyp3 = 0.0f;
    if (wave == 0 && lane < P) { hyp0 = mdl->nig_post[lane][0]; hyp1 = mdl->nig_post[lane][1]; hyp2 = mdl->nig_post[lane][2]; hyp3 = mdl->nig_post[lane][3]; }
    if (wave == 1 && lane < K) hyp0 = mdl->dirPi[lane];
    if (tid >= 128 && tid < 128 + K * K) hyp0 = mdl->dirA[tid - 128];
    // gather the split integer accumulators (SPREAD: every workgroup reads them, the one that goes on resets them - below, once it
    // knows it is the one; each workgroup has read them before it takes its ticket)
    unsigned long long acc_total = 0ull;
    if (tid >= 128 && tid < 128 + K * K) {
        const int e = tid - 128;
        unsigned long long v[HML_CNT_SPLIT];
#pragma unroll
        for (int sp = 0; sp < HML_CNT_SPLIT; ++sp) v[sp] = mdl->trans[sp][e];   // all loads in flight together
        unsigned long long t = 0ull;
#pragma unroll
        for (int sp = 0; sp < HML_CNT_SPLIT; ++sp) { t += v[sp]; if (!SPREAD) mdl->trans[sp][e] = 0ull; }
        s_trans[e] = t;
        if (!SPREAD) mdl->last_trans[e] = t;
        acc_total = t;
    }
    if (tid >= 512 && tid < 512 + K) {
        const int k = tid - 512;
        unsigned long long v[HML_CNT_SPLIT];
#pragma unroll
        for (int sp = 0; sp < HML_CNT_SPLIT; ++sp) v[sp] = mdl->occ[sp][k];
        unsigned long long t = 0ull;
#pragma unroll
        for (int sp = 0; sp < HML_CNT_SPLIT; ++sp) { t += v[sp]; if (!SPREAD) mdl->occ[sp][k] = 0ull; }
        s_occ[k] = t;
        if (!SPREAD) mdl->last_occ[k] = t;
        acc_total = t;
    }
    // SPREAD: this workgroup's share of the tree's first level - wavefront v takes statistics v, v + 16 - left behind the group partials
    typename std::conditional<SPREAD, double, const double>::type* const level1 = partial + (uint64_t)HML_REDUCE_GROUPS * 2 * K;   // [16][2 K]
    if constexpr (SPREAD) {
        for (int idx = wave; idx < 2 * K; idx += 16) {
            const double r = hml_wave_tree_f64(partial[(uint64_t)idx * HML_REDUCE_GROUPS + 64u * (uint32_t)leaf + (uint32_t)lane]);
            if (lane == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(level1) + leaf * 2 * K + idx, hml_d2u(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // this thread's group partials of the count pass (the tree below), requested behind the counts - which the first draws
    // wait for - and travelling while those draws run
    // The wavefronts that draw variates (0: theta, 1: pi, 2..: the K * K entries of A) leave their 64 groups of the tree to
    // the last wavefronts, which take a second set: drawing and summing then run side by side.
    constexpr int NDRAW = 2 + (K * K + 63) / 64;
    static_assert(2 * NDRAW <= 16, "wavefronts of the parameter kernel");
    const bool draws = wave < NDRAW;                       // this wavefront draws; its tree groups are taken by wavefront wave + 16 - NDRAW
    const int mirror = wave - (16 - NDRAW);                // >= 0: this wavefront also sums the groups of wavefront `mirror`
    double part_s[K], part_q[K], part2_s[K], part2_q[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
        part_s[s] = (!SPREAD && mode == 0 && !draws) ? partial[(uint64_t)(s * 2 + 0) * HML_REDUCE_GROUPS + tid] : 0.0;
        part_q[s] = (!SPREAD && mode == 0 && !draws) ? partial[(uint64_t)(s * 2 + 1) * HML_REDUCE_GROUPS + tid] : 0.0;
        part2_s[s] = (!SPREAD && mode == 0 && mirror >= 0) ? partial[(uint64_t)(s * 2 + 0) * HML_REDUCE_GROUPS + (mirror * 64 + lane)] : 0.0;
        part2_q[s] = (!SPREAD && mode == 0 && mirror >= 0) ? partial[(uint64_t)(s * 2 + 1) * HML_REDUCE_GROUPS + (mirror * 64 + lane)] : 0.0;
    }

    // s_occ / s_trans (LDS) must be visible to the drawing lanes; the group partials requested above may keep travelling:
    // a barrier that waits for LDS traffic only (__syncthreads would also wait for every outstanding global load)
    if (SPREAD) __syncthreads();   // (this workgroup's first-level sums are on their way before its ticket is taken)
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid == 0) mdl->dbg_t[1] = wall_clock64();
    if constexpr (SPREAD) {
        // the ticket, by the last wavefront (it draws nothing) while the others draw
        static_assert(NDRAW <= 8, "wavefronts 8 (occupancies) and 15 (ticket) draw nothing");
        if (wave == 15) {
            uint32_t last = 0u;
            if (lane == 0) {
                __threadfence();   // this workgroup's sums before its ticket
                const uint32_t t = atomicAdd(&mdl->params_ticket, 1u);
                last = ((t + 1u) % (uint32_t)nleaf == 0u) ? 1u : 0u;
                s_last = last;
            }
            last = (uint32_t)__builtin_amdgcn_readfirstlane((int)last);
            if (last != 0u) {   // wave-uniform: fetch the first-level sums of all sixteen
                __threadfence();
                if (lane < 2 * K) {
                    unsigned long long raw[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) raw[i] = __hip_atomic_load(reinterpret_cast<unsigned long long*>(level1) + i * 2 * K + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int i = 0; i < 16; ++i) wp[i][lane >> 1][lane & 1] = hml_u2d(raw[i]);
                }
            }
        }
    }

    // ---- the draws that need the COUNTS only, ahead of the sums: the gamma rejection loop depends on its shape parameter
    // alone (hml_gamma_core_f32), the normal variate on nothing; scale and location follow behind the tree.  Same
    // operations on the same Philox sub-streams as drawing everything afterwards.
    unsigned long long cnt = 0ull;       // number of terms of parameter k (theta lanes)
    float th_alpha = hyp0, th_gcore = 0.0f, th_z = 0.0f;
    if (wave == 0 && lane < P) {
        const int k = lane;
        // the positions of every (state, dimension) mapped to parameter k - the occupancy of state k when D = 1
        if (mode == 0) {
            if (D == 1) cnt = s_occ[k];
            else for (int st = 0; st < K; ++st) for (int d = 0; d < D; ++d) if (mdl->map[st][d] == k) cnt += s_occ[st];
        }
        if (cnt > 0ull) th_alpha = (float)((double)hyp0 + (double)cnt / 2.0);   // Conjugate.hpp:141-146: alpha + N / 2
        hml_dev_src src;
        src.s = hml_stream_open(key, HML_KIND_THETA, epoch, (uint32_t)k);
        th_gcore = hml_gamma_core_f32<hml_devmath>(src, th_alpha);
        hml_normal_f32<hml_devmath> nd;
        th_z = nd.draw_std(src);
        if (tid == 0) mdl->dbg_t[7] = wall_clock64();
    }
    if (mode != 2) {
        if (wave == 1 && lane < K) {
            const int k = lane;
            const float al = hyp0 + (float)s_occ[k];
            hml_dev_src src;
            src.s = hml_stream_open(key, HML_KIND_PI, epoch, (uint32_t)k);
            praw[k] = hml_gamma_f32<hml_devmath>(src, al, 1.0f);
        }
        if (tid >= 128 && tid < 128 + K * K) {
            const int e = tid - 128;
            const float al = hyp0 + (float)s_trans[e];
            hml_dev_src src;
            src.s = hml_stream_open(key, HML_KIND_TRANS, epoch, (uint32_t)e);
            graw[e] = hml_gamma_f32<hml_devmath>(src, al, 1.0f);
        }
    }
    if (tid == 128) mdl->dbg_t[4] = wall_clock64();

    if constexpr (SPREAD) {
        __syncthreads();   // the ticket's outcome and, in the workgroup that goes on, the first-level sums
        if (s_last == 0u) return;   // workgroup-uniform; nothing outside the workgroup was written so far
        if (tid >= 128 && tid < 128 + K * K) {
            const int e = tid - 128;
#pragma unroll
            for (int sp = 0; sp < HML_CNT_SPLIT; ++sp) mdl->trans[sp][e] = 0ull;
            mdl->last_trans[e] = acc_total;
        }
        if (tid >= 512 && tid < 512 + K) {
            const int k = tid - 512;
#pragma unroll
            for (int sp = 0; sp < HML_CNT_SPLIT; ++sp) mdl->occ[sp][k] = 0ull;
            mdl->last_occ[k] = acc_total;
        }
    }
    if (mode == 0) {
        // finish the fixed tree over the 1024 group partials: pairwise inside each wavefront ...
        if (!SPREAD && !draws) {   // wave-uniform
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const double a = hml_wave_tree_f64(part_s[s]), d = hml_wave_tree_f64(part_q[s]);
                if (lane == 0) { wp[wave][s][0] = a; wp[wave][s][1] = d; }
            }
        }
        if (!SPREAD && mirror >= 0) {
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const double a = hml_wave_tree_f64(part2_s[s]), d = hml_wave_tree_f64(part2_q[s]);
                if (lane == 0) { wp[mirror][s][0] = a; wp[mirror][s][1] = d; }
            }
        }
        if (tid == 1023) mdl->dbg_t[8] = wall_clock64();
        __syncthreads();
        // ... then a pairwise tree over the 16 wavefront sums
        if (tid < 2 * K) {
            const int s = tid >> 1, c = tid & 1;
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = wp[i][s][c];
#pragma unroll
            for (int st = 1; st < 16; st <<= 1)
#pragma unroll
                for (int i = 0; i < 16; i += 2 * st) v[i] = v[i] + v[i + st];
            fin[s][c] = (float)v[0];
        }
        __syncthreads();
        if (tid == 0) mdl->dbg_t[2] = wall_clock64();
    }

    if (wave == 0 && lane < P) {
        const int k = lane;
        float alpha = hyp0, beta = hyp1, mu0 = hyp2, nu = hyp3;
        if (mode == 0) { mdl->last_sum[k] = fin[k][0]; mdl->last_sumsq[k] = fin[k][1]; }
        if (cnt > 0ull) {
            // Conjugate<NormalInverseGammaParam>::addObservation (Conjugate.hpp:121-168)
            const float sum = fin[k][0], sumSq = fin[k][1];
            if (sumSq < 0.0f) hml_raise(mdl, HML_DEVERR_NEG_SUMSQ, sumSq);
            const double N = (double)cnt;
            const float xbar = (float)((double)sum / N);
            float ssN = (float)((double)(sum * sum) / N);
            if (ssN > sumSq) ssN = sumSq;
            const float na = th_alpha;   // (float)((double)alpha + N / 2.0), computed ahead of the sums
            const float dxm = (xbar - mu0) * (xbar - mu0);
            const float nb = (float)((double)beta +
                                     (((double)sumSq + (N * (double)nu / (N + (double)nu)) * (double)dxm) - (double)ssN) / 2.0);
            const float nm = (float)((double)(nu * mu0 + sum) / ((double)nu + N));
            const float nn = (float)((double)nu + N);
            if (na <= 0.0f) hml_raise(mdl, HML_DEVERR_NIG_ALPHA, na);
            if (nb <= 0.0f) hml_raise(mdl, HML_DEVERR_NIG_BETA, nb);
            if (nn <= 0.0f) hml_raise(mdl, HML_DEVERR_NIG_NU, nn);
            if (!hml_isfinite(nm)) hml_raise(mdl, HML_DEVERR_NIG_MU0, nm);
            alpha = na; beta = nb; mu0 = nm; nu = nn;
        }
        (void)alpha;
        // Distribution<NormalInverseGamma>::resample (Distribution.hpp:77-87): gamma(alpha, 1 / beta), then the mean
        const float g = th_gcore * (float)(1.0 / (double)beta);
        const float v = (float)(1.0 / (double)g);
        const float m = th_z * HML_SQRTF(v / nu) + mu0;
        if (!hml_isfinite(m)) hml_raise(mdl, HML_DEVERR_MEAN_NOT_FINITE, m);
        if (!hml_isfinite(v)) hml_raise(mdl, HML_DEVERR_VAR_NOT_FINITE, v);
        else if (v <= 0.0f) hml_raise(mdl, HML_DEVERR_VAR_NOT_POSITIVE, v);
        const float sd = HML_SQRTF(v);
        mdl->mu[k] = m; mdl->var[k] = v; mdl->sd[k] = sd;
        mdl->rvar2[k] = 1.0 / (2.0 * (double)v);
        if (mode == 0) {   // hml_derive's logNormalizer, from the registers instead of a round trip through memory
            const float ln = hml_logf(sd) + m * m / (2 * v);
            mdl->logN[k] = ln;
            s_logN[k] = ln;
            s_var[k] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) mdl->nig_post[k][i] = mdl->nig_prior[i];
        if (tid == 0) mdl->dbg_t[3] = wall_clock64();
    }
    __syncthreads();
    if (tid == 0) mdl->dbg_t[5] = wall_clock64();
    if (mode != 2) {
        // dirichlet_sample's normalisation (Distribution.hpp:116-139): float sum in index order
        if (tid < K) {
            float sum = 0.0f;
#pragma unroll
            for (int d = 0; d < K; ++d) sum += graw[tid * K + d];
#pragma unroll
            for (int d = 0; d < K; ++d) mdl->A[tid * K + d] = graw[tid * K + d] / sum;
            if (mode == 0) {
                mdl->logA[tid] = hml_logf(graw[tid * K + tid] / sum);   // hml_derive's logA
                float r = 0.0f;                                          // ... and the state's logNormalizer
                for (int d = 0; d < D; ++d) r += s_logN[mdl->map[tid][d]];
                mdl->logNs[tid] = r;
            }
        }
        if (mode == 0 && tid == 65) {
            // hml_derive's threshold
            float mv = HML_INF_F;
#pragma unroll
            for (int k = 0; k < P; ++k) { const float v = s_var[k]; mv = (v < mv) ? v : mv; }   // std::min(result, var)
            const float l = hml_logf((float)mdl->T);
            const float arg = 2 * l * mv;
            const float t = HML_SQRTF(arg);
            mdl->thr_theta = t;
            if (mdl->dynamic) mdl->thr = t;
        }
        if (tid == 64) {
            float sum = 0.0f;
#pragma unroll
            for (int d = 0; d < K; ++d) sum += praw[d];
#pragma unroll
            for (int d = 0; d < K; ++d) mdl->pi[d] = praw[d] / sum;
        }
        // reset Dirichlet posteriors and the sweep's accumulators
        if (tid >= 128 && tid < 128 + K * K) {
            const int e = tid - 128;
            mdl->dirA[e] = (e / K == e % K) ? mdl->a_diag : mdl->a_off;
        }
        if (tid >= 512 && tid < 512 + K) {
            const int k = tid - 512;
            mdl->dirPi[k] = mdl->pi_alpha;
        }
    }
    if (mode != 0) {
        __threadfence_block();
        __syncthreads();
        hml_derive<K>(mdl, tid);
    }
    if (tid == 1023) {
        // adapt the forward warm-up (speed only - the rows are bit-exact for every W): double it when the repair step
        // had real work - its serial pass ran, or it recomputed more than a handful of chunks - and shrink it slowly
        // after fwd_quiet_need (16) sweeps without a single refit (shrinking while a few chunks still fail was measured: the failure
        // count has a cliff, sweeps with 10^5 refits follow)
        if (mode == 0) {
            uint32_t W = mdl->fwd_W;
            const unsigned long long refits = mdl->forward_refits - mdl->fwd_refits_seen;
            const unsigned long long serial = mdl->forward_serial - mdl->fwd_serial_seen;
            // a handful: none while a sweep has fewer than 2^22 blocks (a repair is then a visible share of the sweep),
            // one chunk in some thousands beyond
            const unsigned long long handful = (mdl->B >> 22) ? (((unsigned long long)(mdl->B >> 16) > 16ull) ? (unsigned long long)(mdl->B >> 16) : 16ull) : 0ull;
            if (mdl->tre_fused && (mdl->B >> 22)) {
                // fused trellis path: a stale chunk costs one wavefront ~0.1 ms, in parallel with all the others, while
                // every block pays for the warm-up - so the warm-up follows the refit count down to a few per
                // ten thousand chunks instead of insisting on none: +8 above B / 2^17 refits (or a sequential finish),
                // -8 below B / 2^20
                const unsigned long long hi = (unsigned long long)(mdl->B >> mdl->tre_hi_shift) + 16ull, lo = (unsigned long long)(mdl->B >> mdl->tre_lo_shift) + 2ull;
                // The number of stale chunks is a cliff in W (a factor 10-100 per 8 rows below some length that depends on the
                // parameters), so the band between `lo` and `hi` may hold no W at all: the rule would then step down into the
                // cliff and back up every other sweep (K = 6 on 5e7 blocks: 61 000 refits per sweep on average, 1.0 ms of a
                // 2.6 ms sweep).  The length that last let the refits explode is therefore remembered as a floor - W stays
                // one step above it - and forgotten one step every 256 sweeps (the parameters move).
                if (serial != 0ull || refits > hi) {
                    W = (W + 8u < (uint32_t)HML_TRE_HALO_MAX) ? W + 8u : (uint32_t)HML_TRE_HALO_MAX;
                    mdl->tre_W_floor = W;
                    mdl->tre_floor_age = 0u;
                } else {
                    if (++mdl->tre_floor_age >= 256u) { mdl->tre_floor_age = 0u; mdl->tre_W_floor = (mdl->tre_W_floor > 8u) ? mdl->tre_W_floor - 8u : 0u; }
                    const uint32_t lowest = (mdl->tre_W_floor > 8u) ? mdl->tre_W_floor : 8u;
                    if (refits < lo && W >= lowest + 8u) W -= 8u;
                }
                if (W > (uint32_t)HML_TRE_HALO_MAX) W = (uint32_t)HML_TRE_HALO_MAX;
            } else
            {
            // (end of round 5) A warm-up that FAILED on a settled chain is remembered for 128 sweeps and the walk down stops one step (8
            // rows) above it: a sweep with stale chunks costs a repair by ONE workgroup - with 8 states on config 3's trace (3 10^5
            // blocks) the warm-up went 48 -> 36 -> 24 -> some two hundred stale chunks -> 48 every 33 sweeps, and the repairs were 137 us
            // of the 169 us average sweep.  (wl_W_need / wl_need_age: the fields of hml_k_wide_lanes.h's rule; a context has one path.)
            if (mdl->wl_W_need != 0u && ++mdl->wl_need_age > 128u) mdl->wl_W_need = 0u;
            if (serial != 0ull || refits > handful) {
                // (a failure of a few chunks is cheap and a short warm-up worth more: the headline's chain walks down to 12 rows and repairs 1-5
                // chunks every twenty-odd sweeps - held at 20-24 rows for 512 sweeps it lost 3 %; remembered from 16 stale chunks on, or a serial
                // pass, and for 128 sweeps; not for chains batched by hml_iterate_many (MANY), whose launches are bound by throughput: eight
                // chains lost 6 % with it)
                if (mdl->sweeps >= (unsigned long long)mdl->fwd_burnin_sweeps && (serial != 0ull || refits >= 16ull) && !MANY) { mdl->wl_W_need = W; mdl->wl_need_age = 0u; }
                W = (2u * W < 1024u) ? 2u * W : 1024u; mdl->fwd_quiet = 0u;
            } else if (refits == 0ull) {
                uint32_t floorW = (mdl->sweeps < (unsigned long long)mdl->fwd_burnin_sweeps) ? mdl->fwd_W_burnin : mdl->fwd_W0;
                if (mdl->wl_W_need != 0u) { const uint32_t keep = ((mdl->wl_W_need + 8u) & ~7u) < 1024u ? ((mdl->wl_W_need + 8u) & ~7u) : 1024u; floorW = keep > floorW ? keep : floorW; }
                if (++mdl->fwd_quiet >= mdl->fwd_quiet_need) {
                    const uint32_t w2 = W - W / 4u;
                    const uint32_t lower = (w2 > floorW) ? ((w2 & ~7u) > floorW ? (w2 & ~7u) : floorW) : floorW;
                    W = lower < W ? lower : W;   // (the floor may lie above the warm-up: the walk down never raises it)
                    mdl->fwd_quiet = 0u;
                }
            } else mdl->fwd_quiet = 0u;
            }
            mdl->fwd_W = W;
            mdl->fwd_refits_seen = mdl->forward_refits;
            mdl->fwd_serial_seen = mdl->forward_serial;
            mdl->fwd_serial_ran = 0u;
        }
        mdl->dbg_t[6] = wall_clock64();
        mdl->fwd_mismatch = 0u;
        mdl->fwd_mismatch2 = 0u;
        mdl->epoch = epoch + 1ull;
        if (mode == 0) { mdl->sweeps += 1ull; mdl->block_updates += (unsigned long long)mdl->B; }
    }
}
// the kernel: hml_b_params over one chain (hml_k_many.h runs it over several chains in one launch)
template <int K>
HML_KERNEL __launch_bounds__(1024) void hml_k_params(hml_model* __restrict__ mdl, const double* __restrict__ partial,
                                                     int mode) {
    hml_b_params<K, false>(mdl, partial, mode, 0, 1);
}
// ... with the tree's first level spread over HML_PARAMS_TREE_WGS workgroups (mode 0)
#define HML_PARAMS_TREE_WGS 16
template <int K>
HML_KERNEL __launch_bounds__(1024) void hml_k_params_spread(hml_model* __restrict__ mdl, double* __restrict__ partial) {
    hml_b_params<K, true>(mdl, partial, 0, (int)blockIdx.x, (int)gridDim.x);
}


#endif
