// Backward categorical sampling, mixture sampling, and the count pass.
#ifndef HML_K_BACKWARD_H
#define HML_K_BACKWARD_H

#include <type_traits>

#include "hml_dist.h"
#include "hml_k_forward.h"
#include "hml_philox.h"
#include "hml_state.h"

#define HML_MAP_IDENTITY 0xfedcba9876543210ull

// (f o g)(x) = f(g(x)) on maps [K]->[K] packed 4 bits per entry
template <int K>
__device__ __forceinline__ unsigned long long hml_map_compose(unsigned long long f, unsigned long long g) {
    unsigned long long r = 0ull;   // entries x >= K are never read
#pragma unroll
    for (int x = 0; x < K; ++x) {
        const unsigned y = (unsigned)(g >> (4 * x)) & 15u;
        const unsigned long long z = (f >> (4 * y)) & 15ull;
        r |= z << (4 * x);
    }
    return r;
}

__device__ __forceinline__ unsigned long long hml_shfl_down_u64(unsigned long long v, int d) {
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    const unsigned lo2 = __shfl_down(lo, d), hi2 = __shfl_down(hi, d);
    return ((unsigned long long)hi2 << 32) | lo2;
}

// std::discrete_distribution draw (hml_dist.h) over K register-resident weights: the literal form - p_i = w_i / sum in
// double, running sum cp_i, first i with !(cp_i < u), cp_{K-1} = 1 - with its K double divisions
template <int K>
__device__ __forceinline__ int hml_categorical_k_literal(const float (&w)[K], double u) {
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) sum += (double)w[i];
    double cp = 0.0;
    int res = K - 1;
    bool done = false;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        cp += (double)w[i] / sum;
        const double c = (i == K - 1) ? 1.0 : cp;
        if (!done && !(c < u)) { res = i; done = true; }
    }
    return res;
}

// The same draw without the divisions: cp_i >= u is decided by s_i >= u * sum, where s_i are the running double
// sums of the weights (sum = s_{K-1} is the very double the literal form divides by).  With R_i the real quotient
// (w_0 + .. + w_i) / sum, the literal cp_i lies within (i+1) 2^-53 R_i of R_i (one rounding per quotient, one per
// addition; all terms are non-negative) and (s_i - u sum) / sum within (i+1) 2^-53 max(R_i, u) of R_i - u.  So
// whenever |s_i - u sum| > 2^-44 sum for every i both forms make the same comparison as the real numbers
// (K <= 16: 34 * 2^-53 < 2^-47); otherwise - about K * 2^-43 of the draws, and every row whose sum is not a
// positive finite number (all-zero rows, negative or NaN weights) - `unsure` is set and the caller lets the literal
// form decide.
template <int K>
__device__ __forceinline__ int hml_categorical_k_fast(const float (&w)[K], double u, bool& unsure_out) {
    double s[K];
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) { acc += (double)w[i]; s[i] = acc; }
    const double t = u * acc;
    const double margin = acc * 5.684341886080802e-14;   // 2^-44
    bool unsure = !(acc > 0.0) || !(acc < 1.7976931348623157e308);
    int res = K - 1;
    bool done = false;
#pragma unroll
    for (int i = 0; i < K - 1; ++i) {
        const double d = s[i] - t;
        unsure = unsure || !(__builtin_fabs(d) > margin);
        if (!done && d >= 0.0) { res = i; done = true; }
    }
    unsure_out = unsure_out || unsure;
    return res;
}

// The float screen in front of both: the same comparisons with float running sums and a float copy of u.  The float
// sums carry a relative error of at most (K-1) 2^-24, the product fl(u) * sum one of (K+1) 2^-24 and the difference one
// more rounding, so for K <= 16 the computed s_i - u sum lies within 2^-18.9 sum of the real one: whenever it is
// farther than 2^-17 sum from zero for every i the real comparison - and with it the literal form, which agrees with
// the real numbers outside 2^-44 sum - comes out the same way.  About 3 in 10^5 draws (measured on the dense regime of
// config 3; and sums that are tiny, not finite or not positive) go on to the double forms; the rest never touch a double.
template <int K>
__device__ __forceinline__ int hml_categorical_k_screen(const float (&w)[K], float uf, bool& unsure_out) {
    static_assert(K <= 16, "error bound of the float screen");
    float s[K];
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < K; ++i) { acc += w[i]; s[i] = acc; }
    const float t = uf * acc;
    const float margin = acc * 7.62939453125e-06f;   // 2^-17
    bool unsure = !(acc > 7.888609052210118e-31f) || !(acc < 3.4028234663852886e38f);   // 2^-100 < sum < inf
    int res = K - 1;
    bool done = false;
#pragma unroll
    for (int i = 0; i < K - 1; ++i) {
        const float d = s[i] - t;
        unsure = unsure || !(__builtin_fabsf(d) > margin);
        if (!done && d >= 0.0f) { res = i; done = true; }
    }
    unsure_out = unsure;
    return res;
}

template <int K>
__device__ __forceinline__ int hml_categorical_k(const float (&w)[K], double u) {
    bool unsure = false;
    int res = hml_categorical_k_screen<K>(w, (float)u, unsure);
    if (unsure) {
        // (the empty statement keeps the compiler from evaluating the double forms ahead of the branch for every draw)
        asm volatile("" ::: "memory");
        unsure = false;
        res = hml_categorical_k_fast<K>(w, u, unsure);
        if (unsure) res = hml_categorical_k_literal<K>(w, u);
    }
    return res;
}

// ------------------------------------------------------------------------------------------
// K7a backward_maps - backward sampling (reference src/StateSequence/ForwardBackward.hpp:133-162,
// Trellis::sample src/Trellis.hpp:61-66):  q_B ~ Cat(r_B),  q_t ~ Cat(r_t[i] * A(i, q_{t+1})).
// Row t's uniform u_t is addressed by the row alone (hml_cat_uniform, hml_dist.h), so for every possible
// successor state x the draw cand_t(x) is known in advance: row t is a MAP [K]->[K].  The state
// sequence is the composition of those maps applied from the last row down - associative, hence a
// scan.  One wavefront owns 64 consecutive rows: lane l evaluates row t = 64c + l + 1, then a
// suffix scan over the wavefront yields S_t = cand_t o ... o cand_{64c+64} for each row and the
// chunk's map S_{64c+1}.
// ------------------------------------------------------------------------------------------
// row t = 64c + lane + 1 of backward chunk c (rows are stored by block b = t-1).  `starts` != nullptr: the forward pass
// stored the rows unscaled (it was given no plane of rescale factors) and the factor of row t < B - expf((N - 1) logA_s)
// with N the size of block t - 1, the very expression of hml_emit_compute - is applied here (ForwardBackward.hpp:115-119).
// The model's read-only values the backward maps need, fetched once per wavefront through a pointer the compiler may
// treat as constant (scalar loads, one wait): read through the kernels' writable hml_model* every element of A was a
// separate vector load with its own wait - 25 cache round trips in a row per chunk at K = 5.
template <int K>
struct hml_bwd_ctx {
    hml_amat<K> A;   // registers up to 7 states, the workgroup's LDS copy beyond (hml_k_forward.h)
    float logA[K];
    bool self;
};
template <int K>
__device__ __forceinline__ void hml_bwd_ctx_load(hml_bwd_ctx<K>& bx, const hml_model* __restrict__ mdl_ro, const float* lds_A) {
    bx.A.attach(mdl_ro, lds_A);
#pragma unroll
    for (int i = 0; i < K; ++i) bx.logA[i] = mdl_ro->logA[i];
    bx.self = mdl_ro->self_trans != 0;
}

template <int K>
__device__ __forceinline__ void hml_bwd_row_load(const float* __restrict__ rows, const hml_layout lay, uint32_t c, int lane,
                                                 uint32_t B, float (&r)[K], const uint32_t* __restrict__ starts,
                                                 const hml_bwd_ctx<K>& bx) {
    const uint32_t t = c * HML_BWD_CHUNK + (uint32_t)lane + 1u;
    uint32_t st = 0u, en = 1u;
    if (starts && t < B) { st = starts[t - 1u]; en = starts[t]; }
#pragma unroll
    for (int i = 0; i < K; ++i) r[i] = (t <= B) ? rows[hml_bk(lay, t - 1u, K, i)] : 0.0f;
    if (starts && t < B && bx.self) {
        const float N = (float)(en - st);
#pragma unroll
        for (int i = 0; i < K; ++i) r[i] = r[i] * hml_expf((N - 1.0f) * bx.logA[i]);
    }
}

// the maps of backward chunk c (rows 64c+1 .. 64c+64), by one wavefront
template <int K>
__device__ __forceinline__ void hml_bwd_chunk_maps(const float (&r)[K], hml_model* __restrict__ mdl, const hml_bwd_ctx<K>& bx,
                                                   unsigned long long* __restrict__ smap, unsigned long long* __restrict__ cmap,
                                                   uint32_t c, int lane, uint32_t B, unsigned long long epoch, const hml_key key) {
    const uint32_t t = c * HML_BWD_CHUNK + (uint32_t)lane + 1u;
    unsigned long long map = HML_MAP_IDENTITY;
    if (t <= B) {
        const double u = hml_cat_uniform(key, epoch, t);
        map = 0ull;
        // "Negative backward variable!" (ForwardBackward.hpp:147-149): the products r_i A(i, x) below are negative exactly
        // when r_i is (A holds probabilities), so the row is checked once instead of K * K times with a branch each
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (r[i] < 0.0f) hml_raise(mdl, HML_DEVERR_NEG_BACKWARD, r[i]);
        if (t == B) {
            const unsigned long long st = (unsigned long long)hml_categorical_k<K>(r, u);
#pragma unroll
            for (int x = 0; x < K; ++x) map |= st << (4 * x);
        } else {
#pragma unroll
            for (int x = 0; x < K; ++x) {
                float w[K];
#pragma unroll
                for (int i = 0; i < K; ++i) w[i] = r[i] * bx.A[i * K + x];
                map |= (unsigned long long)hml_categorical_k<K>(w, u) << (4 * x);
            }
        }
    }
    // suffix scan of map composition across the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long o = hml_shfl_down_u64(map, d);
        if (lane + d >= 64) o = HML_MAP_IDENTITY;
        map = hml_map_compose<K>(map, o);
    }
    if (t <= B) smap[t] = map;
    if (lane == 0) cmap[c] = map;
}

// One wavefront per backward chunk.  It first verifies the forward chunks whose rows it is about to read
// (start vector == predecessor's end vector, bit for bit; see hml_k_forward): after a failed check the chunk
// goes on the list of the repair step (fail_list, counted by mdl->fwd_mismatch), which also computes its maps.
template <int K>
__device__ __forceinline__ void hml_b_backward_maps(const float* __restrict__ rows, hml_model* __restrict__ mdl,
                                                           unsigned long long* __restrict__ smap,
                                                           unsigned long long* __restrict__ cmap, const hml_layout lay,
                                                           const float* __restrict__ entry, const float* __restrict__ exitv,
                                                           uint32_t* __restrict__ fail_list, int L,
                                                           const uint32_t* __restrict__ starts,
                                                           const hml_model* __restrict__ mdl_ro) {
    const uint32_t B = mdl_ro->B;
    const uint32_t nchunks = (B + HML_BWD_CHUNK - 1u) / HML_BWD_CHUNK;
    const int lane = threadIdx.x & 63;
    const uint32_t wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const unsigned long long epoch = mdl_ro->epoch;
    const hml_key key = mdl_ro->key;
    const int W = (int)mdl_ro->fwd_W;
    __shared__ float sm_A[hml_amat<K>::LDS_FLOATS];
    hml_amat_fill<K>(sm_A, mdl_ro, (int)threadIdx.x, (int)blockDim.x);
    hml_bwd_ctx<K> bx;
    hml_bwd_ctx_load<K>(bx, mdl_ro, sm_A);
    const uint32_t C = (B + (uint32_t)L - 1u) / (uint32_t)L;
    for (uint32_t c = wave_global; c < nchunks; c += nwaves) {
        float r[K];
        hml_bwd_row_load<K>(rows, lay, c, lane, B, r, starts, bx);   // in flight together with the verification's loads
        // forward chunks that overlap blocks [64c, 64c+64): at most 64 of them, one per lane
        bool ok = true;
        {
            const uint32_t f0 = (c * HML_BWD_CHUNK) / (uint32_t)L;
            const uint32_t f1 = (c * HML_BWD_CHUNK + HML_BWD_CHUNK - 1u) / (uint32_t)L;
            const uint32_t f = f0 + (uint32_t)lane;
            if (f <= f1 && f < C && !hml_fwd_chunk_exact(f, L, W)) {
                uint32_t differ = 0u;   // (no short circuit: the 2 K loads travel together)
#pragma unroll
                for (int s = 0; s < K; ++s) differ |= hml_f2u(entry[(uint64_t)f * K + s]) ^ hml_f2u(exitv[(uint64_t)(f - 1) * K + s]);
                ok = differ == 0u;
            }
        }
        if (__ballot(!ok) != 0ull) {   // wave-uniform
            if (lane == 0) fail_list[atomicAdd(&mdl->fwd_mismatch, 1u)] = c;   // at most one entry per backward chunk
            continue;
        }
        hml_bwd_chunk_maps<K>(r, mdl, bx, smap, cmap, c, lane, B, epoch, key);
    }
}
// the kernel: hml_b_backward_maps over one chain (hml_k_many.h runs it over several chains in one launch)
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_k_backward_maps(const float* __restrict__ rows, hml_model* __restrict__ mdl,
                                                           unsigned long long* __restrict__ smap,
                                                           unsigned long long* __restrict__ cmap, const hml_layout lay,
                                                           const float* __restrict__ entry, const float* __restrict__ exitv,
                                                           uint32_t* __restrict__ fail_list, int L,
                                                           const uint32_t* __restrict__ starts,
                                                           const hml_model* __restrict__ mdl_ro) {
    hml_b_backward_maps<K>(rows, mdl, smap, cmap, lay, entry, exitv, fail_list, L, starts, mdl_ro);
}


// The same maps with TWO rows per lane: a wavefront owns two consecutive backward chunks (128 rows), lane l the rows
// 2 l' + 1 and 2 l' + 2 (l' = l mod 32) of chunk 2 w + l / 32.  Blocks 2 m and 2 m + 1 share Philox block m of the sweep
// (hml_cat_uniform_pair), so the lane pays for ten rounds ONCE for its two rows, and the suffix scan runs over 32 pairs per
// chunk (five steps for two rows instead of six for one).  Same smap / cmap / fail list as hml_b_backward_maps - the
// chunks stay 64 rows, every consumer is unchanged.  For chains batched by hml_iterate_many, where the kernel is bound by
// vector issue (eight chains: 35.6 us with one row per lane).  Reference: ForwardBackward.hpp:133-162, Trellis.hpp:61-66.
template <int K>
__device__ __forceinline__ unsigned long long hml_bwd_row_map(const float (&r)[K], hml_model* __restrict__ mdl, const hml_bwd_ctx<K>& bx,
                                                              uint32_t t, uint32_t B, double u) {
    if (t > B) return HML_MAP_IDENTITY;
    unsigned long long map = 0ull;
    // "Negative backward variable!" (ForwardBackward.hpp:147-149), checked once per row (see hml_bwd_chunk_maps)
#pragma unroll
    for (int i = 0; i < K; ++i)
        if (r[i] < 0.0f) hml_raise(mdl, HML_DEVERR_NEG_BACKWARD, r[i]);
    if (t == B) {
        const unsigned long long st = (unsigned long long)hml_categorical_k<K>(r, u);
#pragma unroll
        for (int x = 0; x < K; ++x) map |= st << (4 * x);
    } else {
#pragma unroll
        for (int x = 0; x < K; ++x) {
            float w[K];
#pragma unroll
            for (int i = 0; i < K; ++i) w[i] = r[i] * bx.A[i * K + x];
            map |= (unsigned long long)hml_categorical_k<K>(w, u) << (4 * x);
        }
    }
    return map;
}
// row t (1-based) of the trellis as the backward pass reads it (hml_bwd_row_load for one row)
template <int K>
__device__ __forceinline__ void hml_bwd_row_load_t(const float* __restrict__ rows, const hml_layout lay, uint32_t t, uint32_t B, float (&r)[K],
                                                   const uint32_t* __restrict__ starts, const hml_bwd_ctx<K>& bx) {
    uint32_t st = 0u, en = 1u;
    if (starts && t < B) { st = starts[t - 1u]; en = starts[t]; }
#pragma unroll
    for (int i = 0; i < K; ++i) r[i] = (t <= B) ? rows[hml_bk(lay, t - 1u, K, i)] : 0.0f;
    if (starts && t < B && bx.self) {
        const float N = (float)(en - st);
#pragma unroll
        for (int i = 0; i < K; ++i) r[i] = r[i] * hml_expf((N - 1.0f) * bx.logA[i]);
    }
}
template <int K>
__device__ __forceinline__ void hml_b_backward_maps2(const float* __restrict__ rows, hml_model* __restrict__ mdl,
                                                            unsigned long long* __restrict__ smap,
                                                            unsigned long long* __restrict__ cmap, const hml_layout lay,
                                                            const float* __restrict__ entry, const float* __restrict__ exitv,
                                                            uint32_t* __restrict__ fail_list, int L,
                                                            const uint32_t* __restrict__ starts,
                                                            const hml_model* __restrict__ mdl_ro) {
    const uint32_t B = mdl_ro->B;
    const uint32_t nchunks = (B + HML_BWD_CHUNK - 1u) / HML_BWD_CHUNK;
    const uint32_t npairs = (nchunks + 1u) / 2u;
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
    const uint32_t wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const unsigned long long epoch = mdl_ro->epoch;
    const hml_key key = mdl_ro->key;
    const int W = (int)mdl_ro->fwd_W;
    __shared__ float sm_A[hml_amat<K>::LDS_FLOATS];
    hml_amat_fill<K>(sm_A, mdl_ro, (int)threadIdx.x, (int)blockDim.x);
    hml_bwd_ctx<K> bx;
    hml_bwd_ctx_load<K>(bx, mdl_ro, sm_A);
    const uint32_t C = (B + (uint32_t)L - 1u) / (uint32_t)L;
    for (uint32_t pw = wave_global; pw < npairs; pw += nwaves) {
        const uint32_t c = 2u * pw + (uint32_t)half;                          // this lane's backward chunk
        const uint32_t t0 = c * HML_BWD_CHUNK + 2u * (uint32_t)li + 1u;      // its rows: t0, t0 + 1 (blocks 2 m, 2 m + 1 of pair m)
        float r0[K], r1[K];
        hml_bwd_row_load_t<K>(rows, lay, t0, B, r0, starts, bx);              // in flight together with the verification's loads
        hml_bwd_row_load_t<K>(rows, lay, t0 + 1u, B, r1, starts, bx);
        // forward chunks that overlap blocks [64 c, 64 c + 64): up to 64 of them, 32 lanes
        bool ok = true;
        if (c < nchunks) {
            const uint32_t f0 = (c * HML_BWD_CHUNK) / (uint32_t)L;
            const uint32_t f1 = (c * HML_BWD_CHUNK + HML_BWD_CHUNK - 1u) / (uint32_t)L;
            for (uint32_t f = f0 + (uint32_t)li; f <= f1 && f < C; f += 32u) {
                if (hml_fwd_chunk_exact(f, L, W)) continue;
                uint32_t differ = 0u;   // (no short circuit: the 2 K loads travel together)
#pragma unroll
                for (int s = 0; s < K; ++s) differ |= hml_f2u(entry[(uint64_t)f * K + s]) ^ hml_f2u(exitv[(uint64_t)(f - 1) * K + s]);
                ok = ok && differ == 0u;
            }
        }
        const unsigned long long failed = __ballot(!ok);
        const bool my_fail = (half ? (failed >> 32) : (failed & 0xffffffffull)) != 0ull;
        if (my_fail && li == 0 && c < nchunks) fail_list[atomicAdd(&mdl->fwd_mismatch, 1u)] = c;   // at most one entry per backward chunk
        if ((failed & 0xffffffffull) != 0ull && (failed >> 32) != 0ull) continue;   // both chunks failed (wave-uniform): nothing to compute here
        double u0, u1;
        hml_cat_uniform_pair(key, epoch, (t0 - 1u) >> 1, u0, u1);
        const bool live = !my_fail && c < nchunks;
        const unsigned long long m0 = live ? hml_bwd_row_map<K>(r0, mdl, bx, t0, B, u0) : HML_MAP_IDENTITY;
        const unsigned long long m1 = live ? hml_bwd_row_map<K>(r1, mdl, bx, t0 + 1u, B, u1) : HML_MAP_IDENTITY;
        // suffix scan of the pairs within each half of the wavefront
        unsigned long long P = hml_map_compose<K>(m0, m1);
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            unsigned long long o = hml_shfl_down_u64(P, d);
            if (li + d >= 32) o = HML_MAP_IDENTITY;
            P = hml_map_compose<K>(P, o);
        }
        unsigned long long later = hml_shfl_down_u64(P, 1);   // the pairs behind this one
        if (li == 31) later = HML_MAP_IDENTITY;
        if (live) {
            if (t0 <= B) smap[t0] = P;
            if (t0 + 1u <= B) smap[t0 + 1u] = hml_map_compose<K>(m1, later);
            if (li == 0) cmap[c] = P;
        }
    }
}

// K7b repair + chain, one workgroup.  Normally only the chain: compose the chunk maps from the last chunk down,
// entry[c] = state of the first row of chunk c+1 (entry of the last chunk is a dummy 0: its map is constant).
// When a verification failed, the forward repair (hml_fwd_repair) and the maps of the affected chunks come first.
template <int K>
__device__ __forceinline__ void hml_b_backward_chain(unsigned long long* __restrict__ cmap, hml_model* __restrict__ mdl,
                                                             uint8_t* __restrict__ entry_state, const float* __restrict__ em,
                                                             const float* __restrict__ gsc, float* __restrict__ rows,
                                                             float* __restrict__ aprobe, float* __restrict__ entry,
                                                             float* __restrict__ exitv, uint32_t* __restrict__ fb_count,
                                                             const uint32_t* __restrict__ fail_list, uint32_t* __restrict__ touched,
                                                             unsigned long long* __restrict__ smap, int L, const hml_layout lay,
                                                             int mode, int super_level, const uint32_t* __restrict__ starts) {
    // mode: bit 0 = repair step, bit 1 = chain.  super_level: the chain runs over the maps of super-chunks (64 backward
    // chunks each, hml_k_backward_super) - `cmap` and `entry_state` are then the super-level arrays.
    __shared__ unsigned long long P[1024];
    __shared__ hml_repair_lds sh;
    const uint32_t B = mdl->B;
    const uint32_t NCb = (B + HML_BWD_CHUNK - 1u) / HML_BWD_CHUNK;
    const uint32_t NC = super_level ? (NCb + 63u) / 64u : NCb;
    const int tid = threadIdx.x;
    const uint32_t n_fail = (mode & 1) ? mdl->fwd_mismatch : 0u;   // workgroup-uniform: written by the previous launch
    if (n_fail != 0u) {
        const unsigned long long epoch = mdl->epoch;
        const uint32_t gen = (uint32_t)epoch + 1u;
        const hml_key key = mdl->key;
        __shared__ float sm_A[hml_amat<K>::LDS_FLOATS];
        hml_amat_fill<K>(sm_A, mdl, tid, 1024);
        hml_fwd_repair<K>(em, gsc, mdl, rows, aprobe, entry, exitv, fb_count, fail_list, n_fail, touched, gen, L, lay, sh, sm_A);
        // maps of the chunks that failed verification and of those whose rows were recomputed
        const int lane = tid & 63, wave = tid >> 6;
        hml_bwd_ctx<K> bx;
        hml_bwd_ctx_load<K>(bx, mdl, sm_A);
        auto redo = [&](uint32_t c) {
            float r[K];
            hml_bwd_row_load<K>(rows, lay, c, lane, B, r, starts, bx);
            hml_bwd_chunk_maps<K>(r, mdl, bx, smap, cmap, c, lane, B, epoch, key);
        };
        for (uint32_t i = (uint32_t)wave; i < n_fail; i += 16u) redo(fail_list[i]);
        const uint32_t n_touched = sh.tcount;
        if (n_touched <= (uint32_t)HML_REPAIR_TOUCHED_CAP) {
            for (uint32_t i = (uint32_t)wave; i < n_touched; i += 16u) redo(sh.tlist[i]);
        } else {
            // the list overflowed: scan the marks in memory
            for (uint32_t c0 = (uint32_t)wave * 64u; c0 < NCb; c0 += 16u * 64u) {
                const uint32_t c = c0 + (uint32_t)lane;
                unsigned long long todo = __ballot(c < NCb && hml_ld_u32_coherent(touched + c) == gen);
                while (todo) {
                    const int j = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    redo(c0 + (uint32_t)j);
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        if (tid == 0) mdl->fwd_mismatch = 0u;
    }
    if (!(mode & 2)) return;
    const uint32_t per = (NC + 1023u) / 1024u;
    const uint32_t a = (uint32_t)tid * per < NC ? (uint32_t)tid * per : NC;
    const uint32_t b = (a + per < NC) ? a + per : NC;
    const int lane = tid & 63, wave = tid >> 6;
    // product of this thread's maps: cmap[a] o cmap[a+1] o ... o cmap[b-1].  Up to four maps per thread (NC <= 4096: every
    // strongly compressed sweep) are fetched together and kept for the walk at the end - read one by one through the
    // writable pointer each was a cache round trip of its own, twice.
    constexpr int CACHED = 4;
    unsigned long long mine[CACHED];
    unsigned long long prod = HML_MAP_IDENTITY;
    if (per <= (uint32_t)CACHED) {   // workgroup-uniform
#pragma unroll
        for (int j = 0; j < CACHED; ++j) mine[j] = (a + (uint32_t)j < b) ? cmap[a + (uint32_t)j] : HML_MAP_IDENTITY;
#pragma unroll
        for (int j = CACHED - 1; j >= 0; --j) prod = hml_map_compose<K>(mine[j], prod);
    } else {
        for (uint32_t c = b; c > a; --c) prod = hml_map_compose<K>(cmap[c - 1], prod);
    }
    // suffix products over the 1024 threads, Q[tid] = P[tid] o P[tid+1] o ... o P[1023]: inside a wavefront by shuffles
    // (S), the sixteen wavefront products by the first sixteen lanes (LDS, two barriers instead of twenty)
    unsigned long long S = prod;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long o = hml_shfl_down_u64(S, d);
        if (lane + d >= 64) o = HML_MAP_IDENTITY;
        S = hml_map_compose<K>(S, o);
    }
    if (lane == 0) P[wave] = S;
    __syncthreads();
    if (tid < 64) {
        unsigned long long w = (tid < 16) ? P[tid] : HML_MAP_IDENTITY;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            unsigned long long o = hml_shfl_down_u64(w, d);
            if (tid + d >= 16) o = HML_MAP_IDENTITY;
            w = hml_map_compose<K>(w, o);
        }
        if (tid < 16) P[16 + tid] = w;   // P[16 + w] = product of the wavefronts w .. 15
    }
    __syncthreads();
    const unsigned long long after_wave = (wave < 15) ? P[16 + wave + 1] : HML_MAP_IDENTITY;
    // the composition of everything behind this thread: Q[tid + 1]
    unsigned long long next_S = hml_shfl_down_u64(S, 1);
    const unsigned long long later = (lane < 63) ? hml_map_compose<K>(next_S, after_wave) : after_wave;
    if (a >= b) return;
    // state entering this thread's last chunk = (composition of all later chunks)(dummy 0)
    unsigned x = (unsigned)(later & 15ull);
    if (per <= (uint32_t)CACHED) {
#pragma unroll
        for (int j = CACHED - 1; j >= 0; --j) {
            if (a + (uint32_t)j < b) {
                entry_state[a + (uint32_t)j] = (uint8_t)x;
                x = (unsigned)(mine[j] >> (4 * x)) & 15u;
            }
        }
    } else {
        for (uint32_t c = b; c > a; --c) {
            entry_state[c - 1] = (uint8_t)x;
            x = (unsigned)(cmap[c - 1] >> (4 * x)) & 15u;
        }
    }
}
// the kernel: hml_b_backward_chain over one chain (hml_k_many.h runs it over several chains in one launch)
template <int K>
HML_KERNEL __launch_bounds__(1024) void hml_k_backward_chain(unsigned long long* __restrict__ cmap, hml_model* __restrict__ mdl,
                                                             uint8_t* __restrict__ entry_state, const float* __restrict__ em,
                                                             const float* __restrict__ gsc, float* __restrict__ rows,
                                                             float* __restrict__ aprobe, float* __restrict__ entry,
                                                             float* __restrict__ exitv, uint32_t* __restrict__ fb_count,
                                                             const uint32_t* __restrict__ fail_list, uint32_t* __restrict__ touched,
                                                             unsigned long long* __restrict__ smap, int L, const hml_layout lay,
                                                             int mode, int super_level, const uint32_t* __restrict__ starts) {
    hml_b_backward_chain<K>(cmap, mdl, entry_state, em, gsc, rows, aprobe, entry, exitv, fb_count, fail_list, touched, smap, L, lay, mode, super_level, starts);
}


// Two-level chain for sweeps with millions of backward chunks (one workgroup walking all chunk maps would take
// milliseconds): a wavefront composes the maps of 64 consecutive chunks - scmap[c] = cmap[c] o .. o cmap[last chunk of
// its super-chunk], super[S] = scmap[64 S] -, the one-workgroup chain runs over the super maps only (super_level = 1),
// and the state entering chunk c follows from its super-chunk's: entry[c] = scmap[c + 1](entry2[S]).
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_k_backward_super(const unsigned long long* __restrict__ cmap, const hml_model* __restrict__ mdl,
                                                            unsigned long long* __restrict__ scmap, unsigned long long* __restrict__ super) {
    const uint32_t NC = (mdl->B + HML_BWD_CHUNK - 1u) / HML_BWD_CHUNK;
    const uint32_t NS = (NC + 63u) / 64u;
    const int lane = threadIdx.x & 63;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t S = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; S < NS; S += nwaves) {
        const uint32_t c = S * 64u + (uint32_t)lane;
        unsigned long long map = c < NC ? cmap[c] : HML_MAP_IDENTITY;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long o = hml_shfl_down_u64(map, d);
            if (lane + d >= 64) o = HML_MAP_IDENTITY;
            map = hml_map_compose<K>(map, o);
        }
        if (c < NC) scmap[c] = map;
        if (lane == 0) super[S] = map;
    }
}

HML_KERNEL __launch_bounds__(256) void hml_k_backward_entries(const unsigned long long* __restrict__ scmap, const uint8_t* __restrict__ entry2,
                                                              const hml_model* __restrict__ mdl, uint8_t* __restrict__ entry_state) {
    const uint32_t NC = (mdl->B + HML_BWD_CHUNK - 1u) / HML_BWD_CHUNK;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < NC; c += stride) {
        const unsigned e2 = entry2[c >> 6];
        const bool last = (c & 63u) == 63u || c + 1u == NC;
        entry_state[c] = last ? (uint8_t)e2 : (uint8_t)((scmap[c + 1u] >> (4u * e2)) & 15ull);
    }
}

// ------------------------------------------------------------------------------------------
// K6m mixture - StateSequence<Mixture>::sample (reference src/StateSequence/Mixture.hpp:90-129):
// q_b ~ Cat(expf(E_s - max E)) independently per block, uniform from sub-stream (MIX, epoch, b).
// `em` holds the weights written by the emission kernel in mixture mode.
// ------------------------------------------------------------------------------------------
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_k_mixture(const float* __restrict__ em, const hml_model* __restrict__ mdl,
                                                     int16_t* __restrict__ q, const hml_layout lay) {
    const uint32_t B = mdl->B;
    const unsigned long long epoch = mdl->epoch;
    const hml_key key = mdl->key;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += stride) {
        float w[K];
#pragma unroll
        for (int s = 0; s < K; ++s) w[s] = em[hml_bk(lay, b, K, s)];
        const hml_u32x4 o = hml_stream4(key, HML_KIND_MIX, epoch, b, 0);
        q[b] = (int16_t)hml_categorical_k<K>(w, hml_canonical_f64(o.v[0], o.v[1]));
    }
}

// ------------------------------------------------------------------------------------------
// K8 state_reduce - the count pass (reference src/StateSequence/ForwardBackward.hpp:170-200,
// Mixture.hpp:113-128): K x K transition counts ([s][s] += N-1, [prev][s] += 1, prev_0 = 0),
// occupancies, per-state (sum x, sum x^2, n).  Counts are exact integers (device atomics).  The
// floating-point sums use a FIXED tree so that the result does not depend on scheduling: block b belongs
// to chunk c = b / 256, wavefront w = (b / 64) % 4, lane l = b % 64 and group g = c % 1024; accumulator
// (g, w, l) adds its blocks' terms in increasing block order (registers, no cross-lane traffic in the
// loop), then a pairwise tree over the 64 lanes, the four wavefronts in order, and - in the parameter
// kernel - a pairwise tree over the 1024 groups.  Doubles throughout; the CPU checker mirrors the tree.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double hml_shfl_xor_f64(double v, int m) {
    const unsigned long long u = hml_d2u(v);
    const unsigned lo = __shfl_xor((unsigned)u, m), hi = __shfl_xor((unsigned)(u >> 32), m);
    return hml_u2d(((unsigned long long)hi << 32) | lo);
}

// Pairwise tree over the 64 lanes of a wavefront on DPP lane exchanges (pairs, quads, half rows, rows: the partner lane
// always holds the sibling subtree's sum, and IEEE addition is commutative, so the result is exactly the tree
// ((l0 + l1) + (l2 + l3)) + ... of the CPU checker), finished with the four row sums read into scalars.
__device__ __forceinline__ double hml_dpp_f64(double v, int ctrl_quad1, int which) {
    // lane exchange of a double through DPP: which = 0 quad_perm [1,0,3,2], 1 quad_perm [2,3,0,1], 2 row_half_mirror, 3 row_mirror
    const unsigned long long u = hml_d2u(v);
    int lo = (int)(unsigned)u, hi = (int)(unsigned)(u >> 32);
    (void)ctrl_quad1;
    switch (which) {
        case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;
        case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); break;
        case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); break;
        default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); break;
    }
    return hml_u2d(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double hml_readlane_f64(double v, int src) {
    const unsigned long long u = hml_d2u(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return hml_u2d(((unsigned long long)hi << 32) | lo);
}
// pairwise tree over the 64 lanes (the value lane 0 of hml_k_counts' xor butterfly ends with), wave-uniform result
__device__ __forceinline__ double hml_wave_tree_f64(double a) {
    a = a + hml_dpp_f64(a, 0, 0);
    a = a + hml_dpp_f64(a, 0, 1);
    a = a + hml_dpp_f64(a, 0, 2);
    a = a + hml_dpp_f64(a, 0, 3);
    const double r0 = hml_readlane_f64(a, 0), r1 = hml_readlane_f64(a, 16), r2 = hml_readlane_f64(a, 32), r3 = hml_readlane_f64(a, 48);
    return (r0 + r1) + (r2 + r3);
}

// FB = true: the states come straight from the backward maps (q_b = S_{b+1}(entry[chunk])) and are
// written to q[] on the way; FB = false (mixture sweeps): q[] was written by the mixture kernel.
// MV = true ("-s C P D"): the floating-point sums are per emission PARAMETER; a lane's term for parameter p is the sum,
// in dimension order, of its block's statistics of the dimensions that the block's state maps to p.
template <int K, bool FB, bool MV = false>
__device__ __forceinline__ void hml_b_counts(int16_t* __restrict__ q, const uint32_t* __restrict__ starts,
                                                    const float2* __restrict__ bstat, hml_model* __restrict__ mdl,
                                                    double* __restrict__ partial /*[K][2][GROUPS]: plane (state, sum | sum of squares), one double per group*/,
                                                    const unsigned long long* __restrict__ smap,
                                                    const uint8_t* __restrict__ entry) {
    __shared__ unsigned long long h_trans[K * K];
    __shared__ unsigned long long h_occ[K];
    __shared__ double wsum[4][K][2];
    const uint32_t B = mdl->B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t g = blockIdx.x;
    // a group without a chunk (strongly compressed sweeps: B / 256 < 1024 groups) has nothing to add: its sums are the zeros
    // the tree would produce, and the workgroup leaves at once (a third of the grid on config 3 - slots that the other chains
    // of a batched launch can use)
    if (g >= (B + HML_REDUCE_CHUNK - 1u) / HML_REDUCE_CHUNK) {   // workgroup-uniform, before any barrier
        if (tid < 2 * K) partial[(uint64_t)tid * HML_REDUCE_GROUPS + g] = 0.0;
        return;
    }
    for (int i = tid; i < K * K; i += 256) h_trans[i] = 0ull;
    if (tid < K) h_occ[tid] = 0ull;
    __syncthreads();
    double acc_s[K], acc_q[K];   // this lane's terms, accumulated in block order
#pragma unroll
    for (int s = 0; s < K; ++s) { acc_s[s] = 0.0; acc_q[s] = 0.0; }
    // integer counts: every lane keeps, per state, the positions and blocks it saw in that state and the blocks that
    // stayed in it (registers, no cross-lane traffic in the loop); only changes of state - rare - go to LDS.  From
    // these: occ[s] = positions, trans[s][s] = positions - blocks + stayed, trans[p][s] (p != s) from the LDS counters.
    unsigned long long n_pos[K];
    uint32_t n_blk[K], n_stay[K];
#pragma unroll
    for (int s = 0; s < K; ++s) { n_pos[s] = 0ull; n_blk[s] = 0u; n_stay[s] = 0u; }
    const uint32_t nchunks = (B + HML_REDUCE_CHUNK - 1u) / HML_REDUCE_CHUNK;
    // with one or two chunks per workgroup (strongly compressed sweeps) the fold at the end would cost more than it
    // saves: the counts then go straight to LDS
#ifndef HML_COUNTS_DIRECT_CHUNKS
#define HML_COUNTS_DIRECT_CHUNKS (2u * HML_REDUCE_GROUPS)
#endif
    const bool direct = nchunks <= HML_COUNTS_DIRECT_CHUNKS;
    // one chunk of loads ahead
    struct in_t { unsigned long long m1, m0; uint32_t e1, e0, s1, s0; float2 v; int16_t q1, q0; };
    auto fetch = [&](uint32_t c, in_t& r) {
        const uint32_t b = c * HML_REDUCE_CHUNK + (uint32_t)tid;
        r.m1 = r.m0 = 0ull; r.e1 = r.e0 = 0u; r.s1 = r.s0 = 0u; r.v = make_float2(0.0f, 0.0f); r.q1 = r.q0 = 0;
        if (c < nchunks && b < B) {
            if (FB) {
                r.m1 = smap[b + 1]; r.e1 = entry[b / HML_BWD_CHUNK];
                if (b != 0) { r.m0 = smap[b]; r.e0 = entry[(b - 1) / HML_BWD_CHUNK]; }
            } else {
                r.q1 = q[b];
                if (b != 0) r.q0 = q[b - 1];
            }
            r.s1 = starts[b + 1]; r.s0 = starts[b];
            r.v = bstat[b];
        }
    };
    in_t nxt;
    fetch(g, nxt);
    for (uint32_t c = g; c < nchunks; c += HML_REDUCE_GROUPS) {
        const uint32_t b = c * HML_REDUCE_CHUNK + (uint32_t)tid;
        const in_t cur = nxt;
        fetch(c + HML_REDUCE_GROUPS, nxt);
        if (b < B) {
            int st, prev;
            if (FB) {
                st = (int)((cur.m1 >> (4 * cur.e1)) & 15ull);
                prev = (b == 0) ? 0 : (int)((cur.m0 >> (4 * cur.e0)) & 15ull);
                q[b] = (int16_t)st;
            } else {
                st = cur.q1;
                prev = (b == 0) ? 0 : (int)cur.q0;
            }
            const uint32_t n = cur.s1 - cur.s0;
            if (direct) {
                atomicAdd(&h_trans[st * K + st], (unsigned long long)(n - 1u));
                atomicAdd(&h_trans[prev * K + st], 1ull);
                atomicAdd(&h_occ[st], (unsigned long long)n);
            } else {
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    if (st == s) { n_pos[s] += (unsigned long long)n; n_blk[s] += 1u; if (prev == s) n_stay[s] += 1u; }
                }
                if (prev != st) atomicAdd(&h_trans[prev * K + st], 1ull);
            }
            if (MV) {
                // the term for parameter p: the block's statistics of the dimensions mapped to p, added in dimension order
                const int nD = mdl->D;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    double ts = 0.0, tq = 0.0;
                    bool any = false;
                    for (int dd = 0; dd < nD; ++dd) {
                        if (mdl->map[st][dd] == s) {
                            const float2 v2 = (dd == 0) ? cur.v : bstat[(uint64_t)dd * mdl->stat_stride + b];
                            ts = ts + (double)v2.x; tq = tq + (double)v2.y; any = true;
                        }
                    }
                    if (any) { acc_s[s] = acc_s[s] + ts; acc_q[s] = acc_q[s] + tq; }
                }
            } else {
                const double vx = (double)cur.v.x, vq = (double)cur.v.y;
#pragma unroll
                for (int s = 0; s < K; ++s)
                    if (st == s) { acc_s[s] = acc_s[s] + vx; acc_q[s] = acc_q[s] + vq; }
            }
        }
    }
    // the tree: pairwise over the 64 lanes of a wavefront, the four wavefronts in order
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const double a = hml_wave_tree_f64(acc_s[s]), d = hml_wave_tree_f64(acc_q[s]);
        if (lane == 0) { wsum[wave][s][0] = a; wsum[wave][s][1] = d; }
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
            double cs = 0.0, cq = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) { cs = cs + wsum[wv][s][0]; cq = cq + wsum[wv][s][1]; }
            partial[(uint64_t)(s * 2 + 0) * HML_REDUCE_GROUPS + g] = cs;
            partial[(uint64_t)(s * 2 + 1) * HML_REDUCE_GROUPS + g] = cq;
        }
    }
    // fold the per-lane counters: wavefront sums, then one LDS update per wavefront and state
    if (!direct) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
        unsigned long long np = n_pos[s];
        uint32_t nb = n_blk[s], ns = n_stay[s];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const uint32_t lo = __shfl_xor((uint32_t)np, m), hi = __shfl_xor((uint32_t)(np >> 32), m);
            np += ((unsigned long long)hi << 32) | lo;
            nb += __shfl_xor(nb, m);
            ns += __shfl_xor(ns, m);
        }
        if (lane == 0) {
            atomicAdd(&h_occ[s], np);
            atomicAdd(&h_trans[s * K + s], np - (unsigned long long)nb + (unsigned long long)ns);
        }
    }
    }
    __syncthreads();
    const int slot = (int)(g % HML_CNT_SPLIT);
    for (int i = tid; i < K * K; i += 256)
        if (h_trans[i]) atomicAdd(&mdl->trans[slot][i], h_trans[i]);
    if (tid < K && h_occ[tid]) atomicAdd(&mdl->occ[slot][tid], h_occ[tid]);
}
// the kernel: hml_b_counts over one chain (hml_k_many.h runs it over several chains in one launch)
template <int K, bool FB, bool MV = false>
HML_KERNEL __launch_bounds__(256) void hml_k_counts(int16_t* __restrict__ q, const uint32_t* __restrict__ starts,
                                                    const float2* __restrict__ bstat, hml_model* __restrict__ mdl,
                                                    double* __restrict__ partial ,
                                                    const unsigned long long* __restrict__ smap,
                                                    const uint8_t* __restrict__ entry) {
    hml_b_counts<K, FB, MV>(q, starts, bstat, mdl, partial, smap, entry);
}


// ------------------------------------------------------------------------------------------
// K8 for sweeps with many blocks (weakly compressed input: hundreds of 256-block chunks per workgroup), univariate: the
// same sums and the same tree as hml_k_counts, as a streaming loop - per-lane integer counters always (registers;
// changes of state - rare - in LDS), loads one chunk ahead, nothing but the lane's own accumulators in the loop.
// ------------------------------------------------------------------------------------------
#define HML_CNT_DENSE_AHEAD 2   // chunks of a workgroup per group of requests (one group is requested ahead)
template <int K, bool FB>
HML_KERNEL __launch_bounds__(256) void hml_k_counts_dense(int16_t* __restrict__ q, const uint32_t* __restrict__ starts,
                                                          const float2* __restrict__ bstat, hml_model* __restrict__ mdl,
                                                          double* __restrict__ partial /*[K][2][GROUPS]: plane (state, sum | sum of squares), one double per group*/,
                                                          const unsigned long long* __restrict__ smap,
                                                          const uint8_t* __restrict__ entry) {
    __shared__ unsigned long long h_trans[K * K];
    __shared__ unsigned long long h_occ[K];
    __shared__ double wsum[4][K][2];
    const uint32_t B = mdl->B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t g = blockIdx.x;
    for (int i = tid; i < K * K; i += 256) h_trans[i] = 0ull;
    if (tid < K) h_occ[tid] = 0ull;
    __syncthreads();
    double acc_s[K], acc_q[K];   // this lane's terms, accumulated in block order
#pragma unroll
    for (int s = 0; s < K; ++s) { acc_s[s] = 0.0; acc_q[s] = 0.0; }
    unsigned long long n_pos[K];
    uint32_t n_blk[K], n_stay[K];
#pragma unroll
    for (int s = 0; s < K; ++s) { n_pos[s] = 0ull; n_blk[s] = 0u; n_stay[s] = 0u; }
    const uint32_t nchunks = (B + HML_REDUCE_CHUNK - 1u) / HML_REDUCE_CHUNK;
    // uncompressed input (every position a block of its own: B = T): the block lengths are all 1 and starts[] is not read -
    // 4 of the pass's 14 bytes per block
    const bool unit_blocks = (B == mdl->T);
    struct in_t { unsigned long long m1, m0; uint32_t e1, e0, s1, s0; float2 v; int16_t q1, q0; };
    // The loads of a block are STRAIGHT-LINE code (round 5): the block index is clamped into the sweep instead of the loads being
    // skipped, and only the additions are predicated.  Round 3's form fetched behind `b < B` and `c + GROUPS < nchunks` - divergent
    // code, at whose end the compiler waits for every load it issued (s_waitcnt vmcnt(0) at the loop's head) - so every chunk step was
    // one whole memory round trip "however many chunks are requested ahead" (381 steps of 0.79 us).  A workgroup
    // requests several of its chunks together and adds them in chunk order: the same terms into the same accumulators in the same
    // order (the loop below).
    const uint32_t Bm1 = B ? B - 1u : 0u;
    auto fetch = [&](auto unit_c, uint32_t b, in_t& r) __attribute__((always_inline)) {
        constexpr bool UNIT = decltype(unit_c)::value;    // (a compile-time value: no branch between the loads)
        const uint32_t bc = b < Bm1 ? b : Bm1;            // (a block beyond the sweep reads the last block's words; nothing is added for it)
        const uint32_t bp = bc ? bc - 1u : 0u;
        r.m1 = r.m0 = 0ull; r.e1 = r.e0 = 0u; r.q1 = r.q0 = 0;
        if (FB) {
            r.m1 = smap[bc + 1u]; r.e1 = entry[bc / HML_BWD_CHUNK];
            r.m0 = smap[bp + 1u]; r.e0 = entry[bp / HML_BWD_CHUNK];      // (b = 0: unused)
        } else {
            r.q1 = q[bc];
            r.q0 = q[bp];
        }
        if (UNIT) { r.s1 = 1u; r.s0 = 0u; }
        else { r.s1 = starts[bc + 1u]; r.s0 = starts[bc]; }
        r.v = bstat[bc];
    };
    // A block beyond the sweep is "in no state" (nothing matches it) instead of being branched around: the compiler sank the loads of
    // a step's first chunk into that branch and waited for ALL loads there (s_waitcnt vmcnt(0)).  A state's terms stay behind a branch
    // of their own: the 64 consecutive blocks of a wavefront are mostly in one or two states (selecting the terms instead of branching
    // to them was measured: 315 against 287 us per launch at 10^8 blocks, profiles/round5_dense_counts_states.txt).
    auto add = [&](auto unit_c, uint32_t b, const in_t& cur) __attribute__((always_inline)) {
        constexpr bool UNIT = decltype(unit_c)::value;
        const bool in = b < B;
        int st, prev;
        if (FB) {
            st = (int)((cur.m1 >> (4 * cur.e1)) & 15ull);
            prev = (b == 0) ? 0 : (int)((cur.m0 >> (4 * cur.e0)) & 15ull);
        } else {
            st = cur.q1;
            prev = (b == 0) ? 0 : (int)cur.q0;
        }
        st = in ? st : -1;
        if (FB) { if (in) q[b] = (int16_t)st; }
        const uint32_t n = UNIT ? 1u : cur.s1 - cur.s0;
        const double vx = (double)cur.v.x, vq = (double)cur.v.y;
#pragma unroll
        for (int s = 0; s < K; ++s) {
            if (st == s) {
                n_pos[s] += (unsigned long long)n; n_blk[s] += 1u;
                if (prev == s) n_stay[s] += 1u;
                acc_s[s] = acc_s[s] + vx; acc_q[s] = acc_q[s] + vq;
            }
        }
        if (in && prev != st) atomicAdd(&h_trans[prev * K + st], 1ull);
    };
    // workgroup g owns chunks g, g + GROUPS, g + 2 GROUPS, ...; wavefront w streams quarter w of each of them, every lane
    // adding its block's term to its own accumulator - no tree, no barrier in the loop
    auto stream = [&](auto unit_c) __attribute__((always_inline)) {
        constexpr uint32_t STEP = HML_REDUCE_GROUPS * HML_REDUCE_CHUNK;
        // The NEXT group of HML_CNT_DENSE_AHEAD chunks is requested before this group's terms are added (same terms into the same
        // accumulators in the same order).  Two chunks a group: 253-262 us per launch at 10^8 blocks against 274-287 for round 5's four
        // chunks at once without the look-ahead; one, three, four or eight chunks a group, and two register sets that swap roles so
        // that twice as many loads stay in flight, are all slower (262-350 us) - more requests in flight cost more than they hide here.
        in_t r[HML_CNT_DENSE_AHEAD], nx[HML_CNT_DENSE_AHEAD];
        uint32_t bi[HML_CNT_DENSE_AHEAD], bn[HML_CNT_DENSE_AHEAD];
        auto index = [&](uint32_t c, uint32_t (&o)[HML_CNT_DENSE_AHEAD]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < HML_CNT_DENSE_AHEAD; ++k) {   // (blocks of chunks beyond the sweep: index B, for which nothing is added)
                const uint64_t bk = (uint64_t)c * HML_REDUCE_CHUNK + (uint32_t)tid + (uint64_t)k * STEP;
                o[k] = (c < nchunks && bk < B) ? (uint32_t)bk : B;
            }
        };
        index(g, bi);
#pragma unroll
        for (int k = 0; k < HML_CNT_DENSE_AHEAD; ++k) fetch(unit_c, bi[k], r[k]);
        for (uint32_t c = g; c < nchunks; c += (uint32_t)HML_CNT_DENSE_AHEAD * HML_REDUCE_GROUPS) {   // workgroup-uniform
            const uint64_t cn = (uint64_t)c + (uint64_t)HML_CNT_DENSE_AHEAD * HML_REDUCE_GROUPS;
            index(cn < nchunks ? (uint32_t)cn : nchunks, bn);
#pragma unroll
            for (int k = 0; k < HML_CNT_DENSE_AHEAD; ++k) fetch(unit_c, bn[k], nx[k]);
#pragma unroll
            for (int k = 0; k < HML_CNT_DENSE_AHEAD; ++k) add(unit_c, bi[k], r[k]);
#pragma unroll
            for (int k = 0; k < HML_CNT_DENSE_AHEAD; ++k) { r[k] = nx[k]; bi[k] = bn[k]; }
        }
    };
    if (unit_blocks) stream(std::true_type{}); else stream(std::false_type{});
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const double a = hml_wave_tree_f64(acc_s[s]), d = hml_wave_tree_f64(acc_q[s]);
        if (lane == 0) { wsum[wave][s][0] = a; wsum[wave][s][1] = d; }
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
            double cs = 0.0, cq = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) { cs = cs + wsum[wv][s][0]; cq = cq + wsum[wv][s][1]; }
            partial[(uint64_t)(s * 2 + 0) * HML_REDUCE_GROUPS + g] = cs;
            partial[(uint64_t)(s * 2 + 1) * HML_REDUCE_GROUPS + g] = cq;
        }
    }
    // fold the per-lane counters: wavefront sums, then one LDS update per wavefront and state
#pragma unroll
    for (int s = 0; s < K; ++s) {
        unsigned long long np = n_pos[s];
        uint32_t nb = n_blk[s], ns = n_stay[s];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const uint32_t lo = __shfl_xor((uint32_t)np, m), hi = __shfl_xor((uint32_t)(np >> 32), m);
            np += ((unsigned long long)hi << 32) | lo;
            nb += __shfl_xor(nb, m);
            ns += __shfl_xor(ns, m);
        }
        if (lane == 0) {
            atomicAdd(&h_occ[s], np);
            atomicAdd(&h_trans[s * K + s], np - (unsigned long long)nb + (unsigned long long)ns);
        }
    }
    __syncthreads();
    const int slot = (int)(g % HML_CNT_SPLIT);
    for (int i = tid; i < K * K; i += 256)
        if (h_trans[i]) atomicAdd(&mdl->trans[slot][i], h_trans[i]);
    if (tid < K && h_occ[tid]) atomicAdd(&mdl->occ[slot][tid], h_occ[tid]);
}

// ------------------------------------------------------------------------------------------
// K10 marginals_accumulate - Records::record(state, N) + StateMarginals::addRecord (reference
// src/Records.hpp:155-235, src/StateMarginals.hpp:51-137).  Adjacent blocks in the same state form
// one segment; every segment adds one count for its state to all of its positions, and the
// marginals file is cut wherever any recorded sweep had a segment boundary.  Device form: +1/-1
// into a per-state difference array at segment starts and a boundary bit; one thread per block.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void hml_b_record(const int16_t* __restrict__ q, const uint32_t* __restrict__ starts,
                                                    hml_model* __restrict__ mdl, int32_t* __restrict__ diff,
                                                    uint32_t* __restrict__ boundary) {
    if (mdl->halted != 0u) return;   // (hml_state.h: the sweep did not happen)
    const uint32_t B = mdl->B;
    const uint64_t T1 = (uint64_t)mdl->T + 1u;
    const uint32_t stride = gridDim.x * blockDim.x;
    int mx = -1;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += stride) {
        const int st = q[b];
        const int prev = (b == 0) ? -1 : (int)q[b - 1];
        if (st != prev) {
            const uint32_t t = starts[b];
            // (plain read-modify-writes: cell (state, t) belongs to the one block that starts at t, and st != prev - on
            // uncompressed input a sweep has 10^7 changes of state, and atomics to as many random lines cost 3 ms)
            diff[(uint64_t)st * T1 + t] += 1;
            if (prev >= 0) diff[(uint64_t)prev * T1 + t] -= 1;
            atomicOr(&boundary[t >> 5], 1u << (t & 31u));
            mx = st > mx ? st : mx;
        }
    }
    if (mx >= 0) atomicMax(&mdl->max_state_recorded, mx);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long n = atomicAdd(&mdl->n_recorded, 1ull);
        if (n >= 0x7fffffffull) hml_raise(mdl, HML_DEVERR_TOO_MANY_RECORDS, 0.0f);
    }
}
// the kernel: hml_b_record over one chain (hml_k_many.h runs it over several chains in one launch)
HML_KERNEL __launch_bounds__(256) void hml_k_record(const int16_t* __restrict__ q, const uint32_t* __restrict__ starts,
                                                    hml_model* __restrict__ mdl, int32_t* __restrict__ diff,
                                                    uint32_t* __restrict__ boundary) {
    hml_b_record(q, starts, mdl, diff, boundary);
}


#endif
