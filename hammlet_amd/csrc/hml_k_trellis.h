// The trellis of a weakly compressed sweep (millions of blocks) in ONE pass over the blocks: emission terms (K6a), the
// forward filter (K6b) and the backward candidate maps (K7) fused, with the per-block vectors living in LDS only.
// Reference: StateSequence<ForwardBackward>::sample, src/StateSequence/ForwardBackward.hpp:67-162 (emission terms :67-84,
// forward recursion :86-123, backward sampling :133-162), Trellis::sample src/Trellis.hpp:61-66.
//
// Why: with B ~ T the separate kernels of the strongly compressed sweep move 4K-float vectors per block through HBM three
// times (emission terms and rescale factors out, in again for the filter, rows out, in again for the maps: ~180 B per
// block at K = 5) and the filter pays (W + L) / L = 7 times its work for the warm-up its chunks of L = 16 need.  Here
//   * a WAVEFRONT owns 64 forward chunks of L consecutive blocks (L = 32 ... 256, by the host: the more blocks, the longer),
//     one chunk per lane for the sequential filter.  It walks its chunks in batches of R = 8 rows: all 64 lanes compute
//     the emission terms of the batch's 512 blocks into LDS (lane = (chunk, row): runs of 8 consecutive blocks per
//     chunk), then every lane advances its own chunk's filter by the 8 rows, overwriting the terms with the normalised
//     rows, then all lanes turn the 512 rows into candidate maps, then every lane folds its 8 maps into its chunk map.
//     No workgroup barrier, no idle wavefront, 14 KB of LDS per wavefront at K = 5.  Per block 4 + 16 bytes are read
//     (block start, integral-array gathers) and 8 + 8 written (statistics for the count pass, candidate map);
//   * the warm-up (W <= HML_TRE_HALO blocks before the chunk, emission terms included) is short and paid once per L
//     blocks; chunks whose filter had not forgotten its start by then are REFITTED IN PARALLEL (hml_k_trellis_refit: one
//     wavefront per stale chunk, starting from its predecessor's end vector; a few rounds - hml_ctx.hpp: tre_refit_rounds -, each verified again) instead
//     of lengthening everybody's warm-up until nobody fails; runs of more consecutive stale chunks than there are rounds are
//     finished sequentially (hml_k_trellis_serial).  As in hml_k_forward.h the rows are bit for bit those of the sequential recursion: a chunk
//     is accepted only when the vector it started from equals, bit for bit, the vector its predecessor ended in, and
//     induction from chunk 0 (which starts from pi) does the rest.
// Candidate maps instead of suffix-composed maps: row t's map cand_t (successor state -> sampled state) is stored as it
// is; a chunk's states follow from the state entering it by walking its L maps (hml_k_trellis_states), so a refitted
// chunk touches nothing outside itself.
#ifndef HML_K_TRELLIS_H
#define HML_K_TRELLIS_H

#include <type_traits>

#include "hml_k_backward.h"
#include "hml_k_forward.h"

#define HML_TRE_CKPT_ROWS 32  // rows between the first pass's checkpoints = rows per batch of a refit (a power of two, 32 or 64; chunk lengths are multiples of 32)
#define HML_TRE_NCH 64      // forward chunks per wavefront = lanes of the filter
#define HML_TRE_R 4         // rows per batch
#define HML_TRE_HALO HML_TRE_HALO_MAX   // longest warm-up of the first pass (a multiple of HML_TRE_R and of 16; hml_state.h)
#define HML_TRE_GTAB 64     // block sizes whose rescale factors expf((N-1) logA_s) come from a table
#define HML_TRE_MIN_L 32    // shortest chunk (a multiple of 32; per-chunk arrays are sized for it)
#define HML_TRE_MAX_L 1024  // longest chunk

// warm-up of the first pass: the chain's adaptive warm-up, capped and rounded up to whole batches
__device__ __forceinline__ uint32_t hml_tre_warmup(const hml_model* mdl) {
    const uint32_t w = (mdl->fwd_W < (uint32_t)HML_TRE_HALO) ? mdl->fwd_W : (uint32_t)HML_TRE_HALO;
    return (w + (uint32_t)HML_TRE_R - 1u) / (uint32_t)HML_TRE_R * (uint32_t)HML_TRE_R;
}

// candidate maps in memory: 4 bits per state, so 32 bits hold a map of up to 8 states (half the traffic of the first pass's
// stores and of the states kernel's loads); 64 bits beyond
template <int K>
struct hml_tre_map {
    typedef typename std::conditional<(K <= 8), uint32_t, unsigned long long>::type stored;
};
template <int K>
__device__ __forceinline__ void hml_tre_store_cand(unsigned long long* __restrict__ cand, uint32_t t, unsigned long long m) {
    reinterpret_cast<typename hml_tre_map<K>::stored*>(cand)[t] = (typename hml_tre_map<K>::stored)m;
}
template <int K>
__device__ __forceinline__ unsigned long long hml_tre_load_cand(const unsigned long long* __restrict__ cand, uint64_t t) {
    return (unsigned long long)reinterpret_cast<const typename hml_tre_map<K>::stored*>(cand)[t];
}

// E_s and e_s = expf(E_s - max E) of one block (hml_emit_compute without the rescale factors)
template <int K>
__device__ __forceinline__ void hml_tre_emit(const hml_emit_params<K>& p, hml_model* mdl, float sx, float sq, float N, float (&E)[K], float (&ev)[K],
                                             const uint64_t* exp_tab = HML_EXP2F_TAB) {
    float maxE = -3.40282346638528859812e+38f;   // numeric_limits<float>::lowest()
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const float ip = hml_inner_product(p.mu[s], p.var[s], p.rvar[s], sx, sq);
        if (!hml_isfinite(ip)) hml_raise(mdl, HML_DEVERR_IP_NOT_FINITE, ip);
        float e = (0.0f + ip) - N * p.logN[s];
        if (p.self) e += (N - 1.0f) * p.logA[s];
        E[s] = e;
        maxE = (e < maxE) ? maxE : e;
    }
#pragma unroll
    for (int s = 0; s < K; ++s) ev[s] = hml_expf_tab(E[s] - maxE, exp_tab);
}

// candidate map of row t from its (rescaled) row r and the row's uniform u (hml_cat_uniform): cand(x) = draw of
// Cat(r_i A(i, x)); the last row's map is constant
template <int K>
__device__ __forceinline__ unsigned long long hml_tre_cand_u(const float (&r)[K], const hml_amat<K>& A, hml_model* mdl, uint32_t t,
                                                             uint32_t B, const double u) {
    unsigned long long map = 0ull;
    // "Negative backward variable!" (ForwardBackward.hpp:147-149): the products r_i A(i, x) below are negative exactly
    // when r_i is (A holds probabilities), so the row is checked once instead of K times
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
            for (int i = 0; i < K; ++i) w[i] = r[i] * A[i * K + x];
            map |= (unsigned long long)hml_categorical_k<K>(w, u) << (4 * x);
        }
    }
    return map;
}
template <int K>
__device__ __forceinline__ unsigned long long hml_tre_cand(const float (&r)[K], const hml_amat<K>& A, hml_model* mdl, uint32_t t,
                                                           uint32_t B, unsigned long long epoch, const hml_key key) {
    return hml_tre_cand_u<K>(r, A, mdl, t, B, hml_cat_uniform(key, epoch, t));
}

// ------------------------------------------------------------------------------------------
// the first pass: one wavefront (= one workgroup of 64 threads) per 64 chunks
// ------------------------------------------------------------------------------------------
template <int K>
HML_KERNEL __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(K <= 6 ? 4 : 1, K <= 6 ? 4 : 8))) void hml_k_trellis_tile(const float2* __restrict__ ia, const uint32_t* __restrict__ starts,
                                                         hml_model* __restrict__ mdl, const hml_model* __restrict__ mdl_ro,
                                                         float2* __restrict__ bstat, unsigned long long* __restrict__ cand,
                                                         unsigned long long* __restrict__ fmap, float* __restrict__ entry,
                                                         float* __restrict__ exitv, uint32_t* __restrict__ fb_count,
                                                         float* __restrict__ eprobe, float* __restrict__ aprobe, uint32_t L) {
    constexpr int R = HML_TRE_R, PITCH = HML_TRE_R + 1, PLANE = HML_TRE_NCH * PITCH;
    __shared__ float sm_v[K * PLANE];                          // e_s, then alpha_s in place: [s][chunk][row of the batch], chunk pitch R + 1
    __shared__ unsigned long long sm_c[HML_TRE_NCH * PITCH];   // the batch's candidate maps
    // the batch's block sizes (0: no block in this slot).  Up to 8 states they need no array of their own: between P1a and
    // P1b a size waits in plane 0 of sm_v (free until P1b writes the terms), from P1b to P3 in the upper half of the slot's
    // sm_c word (the statistics are consumed by then, a map of <= 8 states takes the lower half) - 10 KB of LDS per
    // wavefront instead of 11.5, i.e. 16 wavefronts per CU instead of 13
    constexpr bool PACKN = (K <= 8);
    __shared__ uint32_t sm_n[PACKN ? 1 : HML_TRE_NCH * PITCH];
    __shared__ float gtab[HML_TRE_GTAB * K];
    const uint32_t B = mdl_ro->B;
    const int lane = threadIdx.x;
    // one wavefront per workgroup: the phases below hand data to each other through LDS and rely on the in-order LDS traffic
    // of ONE wavefront (no barriers).  __launch_bounds__ only bounds the shape; another one is reported, not run.
    if (blockDim.x != 64u) { if (threadIdx.x == 0) hml_raise(mdl, HML_DEVERR_LAUNCH_GEOMETRY, (float)blockDim.x); return; }
    hml_emit_params<K> p;
    hml_emit_load<K>(p, mdl_ro, 0);
    const uint32_t Wt = hml_tre_warmup(mdl_ro);
    const unsigned long long epoch = mdl_ro->epoch;
    const hml_key key = mdl_ro->key;
    for (int i = lane; i < HML_TRE_GTAB * K; i += 64) gtab[i] = hml_expf(((float)(i / K + 1) - 1.0f) * mdl_ro->logA[i % K]);   // N = i / K + 1
    __shared__ float sm_A[hml_amat<K>::LDS_FLOATS];
    hml_amat_fill<K>(sm_A, mdl_ro, lane, 64);
    hml_fwd_ctx<K> cx;
    hml_fwd_ctx_load<K>(cx, mdl_ro, sm_A);
    if (blockIdx.x == 0 && lane < K && aprobe) aprobe[lane] = mdl_ro->pi[lane];
    const uint32_t C = (B + L - 1u) / L;
    const uint32_t n_groups = (C + (uint32_t)HML_TRE_NCH - 1u) / (uint32_t)HML_TRE_NCH;
    __syncthreads();
    for (uint32_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const uint32_t f0 = grp * (uint32_t)HML_TRE_NCH;
        // this lane's own chunk (the filter's view)
        const uint32_t f = f0 + (uint32_t)lane;
        const long long first = (long long)f * L;
        const bool active = f < C;
        const long long last = active ? ((first + L < (long long)B) ? first + L : (long long)B) : first;
        const long long ws = (first >= (long long)Wt) ? first - (long long)Wt : 0ll;
        float alpha[K];
#pragma unroll
        for (int s = 0; s < K; ++s) alpha[s] = (ws == 0ll) ? mdl_ro->pi[s] : cx.invK;
        uint32_t nfb = 0u;
        unsigned long long cmap = HML_MAP_IDENTITY;
        for (int rel0 = -(int)Wt; rel0 < (int)L; rel0 += R) {   // wave-uniform
            // ---------------- P1: emission terms of the batch, lane = (chunk, row): slot k * 64 + lane
            // P1a, unrolled: block starts and integral-array gathers of all eight slots in flight together; the block
            // statistics wait in LDS (the lane's own slots of sm_c / sm_n, free until P3)
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int slot = k * 64 + lane;
                const int c = slot / R, r = slot % R;
                const long long cf = (long long)(f0 + (uint32_t)c) * L;
                const long long b = cf + rel0 + r;
                const long long cl = (cf + L < (long long)B) ? cf + L : (long long)B;
                uint32_t nb = 0u;
                float sx = 0.0f, sq = 0.0f;
                if (f0 + (uint32_t)c < C && b >= 0ll && b < cl) {
                    const uint32_t st = starts[b], en = starts[b + 1];
                    hml_block_stats_one(ia, st, en, sx, sq);
                    nb = en - st;
                }
                if (PACKN) sm_v[c * PITCH + r] = hml_u2f(nb); else sm_n[c * PITCH + r] = nb;
                sm_c[c * PITCH + r] = ((unsigned long long)hml_f2u(sq) << 32) | hml_f2u(sx);
            }
            // P1b, one copy of the arithmetic (eight unrolled copies here and in P3 made 100 KB of code)
#pragma unroll 1
            for (int k = 0; k < R; ++k) {
                const int slot = k * 64 + lane;
                const int c = slot / R, r = slot % R;
                const uint32_t nb = PACKN ? hml_f2u(sm_v[c * PITCH + r]) : sm_n[c * PITCH + r];
                const unsigned long long pk = sm_c[c * PITCH + r];
                if (PACKN) sm_c[c * PITCH + r] = (unsigned long long)nb << 32;
                if (nb != 0u) {
                    const float sx = hml_u2f((uint32_t)pk), sq = hml_u2f((uint32_t)(pk >> 32));
                    float E[K], ev[K];
                    hml_tre_emit<K>(p, mdl, sx, sq, (float)nb, E, ev);
#pragma unroll
                    for (int s = 0; s < K; ++s) sm_v[s * PLANE + c * PITCH + r] = ev[s];
                    if (rel0 >= 0) {
                        const long long b = (long long)(f0 + (uint32_t)c) * L + rel0 + r;
                        bstat[b] = make_float2(sx, sq);
                        if (eprobe) {
#pragma unroll
                            for (int s = 0; s < K; ++s) eprobe[(uint64_t)b * K + s] = E[s];
                        }
                    }
                }
            }
            __syncthreads();
            // ---------------- P2: every lane advances its own chunk's filter over the batch's rows
            if (rel0 == 0 && active) {
#pragma unroll
                for (int s = 0; s < K; ++s) entry[(uint64_t)f * K + s] = alpha[s];
            }
            if (active) {
#pragma unroll 1
                for (int r = 0; r < R; ++r) {
                    const long long b = first + rel0 + r;
                    if (b < ws || b >= last) continue;
                    float e[K];
#pragma unroll
                    for (int s = 0; s < K; ++s) e[s] = sm_v[s * PLANE + lane * PITCH + r];
                    const bool fb = hml_fwd_step<K>(cx, alpha, e);
                    if (rel0 >= 0) {
                        if (fb) nfb++;
#pragma unroll
                        for (int s = 0; s < K; ++s) {
                            sm_v[s * PLANE + lane * PITCH + r] = alpha[s];
                            if (aprobe) aprobe[(uint64_t)(b + 1) * K + s] = alpha[s];
                        }
                    }
                }
            }
            __syncthreads();
            if (rel0 >= 0) {
                // ---------------- P3: candidate maps of the batch's rows, lane = (chunk, pair of rows): the two rows of a
                // pair (blocks 2m, 2m + 1: chunks and batches start at even blocks) share one Philox block
                static_assert(HML_TRE_R % 2 == 0 && HML_TRE_MIN_L % 2 == 0, "row pairs");
#pragma unroll 1
                for (int k = 0; k < R / 2; ++k) {
                    const int slot = k * 64 + lane;
                    const int c = slot / (R / 2), r0 = 2 * (slot % (R / 2));
                    const uint32_t n1 = PACKN ? (uint32_t)(sm_c[c * PITCH + r0 + 1] >> 32) : sm_n[c * PITCH + r0 + 1];   // (before row r0's map lands next to it)
                    const uint32_t n0 = PACKN ? (uint32_t)(sm_c[c * PITCH + r0] >> 32) : sm_n[c * PITCH + r0];
                    if (n0 != 0u) {   // (a block of the batch, see P1; the pair's second row exists only behind its first)
                        const uint32_t b0 = (f0 + (uint32_t)c) * L + (uint32_t)rel0 + (uint32_t)r0;
                        double u2[2];
                        hml_cat_uniform_pair(key, epoch, b0 >> 1, u2[0], u2[1]);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const uint32_t n = h ? n1 : n0;
                            if (n != 0u) {
                                const uint32_t t = b0 + (uint32_t)h + 1u;
                                float row[K];
#pragma unroll
                                for (int s = 0; s < K; ++s) row[s] = sm_v[s * PLANE + c * PITCH + r0 + h];
                                if (cx.self && t < B) {   // the reference rescales row t after step t + 1 has consumed it (ForwardBackward.hpp:115-119)
#pragma unroll
                                    for (int s = 0; s < K; ++s)
                                        row[s] = row[s] * ((n <= (uint32_t)HML_TRE_GTAB) ? gtab[(n - 1u) * K + s] : hml_expf(((float)n - 1.0f) * p.logA[s]));
                                }
                                const unsigned long long cm = hml_tre_cand_u<K>(row, cx.A, mdl, t, B, u2[h]);
                                hml_tre_store_cand<K>(cand, t, cm);
                                sm_c[c * PITCH + r0 + h] = cm;
                            }
                        }
                    }
                }
                __syncthreads();
                // ---------------- P4: every lane folds its chunk's maps of the batch into the chunk map
                if (active) {
#pragma unroll 1
                    for (int r = 0; r < R; ++r) {
                        const long long b = first + rel0 + r;
                        if (b >= last) break;
                        cmap = hml_map_compose<K>(cmap, sm_c[lane * PITCH + r]);
                    }
                }
                // (no barrier here: the next batch's P1a writes sm_c - and plane 0 of sm_v - straight behind these reads, and the
                // warm-up batches go from P2's barrier straight into P1a.  That is sound because the workgroup IS one
                // wavefront - launch_bounds(64), asserted by the host's launch - whose LDS instructions execute in order.)
            }
        }
        if (active) {
#pragma unroll
            for (int s = 0; s < K; ++s) exitv[(uint64_t)f * K + s] = alpha[s];
            fb_count[f] = nfb;
            if (nfb) atomicAdd(&mdl->uniform_fallbacks, (unsigned long long)nfb);
            fmap[f] = cmap;
        }
        __syncthreads();
    }
}

// a forward chunk that started from pi itself (its warm-up window reaches block 0): exact by construction
__device__ __forceinline__ bool hml_tre_exact(uint32_t f, uint32_t L, uint32_t Wt) { return f == 0u || (uint64_t)f * L <= (uint64_t)Wt; }

// ------------------------------------------------------------------------------------------
// verification: entry[f] == exit[f-1], bit for bit; stale chunks go on a list
// ------------------------------------------------------------------------------------------
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_k_trellis_verify(hml_model* __restrict__ mdl, const float* __restrict__ entry,
                                                            const float* __restrict__ exitv, uint32_t* __restrict__ list, uint32_t L) {
    const uint32_t B = mdl->B;
    const uint32_t C = (B + L - 1u) / L;
    const uint32_t Wt = hml_tre_warmup(mdl);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < C; f += stride) {
        if (hml_tre_exact(f, L, Wt)) continue;
        bool same = true;
#pragma unroll
        for (int s = 0; s < K; ++s) same = same && (hml_f2u(entry[(uint64_t)f * K + s]) == hml_f2u(exitv[(uint64_t)(f - 1u) * K + s]));
        if (!same) list[atomicAdd(&mdl->fwd_mismatch, 1u)] = f;
    }
}

// after a refit round: the refitted chunks and their successors are checked again; what is (still or newly) inconsistent
// goes on the other list, once (tag[f] == gen marks chunks already listed in this round)
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_k_trellis_verify_list(hml_model* __restrict__ mdl, const float* __restrict__ entry,
                                                                 const float* __restrict__ exitv, const uint32_t* __restrict__ list_in,
                                                                 uint32_t* __restrict__ list_out, uint32_t* __restrict__ tag,
                                                                 int in_is_a, uint32_t round, uint32_t L) {
    const uint32_t B = mdl->B;
    const uint32_t C = (B + L - 1u) / L;
    const uint32_t Wt = hml_tre_warmup(mdl);
    const uint32_t n_in = in_is_a ? mdl->fwd_mismatch : mdl->fwd_mismatch2;
    uint32_t* n_out = in_is_a ? &mdl->fwd_mismatch2 : &mdl->fwd_mismatch;
    const uint32_t gen = ((uint32_t)mdl->epoch << 3) + round + 1u;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < 2u * n_in; i += stride) {
        const uint32_t f = list_in[i >> 1] + (i & 1u);
        if (f >= C || hml_tre_exact(f, L, Wt)) continue;
        bool same = true;
#pragma unroll
        for (int s = 0; s < K; ++s) same = same && (hml_f2u(entry[(uint64_t)f * K + s]) == hml_f2u(exitv[(uint64_t)(f - 1u) * K + s]));
        if (!same && atomicExch(&tag[f], gen) != gen) list_out[atomicAdd(n_out, 1u)] = f;
    }
}

// ------------------------------------------------------------------------------------------
// One forward chunk recomputed by one lane, sequentially: emission terms, filter, rescaled rows, candidate maps, chunk map.
// from_exact = false: warm-up over `Wl` blocks from the uniform vector (or from pi when the window reaches block 0);
// from_exact = true: start from `alpha` (the predecessor's verified end vector).  Same arithmetic, in the same order, as
// the first pass - so a refitted chunk is what the first pass would have produced from that start.
// ------------------------------------------------------------------------------------------
template <int K>
__device__ void hml_tre_chunk_sequential(const hml_emit_params<K>& p, const hml_fwd_ctx<K>& cx, const float2* __restrict__ ia,
                                         const uint32_t* __restrict__ starts, hml_model* __restrict__ mdl,
                                         unsigned long long* __restrict__ cand, unsigned long long* __restrict__ fmap,
                                         float* __restrict__ entry, float* __restrict__ exitv, uint32_t* __restrict__ fb_count,
                                         float* __restrict__ eprobe, float* __restrict__ aprobe, uint32_t f, uint32_t Wl,
                                         bool from_exact, float (&alpha)[K], unsigned long long epoch, const hml_key key, uint32_t L) {
    const uint32_t B = cx.B;
    const uint32_t first = f * L;
    const uint32_t last = (first + L < B) ? first + L : B;
    if (!from_exact) {
        const uint32_t ws = (first >= Wl) ? first - Wl : 0u;
#pragma unroll
        for (int s = 0; s < K; ++s) alpha[s] = (ws == 0u) ? mdl->pi[s] : cx.invK;
        for (uint32_t b = ws; b < first; ++b) {
            const uint32_t st = starts[b], en = starts[b + 1];
            float sx, sq, E[K], e[K];
            hml_block_stats_one(ia, st, en, sx, sq);
            hml_tre_emit<K>(p, mdl, sx, sq, (float)(en - st), E, e);
            hml_fwd_step<K>(cx, alpha, e);
        }
    }
#pragma unroll
    for (int s = 0; s < K; ++s) entry[(uint64_t)f * K + s] = alpha[s];
    uint32_t nfb = 0u;
    unsigned long long m = HML_MAP_IDENTITY;
    for (uint32_t b = first; b < last; ++b) {
        const uint32_t st = starts[b], en = starts[b + 1];
        float sx, sq, E[K], e[K];
        hml_block_stats_one(ia, st, en, sx, sq);
        hml_tre_emit<K>(p, mdl, sx, sq, (float)(en - st), E, e);
        if (hml_fwd_step<K>(cx, alpha, e)) nfb++;
        const uint32_t t = b + 1u;
        float row[K];
#pragma unroll
        for (int s = 0; s < K; ++s) {
            if (eprobe) eprobe[(uint64_t)b * K + s] = E[s];
            if (aprobe) aprobe[(uint64_t)t * K + s] = alpha[s];
            row[s] = alpha[s];
        }
        if (cx.self && t < B) {
            const uint32_t n = en - st;
#pragma unroll
            for (int s = 0; s < K; ++s) row[s] = row[s] * hml_expf(((float)n - 1.0f) * p.logA[s]);   // (the table holds this very value)
        }
        const unsigned long long cm = hml_tre_cand<K>(row, cx.A, mdl, t, B, epoch, key);
        hml_tre_store_cand<K>(cand, t, cm);
        m = hml_map_compose<K>(m, cm);
    }
    fmap[f] = m;
#pragma unroll
    for (int s = 0; s < K; ++s) exitv[(uint64_t)f * K + s] = alpha[s];
    const uint32_t old = fb_count[f];
    fb_count[f] = nfb;
    // keep the global tally of uniform fallbacks consistent (two's-complement delta on the unsigned counter)
    if (nfb != old) atomicAdd(&mdl->uniform_fallbacks, (unsigned long long)(long long)((int)nfb - (int)old));
}

// Refit round: every listed chunk again, STARTING FROM ITS PREDECESSOR'S END VECTOR - no warm-up at all.  A stale chunk is
// nearly always isolated: its predecessor was verified against its own predecessor, so that end vector is the true one and
// the refitted chunk is final.  If the predecessor was on the list too (and changes in this very round) the verification
// behind the round notices (entry != the predecessor's new end vector) and the chunk comes back in the next round: a run
// of k consecutive stale chunks takes k rounds, longer runs are left to the sequential finisher.
// One wavefront (= workgroup) per chunk: its 64 lanes compute emission terms and candidate maps of 64 blocks at a time,
// lane 0 runs the filter over them in between (the same arithmetic, in the same order, as the first pass).
template <int K>
HML_KERNEL __launch_bounds__(64) void hml_k_trellis_refit(const float2* __restrict__ ia, const uint32_t* __restrict__ starts,
                                                          hml_model* __restrict__ mdl, unsigned long long* __restrict__ cand,
                                                          unsigned long long* __restrict__ fmap, float* __restrict__ entry,
                                                          float* __restrict__ exitv, uint32_t* __restrict__ fb_count,
                                                          float* __restrict__ eprobe, float* __restrict__ aprobe,
                                                          const uint32_t* __restrict__ list, int list_is_a,
                                                          uint32_t* __restrict__ ckpt, uint32_t L) {
    __shared__ float sm_e[K * 65];   // the batch's emission terms (all lanes) ...
    __shared__ float sm_a[K * 65];   // ... and its rows (lane 0)
    __shared__ uint32_t sm_met[2];   // lane 0 met the checkpoint again | the fallbacks the old rows had counted up to it
    const uint32_t n = list_is_a ? mdl->fwd_mismatch : mdl->fwd_mismatch2;
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // the other list is written next (hml_k_trellis_verify_list): empty it
        if (list_is_a) mdl->fwd_mismatch2 = 0u; else mdl->fwd_mismatch = 0u;
    }
    if (n == 0u) return;
    // (one wavefront per workgroup, like hml_k_trellis_tile: sm_e / sm_a are exchanged between its lanes)
    if (blockDim.x != 64u) { if (threadIdx.x == 0) hml_raise(mdl, HML_DEVERR_LAUNCH_GEOMETRY, (float)blockDim.x); return; }
    const int lane = threadIdx.x;
    hml_emit_params<K> p;
    hml_emit_load<K>(p, mdl, 0);
    __shared__ float sm_A[hml_amat<K>::LDS_FLOATS];
    hml_amat_fill<K>(sm_A, mdl, lane, (int)blockDim.x);
    hml_fwd_ctx<K> cx;
    hml_fwd_ctx_load<K>(cx, mdl, sm_A);
    const uint32_t B = cx.B;
    const uint32_t C = (B + L - 1u) / L;
    const uint32_t Wt = hml_tre_warmup(mdl);
    const unsigned long long epoch = mdl->epoch;
    const hml_key key = mdl->key;
    constexpr uint32_t CKR = HML_TRE_CKPT_ROWS;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {   // workgroup-uniform
        const uint32_t f = list[i];
        if (hml_tre_exact(f, L, Wt)) continue;
        const uint32_t first = f * L;
        const uint32_t last = (first + L < B) ? first + L : B;
        float alpha[K];
#pragma unroll
        for (int s = 0; s < K; ++s) alpha[s] = hml_u2f(hml_ld_bits_coherent(exitv + (uint64_t)(f - 1u) * K + s));
        if (lane == 0) {
#pragma unroll
            for (int s = 0; s < K; ++s) entry[(uint64_t)f * K + s] = alpha[s];
        }
        uint32_t nfb = 0u;
        unsigned long long m = HML_MAP_IDENTITY;
        bool stopped = false;
        for (uint32_t bb = first; bb < last; bb += CKR) {   // batches of CKR rows (lanes CKR ... 63 idle): a refit stops at the first checkpoint it meets
            const uint32_t b = bb + (uint32_t)lane;
            const bool mine = (uint32_t)lane < CKR && b < last;
            uint32_t nb = 0u;
            if (mine) {
                const uint32_t st = starts[b], en = starts[b + 1];
                float sx, sq, E[K], ev[K];
                hml_block_stats_one(ia, st, en, sx, sq);
                hml_tre_emit<K>(p, mdl, sx, sq, (float)(en - st), E, ev);
                nb = en - st;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    sm_e[s * 65 + lane] = ev[s];
                    if (eprobe) eprobe[(uint64_t)b * K + s] = E[s];
                }
            }
            __syncthreads();
            if (lane == 0) {
                // (the rows go to an array of their own: the reads of the next rows' terms do not wait behind these stores, and the
                // loop is unrolled so that they are requested ahead of the step that needs them)
                const uint32_t nrows = (last - bb < CKR) ? last - bb : CKR;
#pragma unroll 4
                for (uint32_t l = 0; l < nrows; ++l) {
                    float e[K];
#pragma unroll
                    for (int s = 0; s < K; ++s) e[s] = sm_e[s * 65 + l];
                    if (hml_fwd_step<K>(cx, alpha, e)) nfb++;
#pragma unroll
                    for (int s = 0; s < K; ++s) sm_a[s * 65 + l] = alpha[s];
                    if (aprobe) {
#pragma unroll
                        for (int s = 0; s < K; ++s) aprobe[(uint64_t)(bb + l + 1u) * K + s] = alpha[s];
                    }
                }
                // The first pass left its forward vector after every HML_TRE_CKPT_ROWS rows of the chunk (hml_k_trellis_rows).  Where the
                // refitted filter meets it again, bit for bit, every later row of the chunk - emission terms, filter,
                // uniforms, candidate maps - is what it was: the refit stops there.  (A filter that started a few rows
                // too early to have forgotten its start has nearly always done so 32 rows later: checkpoints every 32 rows since the
                // end of round 5 - every 64 before, and a refit's lane 0 filtered 64 rows where it now filters 32.)
                uint32_t met = 0u, old_nfb = 0u;
                if (ckpt && bb + CKR < last) {
                    uint32_t* const ck = ckpt + (uint64_t)((bb - first) / CKR) * (uint32_t)(K + 1) * C + f;
                    met = 1u;
#pragma unroll
                    for (int s = 0; s < K; ++s) {
                        if (ck[(uint64_t)s * C] != hml_f2u(alpha[s])) met = 0u;
                        ck[(uint64_t)s * C] = hml_f2u(alpha[s]);
                    }
                    old_nfb = ck[(uint64_t)K * C];
                    ck[(uint64_t)K * C] = nfb;
                }
                sm_met[0] = met;
                sm_met[1] = old_nfb;
            }
            __syncthreads();
            unsigned long long mb = HML_MAP_IDENTITY;
            if (mine) {
                const uint32_t t = b + 1u;
                float row[K];
#pragma unroll
                for (int s = 0; s < K; ++s) row[s] = sm_a[s * 65 + lane];
                if (cx.self && t < B) {
#pragma unroll
                    for (int s = 0; s < K; ++s) row[s] = row[s] * hml_expf(((float)nb - 1.0f) * p.logA[s]);   // (the first pass's table holds this very value)
                }
                mb = hml_tre_cand<K>(row, cx.A, mdl, t, B, epoch, key);
                hml_tre_store_cand<K>(cand, t, mb);
            }
            // the batch's maps composed in row order across the lanes (composition is associative: the same map as one lane's walk)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                unsigned long long o = hml_shfl_down_u64(mb, d);
                if (lane + d >= 64) o = HML_MAP_IDENTITY;
                mb = hml_map_compose<K>(mb, o);
            }
            m = hml_map_compose<K>(m, mb);   // (lane 0's m is the chunk's map so far)
            const bool met = sm_met[0] != 0u;
            __syncthreads();
            if (met) {   // workgroup-uniform
                // the rest of the chunk stands: its maps (in memory) complete the chunk map, its end vector and the fallbacks
                // counted behind the checkpoint stay
                // (all of them requested before the first is composed - straight-line loads from clamped rows: one memory round trip
                // for the rest of the chunk instead of one per 64 rows)
                constexpr int NB = HML_TRE_MAX_L / 64;
                unsigned long long rest[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const uint64_t b = (uint64_t)bb + CKR + (uint32_t)(j * 64 + lane);
                    rest[j] = hml_tre_load_cand<K>(cand, (b < last ? b : (uint64_t)last - 1u) + 1u);
                    if (!(b < last)) rest[j] = HML_MAP_IDENTITY;
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if ((uint64_t)bb + CKR + (uint32_t)(j * 64) < last) {   // workgroup-uniform
                        unsigned long long mp = rest[j];
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) {
                            unsigned long long o = hml_shfl_down_u64(mp, d);
                            if (lane + d >= 64) o = HML_MAP_IDENTITY;
                            mp = hml_map_compose<K>(mp, o);
                        }
                        m = hml_map_compose<K>(m, mp);   // (lane 0 holds the product of the 64 maps in row order)
                    }
                }
                stopped = true;
                break;
            }
        }
        if (lane == 0) {
            fmap[f] = m;
            if (stopped) nfb += fb_count[f] - sm_met[1];
            else {
#pragma unroll
                for (int s = 0; s < K; ++s) exitv[(uint64_t)f * K + s] = alpha[s];
            }
            const uint32_t old = fb_count[f];
            fb_count[f] = nfb;
            if (nfb != old) atomicAdd(&mdl->uniform_fallbacks, (unsigned long long)(long long)((int)nfb - (int)old));
            atomicAdd(&mdl->forward_refits, 1ull);
        }
    }
}

// what the refit rounds left inconsistent: marked in a bitmap, visited in increasing order by ONE lane, each recomputed
// from its predecessor's true end vector; the walk follows a chain while the recomputed end vector makes the next chunk
// inconsistent.  After this pass induction from chunk 0 holds.  One workgroup of 256 threads.
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_k_trellis_serial(const float2* __restrict__ ia, const uint32_t* __restrict__ starts,
                                                            hml_model* __restrict__ mdl, unsigned long long* __restrict__ cand,
                                                            unsigned long long* __restrict__ fmap, float* __restrict__ entry,
                                                            float* __restrict__ exitv, uint32_t* __restrict__ fb_count,
                                                            float* __restrict__ eprobe, float* __restrict__ aprobe,
                                                            const uint32_t* __restrict__ list, int list_is_a, uint32_t* __restrict__ bitmap, uint32_t L) {
    const uint32_t n = list_is_a ? mdl->fwd_mismatch : mdl->fwd_mismatch2;
    if (n == 0u) return;
    const uint32_t B = mdl->B;
    const uint32_t C = (B + L - 1u) / L;
    const uint32_t words = (C + 31u) / 32u;
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) bitmap[i] = 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) atomicOr(&bitmap[list[i] >> 5], 1u << (list[i] & 31u));
    __syncthreads();
    __shared__ float sm_A[hml_amat<K>::LDS_FLOATS];
    hml_amat_fill<K>(sm_A, mdl, (int)threadIdx.x, (int)blockDim.x);
    if (threadIdx.x != 0) return;
    hml_emit_params<K> p;
    hml_emit_load<K>(p, mdl, 0);
    hml_fwd_ctx<K> cx;
    hml_fwd_ctx_load<K>(cx, mdl, sm_A);
    const unsigned long long epoch = mdl->epoch;
    const hml_key key = mdl->key;
    for (uint32_t wi = 0; wi < words; ++wi) {
        uint32_t bits = hml_ld_u32_coherent(bitmap + wi);
        while (bits) {
            const int bit = __ffs(bits) - 1;
            bits &= bits - 1u;
            float alpha[K];
            bool have_alpha = false;
            for (uint32_t f = wi * 32u + (uint32_t)bit; f < C && f > 0u; ++f) {
                bool same = true;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    if (!have_alpha) alpha[s] = hml_u2f(hml_ld_bits_coherent(exitv + (uint64_t)(f - 1u) * K + s));
                    same = same && (hml_f2u(alpha[s]) == hml_ld_bits_coherent(entry + (uint64_t)f * K + s));
                }
                if (same) break;   // consistent (possibly repaired already by an earlier chain)
                hml_tre_chunk_sequential<K>(p, cx, ia, starts, mdl, cand, fmap, entry, exitv, fb_count, eprobe, aprobe, f, 0u, true, alpha, epoch, key, L);
                have_alpha = true;   // alpha is now this chunk's end vector = the next chunk's true start
                atomicAdd(&mdl->forward_serial, 1ull);
            }
        }
    }
    mdl->fwd_serial_ran = 1u;
}

// ------------------------------------------------------------------------------------------
// the chain over the chunk maps, two levels (hml_k_backward_super / _chain / _entries with a chunk of L rows)
// ------------------------------------------------------------------------------------------
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_k_trellis_super(const unsigned long long* __restrict__ fmap, const hml_model* __restrict__ mdl,
                                                           unsigned long long* __restrict__ scmap, unsigned long long* __restrict__ super, uint32_t L) {
    const uint32_t NC = (mdl->B + L - 1u) / L;
    const uint32_t NS = (NC + 63u) / 64u;
    const int lane = threadIdx.x & 63;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t S = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; S < NS; S += nwaves) {
        const uint32_t c = S * 64u + (uint32_t)lane;
        unsigned long long map = c < NC ? fmap[c] : HML_MAP_IDENTITY;
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

// one workgroup: the state entering every super-chunk (entry2[S] = state of the first row after super-chunk S)
template <int K>
HML_KERNEL __launch_bounds__(1024) void hml_k_trellis_chain(const unsigned long long* __restrict__ super, const hml_model* __restrict__ mdl,
                                                            uint8_t* __restrict__ entry2, uint32_t L) {
    __shared__ unsigned long long P[1024];
    const uint32_t NCf = (mdl->B + L - 1u) / L;
    const uint32_t NC = (NCf + 63u) / 64u;
    const int tid = threadIdx.x;
    const uint32_t per = (NC + 1023u) / 1024u;
    const uint32_t a = (uint32_t)tid * per < NC ? (uint32_t)tid * per : NC;
    const uint32_t b = (a + per < NC) ? a + per : NC;
    unsigned long long prod = HML_MAP_IDENTITY;
    for (uint32_t c = b; c > a; --c) prod = hml_map_compose<K>(super[c - 1], prod);
    P[tid] = prod;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long o = (tid + d < 1024) ? P[tid + d] : HML_MAP_IDENTITY;
        __syncthreads();
        P[tid] = hml_map_compose<K>(P[tid], o);
        __syncthreads();
    }
    if (a >= b) return;
    const unsigned long long later = (tid + 1 < 1024) ? P[tid + 1] : HML_MAP_IDENTITY;
    unsigned x = (unsigned)(later & 15ull);   // (the last chunk's map is constant: the dummy 0 entering it is never used)
    for (uint32_t c = b; c > a; --c) {
        entry2[c - 1] = (uint8_t)x;
        x = (unsigned)(super[c - 1] >> (4 * x)) & 15u;
    }
}

// States of all rows: chunk f's entering state = scmap[f+1](entry2[S]) (or entry2[S] at the end of its super-chunk), then
// the chunk's candidate maps from its last row down, one chunk per lane.  A wavefront (= workgroup) takes 64 chunks and
// walks them in batches of 32 rows: the batch's maps come in through LDS with coalesced loads (lane = (chunk, row) again)
// and the states leave through LDS with coalesced stores.
template <int K>
HML_KERNEL __launch_bounds__(64) void hml_k_trellis_states(const unsigned long long* __restrict__ cand, const unsigned long long* __restrict__ scmap,
                                                           const uint8_t* __restrict__ entry2, const hml_model* __restrict__ mdl,
                                                           int16_t* __restrict__ q, uint32_t L) {
    typedef typename hml_tre_map<K>::stored map_t;
    constexpr int RB = 32, PITCH = RB + 1;
    __shared__ map_t sm_m[HML_TRE_NCH * PITCH];
    __shared__ int16_t sm_q[HML_TRE_NCH * (RB + 2)];
    const map_t* __restrict__ cm = reinterpret_cast<const map_t*>(cand);
    const uint32_t B = mdl->B;
    const uint32_t NC = (B + L - 1u) / L;
    const uint32_t n_groups = (NC + (uint32_t)HML_TRE_NCH - 1u) / (uint32_t)HML_TRE_NCH;
    const int lane = threadIdx.x;
    for (uint32_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const uint32_t f0 = grp * (uint32_t)HML_TRE_NCH;
        const uint32_t f = f0 + (uint32_t)lane;
        const bool active = f < NC;
        unsigned x = 0u;
        if (active) {
            const unsigned e2 = entry2[f >> 6];
            const bool last_of_super = (f & 63u) == 63u || f + 1u == NC;
            x = last_of_super ? e2 : (unsigned)((scmap[f + 1u] >> (4u * e2)) & 15ull);
        }
        const uint64_t first = (uint64_t)f * L;
        const uint64_t last = active ? ((first + L < B) ? first + L : (uint64_t)B) : first;
        // lane = (chunk, row): half a wavefront reads one chunk's 32 consecutive maps; all 32 loads of a lane in flight, and the
        // NEXT batch's loads are issued before the current batch is walked (the walk and the stores hide their latency)
        auto fetch = [&](int rel0, map_t (&v)[RB]) {
            // one scalar base per load and ONE 32-bit lane offset for all of them, clamped into the array instead of predicated (what
            // a row beyond the sweep loads is never used): round 5's form held a 64-bit address per load - 225 vector registers,
            // two wavefronts per SIMD; this one 129 and three (177 -> 168 us per launch at 10^8 blocks)
            if (rel0 < 0) return;                                    // (wavefront-uniform: the prefetch behind the chunks' first batch)
            const uint32_t off_lane = ((uint32_t)lane / (uint32_t)RB) * L + ((uint32_t)lane % (uint32_t)RB) + 1u;   // row t = b + 1
            const char* __restrict__ cmb = reinterpret_cast<const char*>(cm);
            if ((uint64_t)(f0 + (uint32_t)HML_TRE_NCH) * L <= (uint64_t)B) {   // every chunk of the group inside the sweep (all groups but the last)
                const uint32_t boff = off_lane * (uint32_t)sizeof(map_t);
#pragma unroll
                for (int k = 0; k < RB; ++k) {
                    const uint64_t u = (uint64_t)(f0 + (uint32_t)(k * (64 / RB))) * L + (uint32_t)rel0;   // wavefront-uniform
                    v[k] = *reinterpret_cast<const map_t*>(cmb + u * sizeof(map_t) + boff);
                }
            } else {
#pragma unroll
                for (int k = 0; k < RB; ++k) {
                    const uint64_t u = (uint64_t)(f0 + (uint32_t)(k * (64 / RB))) * L + (uint32_t)rel0;
                    const uint32_t rem = u < (uint64_t)B ? (uint32_t)((uint64_t)B - u) : 0u;            // (cand has B + 1 entries)
                    const uint64_t uc = u < (uint64_t)B ? u : (uint64_t)B;
                    v[k] = *reinterpret_cast<const map_t*>(cmb + uc * sizeof(map_t) + (off_lane < rem ? off_lane : rem) * (uint32_t)sizeof(map_t));
                }
            }
        };
        map_t v[RB];
        fetch((int)L - RB, v);
        for (int rel0 = (int)L - RB; rel0 >= 0; rel0 -= RB) {   // batches from the chunks' ends down (L is a multiple of 32)
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const int slot = k * 64 + lane;
                sm_m[(slot / RB) * PITCH + (slot % RB)] = v[k];
            }
            __syncthreads();
            fetch(rel0 - RB, v);
            if (active) {
                map_t mine[RB];
#pragma unroll
                for (int r = 0; r < RB; ++r) mine[r] = sm_m[lane * PITCH + r];
#pragma unroll
                for (int r = RB - 1; r >= 0; --r) {
                    const uint64_t b = first + (uint32_t)rel0 + (uint32_t)r;
                    if (b < last) x = (unsigned)(((unsigned long long)mine[r] >> (4u * x)) & 15ull);   // q_t = cand_t(q_{t+1})
                    sm_q[lane * (RB + 2) + r] = (int16_t)x;   // (rows beyond the chunk's end are not stored below)
                }
            }
            __syncthreads();
#pragma unroll 8
            for (int k = 0; k < RB; ++k) {
                const int slot = k * 64 + lane;
                const int c = slot / RB, r = slot % RB;
                const uint64_t b = (uint64_t)(f0 + (uint32_t)c) * L + (uint32_t)rel0 + (uint32_t)r;
                if (f0 + (uint32_t)c < NC && b < B) q[b] = sm_q[c * (RB + 2) + r];
            }
            __syncthreads();
        }
    }
}

#endif
