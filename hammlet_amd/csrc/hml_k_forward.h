// Emission terms and the forward trellis.
#ifndef HML_K_FORWARD_H
#define HML_K_FORWARD_H

#include "hml_k_blocks.h"
#include "hml_math.h"
#include "hml_state.h"

__device__ __forceinline__ void hml_raise(hml_model* mdl, uint32_t code, float value) {
    if (atomicCAS(&mdl->err_code, 0u, code) == 0u) mdl->err_value = value;
    atomicAdd(&mdl->err_count, 1ull);
}

__device__ __forceinline__ bool hml_isfinite(float x) { return (hml_f2u(x) & 0x7f800000u) != 0x7f800000u; }

// ------------------------------------------------------------------------------------------
// K6a emission - the per-block, per-state terms of StateSequence<ForwardBackward>::sample
// (reference src/StateSequence/ForwardBackward.hpp:67-84) with innerProduct / logNormalizer
// (src/EFD.hpp:23-38,83-93):
//   ip_s = (float)((2.0*mu_s*Sx - Sxx) / (2.0*var_s))                 [double inside]
//   E_s  = (0.0f + ip_s) - N*logN_s  [+ (N-1)*logA_s]                  [float]
//   e_s  = expf(E_s - max_s E_s)
//   g_s  = expf((N-1)*logA_s)       the factor the previous trellis row is rescaled by (:115-119)
// One thread per block; em/g are [B][K] floats.
// ------------------------------------------------------------------------------------------
template <int K>
struct hml_emit_params {
    float mu[K], var[K], logN[K], logA[K];
    bool self;
};

template <int K>
__device__ __forceinline__ void hml_emit_load(hml_emit_params<K>& p, const hml_model* mdl, int mixture) {
#pragma unroll
    for (int s = 0; s < K; ++s) { p.mu[s] = mdl->mu[s]; p.var[s] = mdl->var[s]; p.logN[s] = mdl->logN[s]; p.logA[s] = mdl->logA[s]; }
    p.self = mdl->self_trans != 0 && !mixture;
}

template <int K>
__device__ __forceinline__ void hml_emit_block(const hml_emit_params<K>& p, hml_model* mdl, uint32_t b, float sx, float sq,
                                               float N, float* __restrict__ em, float* __restrict__ gsc,
                                               float* __restrict__ eprobe, int mixture) {
    float E[K];
    float maxE = -3.40282346638528859812e+38f;   // numeric_limits<float>::lowest()
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const double ipd = (2.0 * (double)p.mu[s] * (double)sx - (double)sq) / (2.0 * (double)p.var[s]);
        const float ip = (float)ipd;
        if (!hml_isfinite(ip)) hml_raise(mdl, HML_DEVERR_IP_NOT_FINITE, ip);
        float e = (0.0f + ip) - N * p.logN[s];
        if (p.self) e += (N - 1.0f) * p.logA[s];
        E[s] = e;
        maxE = (e < maxE) ? maxE : e;   // std::max(E, maxE)
    }
#pragma unroll
    for (int s = 0; s < K; ++s) {
        if (eprobe) eprobe[(uint64_t)b * K + s] = E[s];
        em[(uint64_t)b * K + s] = hml_expf(E[s] - maxE);
        if (!mixture) gsc[(uint64_t)b * K + s] = p.self ? hml_expf((N - 1.0f) * p.logA[s]) : 1.0f;
    }
}

template <int K>
__global__ __launch_bounds__(256) void hml_k_emission(const float2* __restrict__ bstat,
                                                      const uint32_t* __restrict__ starts, hml_model* __restrict__ mdl,
                                                      float* __restrict__ em, float* __restrict__ gsc,
                                                      float* __restrict__ eprobe, int mixture) {
    const uint32_t B = mdl->B;
    hml_emit_params<K> p;
    hml_emit_load<K>(p, mdl, mixture);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += stride) {
        const float2 st = bstat[b];
        const float N = (float)(starts[b + 1] - starts[b]);
        hml_emit_block<K>(p, mdl, b, st.x, st.y, N, em, gsc, eprobe, mixture);
    }
}

// K5+K6a fused for the sweeps that rebuild the block structure: one thread per block gathers the block
// statistics from the integral array and emits the per-state terms - one dense launch instead of two.
template <int K>
__global__ __launch_bounds__(256) void hml_k_stats_emission(const float2* __restrict__ ia, const uint32_t* __restrict__ starts,
                                                            hml_model* __restrict__ mdl, float2* __restrict__ bstat,
                                                            float* __restrict__ em, float* __restrict__ gsc,
                                                            float* __restrict__ eprobe, int mixture) {
    const uint32_t B = mdl->B;
    hml_emit_params<K> p;
    hml_emit_load<K>(p, mdl, mixture);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += stride) {
        const uint32_t s = starts[b], e = starts[b + 1];
        float sx, sq;
        hml_block_stats_one(ia, s, e, sx, sq);
        bstat[b] = make_float2(sx, sq);
        hml_emit_block<K>(p, mdl, b, sx, sq, (float)(e - s), em, gsc, eprobe, mixture);
    }
}

// ------------------------------------------------------------------------------------------
// K6b forward - the forward filter (reference src/StateSequence/ForwardBackward.hpp:86-123):
//   f_j = e_t(j) * sum_i alpha_{t-1}(i) A(i,j)   (i in order, float)
//   Z   = sum_j f_j                               (j in order, float)
//   alpha_t = f / Z,  or uniform if Z == 0
// and the stored row r_t = alpha_t * g_t for t < B (the reference rescales row t after step t+1
// has consumed it), r_B = alpha_B.
//
// The recursion is sequential in t.  To run it in parallel WITHOUT changing a single rounding, the
// blocks are cut into chunks of L; chunk c first runs the recursion over the W blocks before it
// from an arbitrary start (a warm-up whose results are discarded) and then over its own blocks.
// A hidden-Markov filter forgets its start exponentially fast, so the warm-up usually ends in
// exactly the bits the sequential recursion would have produced - "usually" is then turned into
// "always": a verification pass compares, bit for bit, the vector each chunk started from with the
// vector its predecessor really ended in, and recomputes stale chunks from the true vector; a final
// serial pass finishes whatever is still inconsistent.  When every comparison passes, induction
// from chunk 0 (which starts from pi itself) proves that the stored rows ARE the sequential ones.
//
// Geometry: 16 lanes per chunk, lane j owns state j; 4 chunks per wavefront.
// ------------------------------------------------------------------------------------------
template <int K>
struct hml_fwd_ctx {
    float Acol[K];     // A(i, j) for this lane's j
    float invK;
    int j;
    bool self;
    uint32_t B;
};

// one step of the recursion for block b (row t = b+1); returns the new alpha_j
template <int K>
__device__ __forceinline__ float hml_fwd_step(const hml_fwd_ctx<K>& c, float alpha, float e, bool& fellback) {
    float tt = 0.0f;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const float p = __shfl(alpha, i, HML_FWD_GROUP);
        tt += p * c.Acol[i];
    }
    const float f = e * tt;
    float Z = 0.0f;
#pragma unroll
    for (int i = 0; i < K; ++i) Z += __shfl(f, i, HML_FWD_GROUP);
    fellback = !(Z != 0.0f);
    return (Z != 0.0f) ? f / Z : c.invK;
}

// MODE 0: speculative main pass.  MODE 1: verify against exit_in and recompute stale chunks.
// MODE 2: verify only - raise mdl->fwd_mismatch if any chunk is still inconsistent (the serial pass then runs).
template <int K, int MODE>
__global__ __launch_bounds__(256) void hml_k_forward(const float* __restrict__ em, const float* __restrict__ gsc,
                                                     hml_model* __restrict__ mdl, float* __restrict__ rows,
                                                     float* __restrict__ aprobe, float* __restrict__ entry,
                                                     const float* __restrict__ exit_in, float* __restrict__ exit_out,
                                                     uint32_t* __restrict__ fb_count, int L, int W) {
    const uint32_t B = mdl->B;
    const uint32_t C = (B + (uint32_t)L - 1u) / (uint32_t)L;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ngroups = (gridDim.x * blockDim.x) / HML_FWD_GROUP;
    const int j = (int)(gid % HML_FWD_GROUP);
    const int lane = threadIdx.x & 63;
    const int grp_in_wave = lane / HML_FWD_GROUP;
    const bool act = (j < K);
    hml_fwd_ctx<K> cx;
    cx.j = j; cx.B = B; cx.self = mdl->self_trans != 0;
    cx.invK = (float)(1.0 / (double)(float)K);
#pragma unroll
    for (int i = 0; i < K; ++i) cx.Acol[i] = act ? mdl->A[i * K + j] : 0.0f;
    if (MODE == 0 && gid < (uint32_t)K) {
        rows[gid] = mdl->pi[gid];
        if (aprobe) aprobe[gid] = mdl->pi[gid];
    }

    // grid-stride over chunks: correctness never depends on the launch size (B is only known on the device)
    for (uint32_t c = gid / HML_FWD_GROUP; c < C; c += ngroups) {
        const uint32_t first = c * (uint32_t)L;
        const uint32_t last = (first + (uint32_t)L < B) ? first + (uint32_t)L : B;   // one past
        const uint32_t ws = (first >= (uint32_t)W) ? first - (uint32_t)W : 0u;
        const bool exact = (ws == 0u);
        float alpha = 0.0f;
        bool run = true;
        if (MODE == 0) {
            alpha = act ? (exact ? mdl->pi[j] : cx.invK) : 0.0f;
            // warm-up over [ws, first)
            for (uint32_t b = ws; b < first; ++b) {
                const float e = act ? em[(uint64_t)b * K + j] : 0.0f;
                bool fb;
                alpha = hml_fwd_step<K>(cx, alpha, e, fb);
            }
            if (act) entry[(uint64_t)c * K + j] = alpha;
        } else {
            // verification: was the vector this chunk started from the one its predecessor really ended in?
            bool same = true;
            float truth = 0.0f;
            if (act && !exact) {
                truth = exit_in[(uint64_t)(c - 1) * K + j];
                same = hml_f2u(truth) == hml_f2u(entry[(uint64_t)c * K + j]);
            }
            const unsigned long long bal = __ballot(same);
            const bool all_same = ((bal >> (grp_in_wave * HML_FWD_GROUP)) & 0xffffull) == 0xffffull;
            if (MODE == 2) {
                if (!exact && !all_same && j == 0) mdl->fwd_mismatch = 1u;
                continue;
            }
            if (exact || all_same) {
                if (act) exit_out[(uint64_t)c * K + j] = exit_in[(uint64_t)c * K + j];
                run = false;
            } else {
                alpha = truth;
                if (act) entry[(uint64_t)c * K + j] = alpha;
                if (j == 0) atomicAdd(&mdl->forward_refits, 1ull);
            }
        }
        if (!run) continue;
        // the chunk proper over [first, last)
        uint32_t nfb = 0;
        for (uint32_t b = first; b < last; ++b) {
            const float e = act ? em[(uint64_t)b * K + j] : 0.0f;
            const float g = (act && cx.self) ? gsc[(uint64_t)b * K + j] : 1.0f;
            bool fb;
            alpha = hml_fwd_step<K>(cx, alpha, e, fb);
            if (fb) nfb++;
            if (act) {
                const uint32_t t = b + 1u;
                const float stored = (cx.self && t < B) ? alpha * g : alpha;
                rows[(uint64_t)t * K + j] = stored;
                if (aprobe) aprobe[(uint64_t)t * K + j] = alpha;
            }
        }
        if (MODE == 1) {
            // a recomputed chunk that ends in different bits leaves its successor inconsistent: only then
            // does the serial pass have work
            const bool unchanged = !act || hml_f2u(alpha) == hml_f2u(exit_in[(uint64_t)c * K + j]);
            const unsigned long long bal2 = __ballot(unchanged);
            if (((bal2 >> (grp_in_wave * HML_FWD_GROUP)) & 0xffffull) != 0xffffull && j == 0) mdl->fwd_mismatch = 1u;
        }
        if (act) exit_out[(uint64_t)c * K + j] = alpha;
        if (j == 0) {
            // "[WARNING] Uniform sampling of forward variables!" events: keep the global tally consistent
            // when a chunk is recomputed (two's-complement delta on the unsigned counter)
            const uint32_t old = (MODE == 0) ? 0u : fb_count[c];
            fb_count[c] = nfb;
            if (nfb != old) atomicAdd(&mdl->uniform_fallbacks, (unsigned long long)(long long)((int)nfb - (int)old));
        }
    }
}

// Final serial pass: find the first chunk whose start vector is not its predecessor's end vector;
// if there is none (the normal case) just add up the fallback counters, otherwise walk the chain
// from there and recompute every inconsistent chunk in order.  One workgroup; the walk is done by
// its first 16 lanes.
template <int K>
__global__ __launch_bounds__(256) void hml_k_forward_serial(const float* __restrict__ em, const float* __restrict__ gsc,
                                                            hml_model* __restrict__ mdl, float* __restrict__ rows,
                                                            float* __restrict__ aprobe, float* __restrict__ entry,
                                                            float* __restrict__ exitv, uint32_t* __restrict__ fb_count,
                                                            int L, int W) {
    __shared__ uint32_t first_bad;
    if (mdl->fwd_mismatch == 0u) return;   // the verification pass found every chunk consistent
    const uint32_t B = mdl->B;
    const uint32_t C = (B + (uint32_t)L - 1u) / (uint32_t)L;
    const int tid = threadIdx.x;
    if (tid == 0) first_bad = 0xffffffffu;
    __syncthreads();
    for (uint32_t c = 1u + (uint32_t)tid; c < C; c += 256u) {
        const uint32_t first = c * (uint32_t)L;
        if (first <= (uint32_t)W) continue;   // started from pi: exact by construction
        bool same = true;
#pragma unroll
        for (int s = 0; s < K; ++s)
            same = same && (hml_f2u(entry[(uint64_t)c * K + s]) == hml_f2u(exitv[(uint64_t)(c - 1) * K + s]));
        if (!same) atomicMin(&first_bad, c);
    }
    __syncthreads();
    const uint32_t fbad = first_bad;
    if (fbad != 0xffffffffu && tid < 64) {
        // serial repair by the first group of 16 lanes (the rest of the wavefront idles through the shuffles)
        const int j = tid % HML_FWD_GROUP;
        const bool act = (tid < HML_FWD_GROUP) && (j < K);
        hml_fwd_ctx<K> cx;
        cx.j = j; cx.B = B; cx.self = mdl->self_trans != 0;
        cx.invK = (float)(1.0 / (double)(float)K);
#pragma unroll
        for (int i = 0; i < K; ++i) cx.Acol[i] = act ? mdl->A[i * K + j] : 0.0f;
        for (uint32_t c = fbad; c < C; ++c) {
            const uint32_t first = c * (uint32_t)L;
            const uint32_t last = (first + (uint32_t)L < B) ? first + (uint32_t)L : B;
            float truth = 0.0f;
            bool same = true;
            if (act) {
                truth = exitv[(uint64_t)(c - 1) * K + j];
                same = hml_f2u(truth) == hml_f2u(entry[(uint64_t)c * K + j]);
            }
            const unsigned long long bal = __ballot(same);
            if ((bal & 0xffffull) == 0xffffull) continue;
            float alpha = truth;
            if (act) entry[(uint64_t)c * K + j] = alpha;
            uint32_t nfb = 0;
            for (uint32_t b = first; b < last; ++b) {
                const float e = act ? em[(uint64_t)b * K + j] : 0.0f;
                const float g = (act && cx.self) ? gsc[(uint64_t)b * K + j] : 1.0f;
                bool fb;
                alpha = hml_fwd_step<K>(cx, alpha, e, fb);
                if (fb) nfb++;
                if (act) {
                    const uint32_t t = b + 1u;
                    rows[(uint64_t)t * K + j] = (cx.self && t < B) ? alpha * g : alpha;
                    if (aprobe) aprobe[(uint64_t)t * K + j] = alpha;
                }
            }
            if (act) exitv[(uint64_t)c * K + j] = alpha;
            if (tid == 0) {
                const uint32_t old = fb_count[c];
                fb_count[c] = nfb;
                if (nfb != old) atomicAdd(&mdl->uniform_fallbacks, (unsigned long long)(long long)((int)nfb - (int)old));
                atomicAdd(&mdl->forward_serial, 1ull);
            }
            __threadfence_block();
        }
    }
    __syncthreads();
    if (tid == 0) mdl->fwd_mismatch = 0u;
}

#endif
