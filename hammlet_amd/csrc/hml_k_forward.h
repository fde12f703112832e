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
    double rvar[K];   // 1 / (2 var), from the model (computed once per parameter draw)
    bool self;
};

// (float)((2.0 mu Sx - Sxx) / (2.0 var)) - the reference's inner product (EFD.hpp:23-33), double inside - with the
// quotient taken as a product with the double reciprocal: the product is within 2 ulp of the correctly rounded
// quotient, so both round to the same float unless the product lies within 4 ulp of the midpoint of two floats (or is
// tiny / not finite); only then is the division carried out.  The reciprocal 1 / (2 var) comes from the model (the
// parameter kernel computes it once per draw), so no lane pays for a division in the common case.
__device__ __forceinline__ float hml_inner_product(float mu, float var, double rvar, float sx, float sq) {
    const double num = 2.0 * (double)mu * (double)sx - (double)sq;
    double ipd = num * rvar;
    const uint64_t bits = hml_d2u(ipd);
    const uint32_t low = (uint32_t)bits & 0x1fffffffu;
    const uint32_t ex = (uint32_t)(bits >> 52) & 0x7ffu;
    const bool close = (low - 0x0ffffffcu) <= 8u;                  // within 4 ulp of a float midpoint
    if (close || ex < 923u || ex > 1150u) {                        // |ip| < 2^-100, > 2^127, inf/NaN (0 is decided by the product)
        if (ipd != 0.0 || close) ipd = num / (2.0 * (double)var);
    }
    return (float)ipd;
}

template <int K>
__device__ __forceinline__ void hml_emit_load(hml_emit_params<K>& p, const hml_model* mdl, int mixture) {
#pragma unroll
    for (int s = 0; s < K; ++s) { p.mu[s] = mdl->mu[s]; p.var[s] = mdl->var[s]; p.logN[s] = mdl->logN[s]; p.logA[s] = mdl->logA[s]; p.rvar[s] = mdl->rvar2[s]; }
    p.self = mdl->self_trans != 0 && !mixture;
}

// the terms of one block in registers: E_s, e_s = expf(E_s - max E), g_s = expf((N-1) logA_s) (1 without self-transitions)
template <int K, bool LEAN = false>
__device__ __forceinline__ void hml_emit_compute(const hml_emit_params<K>& p, hml_model* mdl, float sx, float sq, float N, int mixture,
                                                 float (&E)[K], float (&ev)[K], float (&gv)[K], bool want_g = true,
                                                 const uint64_t* exp_tab = nullptr) {
    const uint64_t* const tab = exp_tab ? exp_tab : HML_EXP2F_TAB;
    float maxE = -3.40282346638528859812e+38f;   // numeric_limits<float>::lowest()
#pragma unroll
    for (int s = 0; s < K; ++s) {
        // (float)((2.0 mu Sx - Sxx) / (2.0 var)) through the reciprocal; divides only where the product could round differently
        const float ip = hml_inner_product(p.mu[s], p.var[s], p.rvar[s], sx, sq);
        if (!hml_isfinite(ip)) hml_raise(mdl, HML_DEVERR_IP_NOT_FINITE, ip);
        float e = (0.0f + ip) - N * p.logN[s];
        if (p.self) e += (N - 1.0f) * p.logA[s];
        E[s] = e;
        maxE = (e < maxE) ? maxE : e;   // std::max(E, maxE)
    }
#pragma unroll
    for (int s = 0; s < K; ++s) {
        ev[s] = LEAN ? hml_expf_lean(E[s] - maxE, tab) : hml_expf_tab(E[s] - maxE, tab);
        gv[s] = (want_g && !mixture && p.self) ? hml_expf_tab((N - 1.0f) * p.logA[s], tab) : 1.0f;
    }
}

template <int K>
__device__ __forceinline__ void hml_emit_store(uint32_t b, const float (&E)[K], const float (&ev)[K], const float (&gv)[K],
                                               float* __restrict__ em, float* __restrict__ gsc, float* __restrict__ eprobe,
                                               int mixture, const hml_layout lay) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
        if (eprobe) eprobe[(uint64_t)b * K + s] = E[s];
        em[hml_bk(lay, b, K, s)] = ev[s];
        if (!mixture && gsc) gsc[hml_bk(lay, b, K, s)] = gv[s];
    }
}

template <int K, bool LEAN = false>
__device__ __forceinline__ void hml_emit_block(const hml_emit_params<K>& p, hml_model* mdl, uint32_t b, float sx, float sq,
                                               float N, float* __restrict__ em, float* __restrict__ gsc,
                                               float* __restrict__ eprobe, int mixture, const hml_layout lay,
                                               const uint64_t* exp_tab = nullptr) {
    // gsc == nullptr: the rescale factors are not stored - the backward maps compute them where they apply them
    // (hml_bwd_row_load; strongly compressed univariate sweeps, where the plane was a third of this kernel's stores)
    float E[K], ev[K], gv[K];
    hml_emit_compute<K, LEAN>(p, mdl, sx, sq, N, mixture, E, ev, gv, gsc != nullptr, exp_tab);
    hml_emit_store<K>(b, E, ev, gv, em, gsc, eprobe, mixture, lay);
}

// The same terms for MANY STATES with few registers (the fused block kernel beyond 6 states: its register budget is
// fixed by the number of workgroups it needs resident): the parameters come from LDS (hml_emit_lds, filled once per
// workgroup), the states are walked twice - first for max E, then again for E_s and e_s = expf(E_s - max E), which are
// stored at once - so no per-state array lives in registers.  Every E_s is computed by the same operations as in
// hml_emit_compute (twice), hence the same bits.
template <int K>
struct hml_emit_lds {
    float mu[K], var[K], logN[K], logA[K];
    double rvar[K];
};
template <int K>
__device__ __forceinline__ void hml_emit_lds_fill(hml_emit_lds<K>& l, const hml_model* mdl, int tid) {
    if (tid < K) { l.mu[tid] = mdl->mu[tid]; l.var[tid] = mdl->var[tid]; l.logN[tid] = mdl->logN[tid]; l.logA[tid] = mdl->logA[tid]; l.rvar[tid] = mdl->rvar2[tid]; }
}
template <int K>
__device__ __forceinline__ float hml_emit_E(const hml_emit_lds<K>& l, hml_model* mdl, int s, bool self, float sx, float sq, float N) {
    const float ip = hml_inner_product(l.mu[s], l.var[s], l.rvar[s], sx, sq);
    if (!hml_isfinite(ip)) hml_raise(mdl, HML_DEVERR_IP_NOT_FINITE, ip);
    float e = (0.0f + ip) - N * l.logN[s];
    if (self) e += (N - 1.0f) * l.logA[s];
    return e;
}
template <int K>
__device__ __forceinline__ void hml_emit_block_looped(const hml_emit_lds<K>& l, hml_model* mdl, bool self_trans, uint32_t b, float sx, float sq,
                                                      float N, float* __restrict__ em, float* __restrict__ gsc,
                                                      float* __restrict__ eprobe, int mixture, const hml_layout lay,
                                                      const uint64_t* exp_tab) {
    const bool self = self_trans && !mixture;
    float maxE = -3.40282346638528859812e+38f;   // numeric_limits<float>::lowest()
#pragma unroll 1
    for (int s = 0; s < K; ++s) {
        const float e = hml_emit_E<K>(l, mdl, s, self, sx, sq, N);
        maxE = (e < maxE) ? maxE : e;
    }
#pragma unroll 1
    for (int s = 0; s < K; ++s) {
        const float e = hml_emit_E<K>(l, mdl, s, self, sx, sq, N);
        if (eprobe) eprobe[(uint64_t)b * K + s] = e;
        em[hml_bk(lay, b, K, s)] = hml_expf_lean(e - maxE, exp_tab);
        if (!mixture && gsc) gsc[hml_bk(lay, b, K, s)] = self ? hml_expf_lean((N - 1.0f) * l.logA[s], exp_tab) : 1.0f;
    }
}

template <int K>
HML_KERNEL __launch_bounds__(256) void hml_k_emission(const float2* __restrict__ bstat,
                                                      const uint32_t* __restrict__ starts, hml_model* __restrict__ mdl,
                                                      float* __restrict__ em, float* __restrict__ gsc,
                                                      float* __restrict__ eprobe, int mixture, const hml_layout lay) {
    const uint32_t B = mdl->B;
    hml_emit_params<K> p;
    hml_emit_load<K>(p, mdl, mixture);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += stride) {
        const float2 st = bstat[b];
        const float N = (float)(starts[b + 1] - starts[b]);
        hml_emit_block<K>(p, mdl, b, st.x, st.y, N, em, gsc, eprobe, mixture, lay);
    }
}

// K5+K6a fused for the sweeps that rebuild the block structure: one thread per block gathers the block
// statistics from the integral array and emits the per-state terms - one dense launch instead of two.
template <int K>
__device__ __forceinline__ void hml_b_stats_emission(const float2* __restrict__ ia, const uint32_t* __restrict__ starts,
                                                            hml_model* __restrict__ mdl, float2* __restrict__ bstat,
                                                            float* __restrict__ em, float* __restrict__ gsc,
                                                            float* __restrict__ eprobe, int mixture, const hml_layout lay) {
    const uint32_t B = mdl->B;
    hml_emit_params<K> p;
    hml_emit_load<K>(p, mdl, mixture);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += stride) {
        const uint32_t s = starts[b], e = starts[b + 1];
        float sx, sq;
        hml_block_stats_one(ia, s, e, sx, sq);
        bstat[b] = make_float2(sx, sq);
        hml_emit_block<K>(p, mdl, b, sx, sq, (float)(e - s), em, gsc, eprobe, mixture, lay);
    }
}
// the kernel: hml_b_stats_emission over one chain (hml_k_many.h runs it over several chains in one launch)
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_k_stats_emission(const float2* __restrict__ ia, const uint32_t* __restrict__ starts,
                                                            hml_model* __restrict__ mdl, float2* __restrict__ bstat,
                                                            float* __restrict__ em, float* __restrict__ gsc,
                                                            float* __restrict__ eprobe, int mixture, const hml_layout lay) {
    hml_b_stats_emission<K>(ia, starts, mdl, bstat, em, gsc, eprobe, mixture, lay);
}


// ------------------------------------------------------------------------------------------
// Multivariate / shared-parameter form ("-s C P D", reference src/Mapping.hpp, src/EFD.hpp:83-93, src/Theta.hpp:148-158):
// the statistics of the D data dimensions lie in planes `stat_stride` apart, state s uses parameter map[s][d] for
// dimension d, and
//   E_s = sum_d fl32( (2.0 mu_p Sx_d - Sxx_d) / (2.0 var_p) )  (float running sum from 0, dimension order, p = map[s][d])
//         - N * logNs[s]  [+ (N-1) logA_s]
// One thread per block.  STATS: block statistics from the integral arrays first, else from the bstat planes.
template <int K, bool STATS>
HML_KERNEL __launch_bounds__(256) void hml_k_emission_mv(const float2* __restrict__ ia, const uint32_t* __restrict__ starts,
                                                         hml_model* __restrict__ mdl, float2* __restrict__ bstat,
                                                         float* __restrict__ em, float* __restrict__ gsc,
                                                         float* __restrict__ eprobe, int mixture, const hml_layout lay) {
    // the model's parameters once per workgroup (round 3: read per block through the writable pointer every value was a
    // cache round trip of its own, and the quotient a double division per state and dimension - hml_inner_product takes
    // it through the reciprocal the parameter kernel keeps, with the division where the product could round differently:
    // the same float)
    __shared__ float s_mu[HML_MAX_K], s_var[HML_MAX_K], s_logNs[K], s_logA[K];
    __shared__ double s_rvar[HML_MAX_K];
    __shared__ uint8_t s_map[K][HML_MAX_D];
    __shared__ uint64_t s_tab[32];
    const uint32_t B = mdl->B;
    const int D = mdl->D;
    const uint64_t T = mdl->T;
    const bool self = mdl->self_trans != 0 && !mixture;
    if (threadIdx.x < (unsigned)mdl->P) { s_mu[threadIdx.x] = mdl->mu[threadIdx.x]; s_var[threadIdx.x] = mdl->var[threadIdx.x]; s_rvar[threadIdx.x] = mdl->rvar2[threadIdx.x]; }
    if (threadIdx.x < (unsigned)K) {
        s_logNs[threadIdx.x] = mdl->logNs[threadIdx.x]; s_logA[threadIdx.x] = mdl->logA[threadIdx.x];
        for (int d = 0; d < HML_MAX_D; ++d) s_map[threadIdx.x][d] = mdl->map[threadIdx.x][d];
    }
    if (threadIdx.x >= 64 && threadIdx.x < 96) s_tab[threadIdx.x - 64] = HML_EXP2F_TAB[threadIdx.x - 64];
    __syncthreads();
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += stride) {
        const uint32_t st = starts[b], en = starts[b + 1];
        float sx[HML_MAX_D], sq[HML_MAX_D];
        for (int d = 0; d < D; ++d) {
            if (STATS) {
                hml_block_stats_one(ia + (uint64_t)d * (T + 1u), st, en, sx[d], sq[d]);
                bstat[(uint64_t)d * T + b] = make_float2(sx[d], sq[d]);
            } else {
                const float2 v = bstat[(uint64_t)d * T + b];
                sx[d] = v.x; sq[d] = v.y;
            }
        }
        const float N = (float)(en - st);
        float E[K];
        float maxE = -3.40282346638528859812e+38f;
#pragma unroll
        for (int s = 0; s < K; ++s) {
            float r = 0.0f;
            for (int d = 0; d < D; ++d) {
                const int pp = s_map[s][d];
                const float ip = hml_inner_product(s_mu[pp], s_var[pp], s_rvar[pp], sx[d], sq[d]);
                if (!hml_isfinite(ip)) hml_raise(mdl, HML_DEVERR_IP_NOT_FINITE, ip);
                r += ip;
            }
            float e = r - N * s_logNs[s];
            if (self) e += (N - 1.0f) * s_logA[s];
            E[s] = e;
            maxE = (e < maxE) ? maxE : e;
        }
#pragma unroll
        for (int s = 0; s < K; ++s) {
            if (eprobe) eprobe[(uint64_t)b * K + s] = E[s];
            em[hml_bk(lay, b, K, s)] = hml_expf_tab(E[s] - maxE, s_tab);
            if (!mixture && gsc) gsc[hml_bk(lay, b, K, s)] = self ? hml_expf_tab((N - 1.0f) * s_logA[s], s_tab) : 1.0f;   // (no plane: the backward maps rescale, hml_bwd_row_load)
        }
    }
}

// ------------------------------------------------------------------------------------------
// Tiled form of the two kernels above for weakly compressed sweeps (millions of blocks, forward chunks of 16 or more
// blocks).  Same values bit for bit; three differences in how they are produced:
//  * stores: the chunk-transposed layout puts consecutive blocks of one chunk into different planes, so a block-per-lane
//    store scatters 4-byte words.  A workgroup takes a tile of whole chunks, keeps the tile's terms in LDS in the
//    layout's own order ([state][row in chunk][chunk], pitch + 1 against bank conflicts) and writes every
//    (row, state) plane segment as one run of consecutive floats.
//  * the quotient of the inner product is a product with the double reciprocal (hml_inner_product);
//  * g = expf((N-1) logA_s) comes from a per-workgroup table for N <= 64 (the same function on the same argument).
template <int K>
struct hml_emit_tile {
    static constexpr int BLOCKS = (K <= 8) ? 512 : 256;   // blocks per tile: whole chunks for L <= 64, LDS < 64 KB
    static constexpr int MAXL = 64;
    static constexpr int GTAB = 64;
};

template <int K>
__device__ __forceinline__ void hml_emit_values_fast(const hml_emit_params<K>& p, const double (&rvar)[K], const float* __restrict__ gtab,
                                                     hml_model* mdl, uint32_t b, float sx, float sq, uint32_t n,
                                                     float* __restrict__ eprobe, int mixture, float (&ev)[K], float (&gv)[K]) {
    const float N = (float)n;
    float E[K];
    float maxE = -3.40282346638528859812e+38f;
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const float ip = hml_inner_product(p.mu[s], p.var[s], rvar[s], sx, sq);
        if (!hml_isfinite(ip)) hml_raise(mdl, HML_DEVERR_IP_NOT_FINITE, ip);
        float e = (0.0f + ip) - N * p.logN[s];
        if (p.self) e += (N - 1.0f) * p.logA[s];
        E[s] = e;
        maxE = (e < maxE) ? maxE : e;
    }
#pragma unroll
    for (int s = 0; s < K; ++s) {
        if (eprobe) eprobe[(uint64_t)b * K + s] = E[s];
        ev[s] = hml_expf(E[s] - maxE);
        gv[s] = 1.0f;
        if (!mixture && p.self) gv[s] = (n <= (uint32_t)hml_emit_tile<K>::GTAB) ? gtab[(n - 1u) * K + s] : hml_expf((N - 1.0f) * p.logA[s]);
    }
}

// STATS = true: block statistics from the integral array first (hml_k_stats_emission); false: from bstat (hml_k_emission)
template <int K, bool STATS>
HML_KERNEL __launch_bounds__(256) void hml_k_emission_tiled(const float2* __restrict__ ia, const uint32_t* __restrict__ starts,
                                                            hml_model* __restrict__ mdl, float2* __restrict__ bstat,
                                                            float* __restrict__ em, float* __restrict__ gsc,
                                                            float* __restrict__ eprobe, int mixture, const hml_layout lay) {
    constexpr int TB = hml_emit_tile<K>::BLOCKS;
    constexpr int GT = hml_emit_tile<K>::GTAB;
    __shared__ float sm_e[K * TB + K * 64 + 64];     // [s][r][cl] with pitch cpt + 1 (cpt = chunks per tile <= TB)
    __shared__ float sm_g[K * TB + K * 64 + 64];
    __shared__ float gtab[GT * K];
    const uint32_t B = mdl->B;
    hml_emit_params<K> p;
    hml_emit_load<K>(p, mdl, mixture);
    double rvar[K];
#pragma unroll
    for (int s = 0; s < K; ++s) rvar[s] = p.rvar[s];
    for (int i = threadIdx.x; i < GT * K; i += 256) {
        const int n1 = i / K, s = i % K;   // n - 1
        gtab[i] = hml_expf((float)n1 * mdl->logA[s]);
    }
    const uint32_t Lr = 1u << lay.lshift;
    if (Lr > (uint32_t)hml_emit_tile<K>::MAXL) return;   // (the host launches the plain kernels for longer chunks)
    const uint32_t cpt = (uint32_t)TB >> lay.lshift;       // chunks per tile
    const uint32_t pitch = cpt + 1u;
    const uint32_t n_tiles = (B + TB - 1u) / TB;
    __syncthreads();
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t b0 = tile * TB;
#pragma unroll
        for (int j = 0; j < TB / 256; ++j) {
            const uint32_t bl = (uint32_t)j * 256u + threadIdx.x;
            const uint32_t b = b0 + bl;
            if (b < B) {
                const uint32_t st = starts[b], en = starts[b + 1];
                float sx, sq;
                if (STATS) { hml_block_stats_one(ia, st, en, sx, sq); bstat[b] = make_float2(sx, sq); }
                else { const float2 v = bstat[b]; sx = v.x; sq = v.y; }
                float ev[K], gv[K];
                hml_emit_values_fast<K>(p, rvar, gtab, mdl, b, sx, sq, en - st, eprobe, mixture, ev, gv);
                const uint32_t r = bl & (Lr - 1u), cl = bl >> lay.lshift;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    sm_e[((uint32_t)s * Lr + r) * pitch + cl] = ev[s];
                    sm_g[((uint32_t)s * Lr + r) * pitch + cl] = gv[s];
                }
            }
        }
        __syncthreads();
        // write-out: for every (row, state) the tile's chunks are consecutive floats in the plane
        const uint32_t c0 = b0 >> lay.lshift;
        const uint32_t nrs = Lr * (uint32_t)K;
        for (uint32_t e = threadIdx.x; e < nrs * cpt; e += 256u) {
            const uint32_t cl = e % cpt, rs = e / cpt;
            const uint32_t r = rs / (uint32_t)K, s = rs % (uint32_t)K;
            const uint32_t b = b0 + (cl << lay.lshift) + r;
            if (b < B) {
                const uint64_t g = ((uint64_t)r * (uint32_t)K + s) * lay.cstride + c0 + cl;
                em[g] = sm_e[(s * Lr + r) * pitch + cl];
                if (!mixture) gsc[g] = sm_g[(s * Lr + r) * pitch + cl];
            }
        }
        __syncthreads();
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
// ------------------------------------------------------------------------------------------
// Geometry: ONE LANE per chunk - the lane carries the whole K-vector in registers, A comes from scalar
// registers (it is the same for every lane), and nothing crosses lanes.  (A 16-lanes-per-chunk layout
// with one state per lane was measured first: its 2K ds_bpermute round trips per step made a step cost
// ~530 cycles and used 9x more wavefront-instructions per block.  The same layout with DPP row broadcasts instead
// (row_newbcast folded into the multiplies and adds: 45 instructions a step, no LDS) was measured in round 2: 26.9 us
// against 12.5 us for the 44 000 chunks of config 3 at W = 24 - with 4 chunks per wavefront instead of 64 the pass
// issues 4x the wavefront-instructions, and 11 000 wavefronts are throughput-bound where 700 are latency-bound.)
// The transition matrix as the kernels see it.  Up to 7 states its K * K floats live in (scalar) registers; beyond that
// they do not fit (at K = 10 a third of the forward kernel's vector instructions were v_readlane / v_writelane moves of
// spilled scalar registers, and the backward maps took 28 us against 5 us at K = 5): the workgroup keeps A in LDS and
// every use is a broadcast read.  hml_amat_fill must be called by all threads of the workgroup (it ends in a barrier).
#define HML_A_REGISTERS_MAX_K 7
template <int K, bool REG = (K <= HML_A_REGISTERS_MAX_K)>
struct hml_amat;
template <int K>
struct hml_amat<K, true> {
    static constexpr int LDS_FLOATS = 1;
    float v[K * K];
    __device__ __forceinline__ void attach(const hml_model* mdl, const float*) {
#pragma unroll
        for (int i = 0; i < K * K; ++i) v[i] = mdl->A[i];
    }
    __device__ __forceinline__ float operator[](int i) const { return v[i]; }
};
template <int K>
struct hml_amat<K, false> {
    static constexpr int LDS_FLOATS = K * K;
    const float* p;
    __device__ __forceinline__ void attach(const hml_model*, const float* lds) { p = lds; }
    __device__ __forceinline__ float operator[](int i) const { return p[i]; }
};
template <int K>
__device__ __forceinline__ void hml_amat_fill(float* lds, const hml_model* mdl, int tid, int nthreads) {
    if (K > HML_A_REGISTERS_MAX_K) {
        for (int i = tid; i < K * K; i += nthreads) lds[i] = mdl->A[i];
        __syncthreads();
    }
}

template <int K>
struct hml_fwd_ctx {
    hml_amat<K> A;
    float invK;
    bool self;
    uint32_t B;
};

// lds_A: the workgroup's copy of A behind hml_amat_fill (unused up to HML_A_REGISTERS_MAX_K states)
template <int K>
__device__ __forceinline__ void hml_fwd_ctx_load(hml_fwd_ctx<K>& cx, const hml_model* mdl, const float* lds_A) {
    cx.A.attach(mdl, lds_A);
    cx.invK = (float)(1.0 / (double)(float)K);
    cx.self = mdl->self_trans != 0;
    cx.B = mdl->B;
}

// f / Z, correctly rounded to float like the IEEE division it stands for, from a double reciprocal r of Z (relative error
// below 2^-27 is enough): q1 = f r, then one correction step q2 = q1 + (f - q1 Z) r with the residual from a fused
// multiply-add, which leaves q2 within 2^-53 q of the quotient q - and EQUAL to q whenever q is a double (r's error enters
// squared).  (float)q2 is the division's result: if q is a double nothing was rounded before the conversion; otherwise q is
// not a float midpoint, and no quotient of two floats lies closer to one than 2^-49 q (numerator minus midpoint times
// denominator is a non-zero multiple of the unit both are multiples of; holds for the sub-normal grid as well), so q2 is on
// q's side of it.  Five divisions of the filter step were 55 of its 101 instructions; tools/div_check.hip and
// test_quotient_by_reciprocal compare with the division itself, exact ties on the sub-normal grid included.
__device__ __forceinline__ double hml_tr2_reciprocal(double Zd) {
    double r = __builtin_amdgcn_rcp(Zd);
    const double e = __builtin_fma(-Zd, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ float hml_tr2_quotient(float f, double Zd, double r) {
    const double fd = (double)f;
    const double q1 = fd * r;
    const double rho = __builtin_fma(-q1, Zd, fd);
    return (float)__builtin_fma(rho, r, q1);
}

// one step of the recursion (reference ForwardBackward.hpp:88-112); alpha is updated in place.  The K divisions by the
// normaliser (an IEEE float division is eleven instructions: 55 of the step's ~125 at K = 5, and the filter is a chain of
// dependent steps) go through ONE double reciprocal where the normaliser is a positive finite float - the same floats
// (hml_tr2_quotient, round 3: first in the weakly compressed sweep's first pass, now in every form of the filter).
template <int K>
__device__ __forceinline__ bool hml_fwd_step(const hml_fwd_ctx<K>& cx, float (&alpha)[K], const float (&e)[K]) {
    float f[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        float tt = 0.0f;
#pragma unroll
        for (int i = 0; i < K; ++i) tt += alpha[i] * cx.A[i * K + j];
        f[j] = e[j] * tt;
    }
    float Z = 0.0f;
#pragma unroll
    for (int j = 0; j < K; ++j) Z += f[j];
    if (__builtin_expect(!(Z > 0.0f) || !(Z < 3.4028234663852886e38f), 0)) {   // 0: the uniform vector; negative, infinite or NaN: whatever the division says
        const bool ok = (Z != 0.0f);
#pragma unroll
        for (int j = 0; j < K; ++j) alpha[j] = ok ? f[j] / Z : cx.invK;
        return !ok;
    }
    const double Zd = (double)Z;
    const double r = hml_tr2_reciprocal(Zd);
#pragma unroll
    for (int j = 0; j < K; ++j) alpha[j] = hml_tr2_quotient(f[j], Zd, r);
    return false;
}

// Runs the recursion over blocks [b0, b1) in batches so that the (recursion-independent) loads of the
// emission terms are in flight together instead of one cache round trip per step.
template <int K, bool STORE>
__device__ __forceinline__ void hml_fwd_run(const hml_fwd_ctx<K>& cx, float (&alpha)[K], const float* __restrict__ em,
                                            const float* __restrict__ gsc, float* __restrict__ rows,
                                            float* __restrict__ aprobe, uint32_t b0, uint32_t b1, uint32_t& nfb,
                                            const hml_layout lay) {
    constexpr int BATCH = (K <= 6) ? 4 : 2;
    for (uint32_t b = b0; b < b1; b += BATCH) {
        float e[BATCH][K], g[BATCH][K];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const uint32_t bi = b + (uint32_t)i;
            const bool ok = bi < b1;
#pragma unroll
            for (int s = 0; s < K; ++s) {
                e[i][s] = ok ? em[hml_bk(lay, bi, K, s)] : 0.0f;
                g[i][s] = (STORE && ok && cx.self && gsc) ? gsc[hml_bk(lay, bi, K, s)] : 1.0f;   // (no plane: the rows stay unscaled)
            }
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const uint32_t bi = b + (uint32_t)i;
            if (bi < b1) {
                const bool fb = hml_fwd_step<K>(cx, alpha, e[i]);
                if (STORE) {
                    if (fb) nfb++;
                    const uint32_t t = bi + 1u;
#pragma unroll
                    for (int s = 0; s < K; ++s) {
                        rows[hml_bk(lay, bi, K, s)] = (cx.self && t < cx.B) ? alpha[s] * g[i][s] : alpha[s];
                        if (aprobe) aprobe[(uint64_t)t * K + s] = alpha[s];
                    }
                }
            }
        }
    }
}

// The speculative pass: every chunk warms up over the W blocks before it, stores the vector it then starts from
// (entry) and the one it ends in (exit).  Verification (entry[c] == exit[c-1], bit for bit) happens in the
// backward-map kernel, which reads the rows anyway; repairs in hml_fwd_repair.
template <int K>
__device__ __forceinline__ void hml_b_forward(const float* __restrict__ em, const float* __restrict__ gsc,
                                                     hml_model* __restrict__ mdl, float* __restrict__ rows,
                                                     float* __restrict__ aprobe, float* __restrict__ entry,
                                                     float* __restrict__ exitv, uint32_t* __restrict__ fb_count, int L,
                                                     const hml_layout lay) {
    __shared__ float sm_A[hml_amat<K>::LDS_FLOATS];
    hml_amat_fill<K>(sm_A, mdl, (int)threadIdx.x, (int)blockDim.x);
    hml_fwd_ctx<K> cx;
    hml_fwd_ctx_load<K>(cx, mdl, sm_A);
    const uint32_t B = cx.B;
    const int W = (int)mdl->fwd_W;   // adaptive, device-resident
    const uint32_t C = (B + (uint32_t)L - 1u) / (uint32_t)L;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nthreads = gridDim.x * blockDim.x;
    if (gid < (uint32_t)K && aprobe) aprobe[gid] = mdl->pi[gid];
    // grid-stride over chunks: correctness never depends on the launch size (B is only known on the device)
    for (uint32_t c = gid; c < C; c += nthreads) {
        const uint32_t first = c * (uint32_t)L;
        const uint32_t last = (first + (uint32_t)L < B) ? first + (uint32_t)L : B;   // one past
        const uint32_t ws = (first >= (uint32_t)W) ? first - (uint32_t)W : 0u;
        const bool exact = (ws == 0u);
        float alpha[K];
        uint32_t nfb = 0;
#pragma unroll
        for (int s = 0; s < K; ++s) alpha[s] = exact ? mdl->pi[s] : cx.invK;
        hml_fwd_run<K, false>(cx, alpha, em, gsc, rows, aprobe, ws, first, nfb, lay);   // warm-up, nothing stored
#pragma unroll
        for (int s = 0; s < K; ++s) entry[(uint64_t)c * K + s] = alpha[s];
        // the chunk proper over [first, last)
        hml_fwd_run<K, true>(cx, alpha, em, gsc, rows, aprobe, first, last, nfb, lay);
#pragma unroll
        for (int s = 0; s < K; ++s) exitv[(uint64_t)c * K + s] = alpha[s];
        // "[WARNING] Uniform sampling of forward variables!" events of this chunk
        fb_count[c] = nfb;
        if (nfb) atomicAdd(&mdl->uniform_fallbacks, (unsigned long long)nfb);
    }
}
// the kernel: hml_b_forward over one chain (hml_k_many.h runs it over several chains in one launch)
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_k_forward(const float* __restrict__ em, const float* __restrict__ gsc,
                                                     hml_model* __restrict__ mdl, float* __restrict__ rows,
                                                     float* __restrict__ aprobe, float* __restrict__ entry,
                                                     float* __restrict__ exitv, uint32_t* __restrict__ fb_count, int L,
                                                     const hml_layout lay) {
    hml_b_forward<K>(em, gsc, mdl, rows, aprobe, entry, exitv, fb_count, L, lay);
}


// a chunk started from pi itself (its warm-up window reaches block 0): exact by construction
__device__ __forceinline__ bool hml_fwd_chunk_exact(uint32_t c, int L, int W) { return c == 0u || c * (uint32_t)L <= (uint32_t)W; }

// a word another wavefront of this workgroup may have rewritten in this launch (the load bypasses the L1)
__device__ __forceinline__ uint32_t hml_ld_u32_coherent(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t hml_ld_bits_coherent(const float* p) {
    return hml_ld_u32_coherent(reinterpret_cast<const uint32_t*>(p));
}

template <int K>
__device__ __forceinline__ bool hml_fwd_chunk_consistent(const float* __restrict__ entry, const float* __restrict__ exitv, uint32_t c) {
    bool same = true;
#pragma unroll
    for (int s = 0; s < K; ++s)
        same = same && (hml_ld_bits_coherent(entry + (uint64_t)c * K + s) == hml_ld_bits_coherent(exitv + (uint64_t)(c - 1) * K + s));
    return same;
}

// ------------------------------------------------------------------------------------------
// Repair, by ONE workgroup, only when the verification found chunks whose start vector is not their
// predecessor's end vector.  The backward-map kernel lists the backward chunks it could not verify
// (fail_list, n_fail entries); all work here is proportional to that list, not to B.  Forward chunks are
// handled in windows of 2^17 (two LDS bitmaps):
//   1. the stale forward chunks of the listed backward chunks are marked and each is recomputed after a 2x
//      longer warm-up (one thread per chunk) - the neighbour is not trusted, it may be stale too; every
//      recomputed chunk and its successor are checked again, and what is still inconsistent repeats the step
//      with 4x, 8x, 16x: this ends in the true vectors unless the filter hardly forgets at all;
//   2. one lane visits the chunks that are inconsistent even then in increasing order, recomputes each from its predecessor's true end vector and follows the chain
//      while the recomputed end vector makes the next chunk inconsistent.  After this pass induction from
//      chunk 0 holds: the rows are the sequential ones.
// Every recomputed chunk puts its backward chunk on the touched list (LDS, and touched[] = gen in memory in
// case the list overflows) so that the caller recomputes its maps.
// ------------------------------------------------------------------------------------------
#define HML_REPAIR_WINDOW_WORDS 4096   // 2^17 forward chunks per window
#define HML_REPAIR_TOUCHED_CAP 2048
struct hml_repair_lds {
    uint32_t bad[HML_REPAIR_WINDOW_WORDS];    // stale after the speculative pass
    uint32_t bad2[HML_REPAIR_WINDOW_WORDS];   // still inconsistent after the long warm-up
    uint32_t tlist[HML_REPAIR_TOUCHED_CAP];
    uint32_t tcount;
    uint32_t next_first_bad;                  // the first chunk of the next window became inconsistent
    uint32_t any_bad;
    uint32_t any_marked, carry;               // this window holds a stale chunk / inherits one from the window before
};

template <int K>
__device__ void hml_fwd_repair(const float* __restrict__ em, const float* __restrict__ gsc, hml_model* __restrict__ mdl,
                               float* __restrict__ rows, float* __restrict__ aprobe, float* __restrict__ entry,
                               float* __restrict__ exitv, uint32_t* __restrict__ fb_count,
                               const uint32_t* __restrict__ fail_list, uint32_t n_fail, uint32_t* __restrict__ touched,
                               uint32_t gen, int L, const hml_layout lay, hml_repair_lds& sh, const float* lds_A) {
    hml_fwd_ctx<K> cx;
    hml_fwd_ctx_load<K>(cx, mdl, lds_A);
    const uint32_t B = cx.B;
    const int W = (int)mdl->fwd_W;
    const uint32_t C = (B + (uint32_t)L - 1u) / (uint32_t)L;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t win = HML_REPAIR_WINDOW_WORDS * 32u;
    const uint32_t nfper = HML_BWD_CHUNK / (uint32_t)L + 1u;   // forward chunks that can overlap one backward chunk
    auto account = [&](uint32_t c, uint32_t nfb) {
        const uint32_t old = fb_count[c];
        fb_count[c] = nfb;
        // keep the global tally of uniform fallbacks consistent (two's-complement delta on the unsigned counter)
        if (nfb != old) atomicAdd(&mdl->uniform_fallbacks, (unsigned long long)(long long)((int)nfb - (int)old));
        for (uint32_t b = c * (uint32_t)L; b < (c + 1u) * (uint32_t)L && b < B; b += HML_BWD_CHUNK) {
            const uint32_t bc = b / HML_BWD_CHUNK;
            touched[bc] = gen;
            const uint32_t k = atomicAdd(&sh.tcount, 1u);
            if (k < (uint32_t)HML_REPAIR_TOUCHED_CAP) sh.tlist[k] = bc;
        }
    };
    if (tid == 0) { mdl->fwd_serial_ran = 1u; sh.tcount = 0u; sh.next_first_bad = 0u; }   // (the parameter kernel lengthens the warm-up)
    __syncthreads();
    for (uint32_t w0 = 0; w0 < C; w0 += win) {
        const uint32_t w1 = (w0 + win < C) ? w0 + win : C;
        for (int i = tid; i < HML_REPAIR_WINDOW_WORDS; i += nthr) { sh.bad[i] = 0u; sh.bad2[i] = 0u; }
        if (tid == 0) { sh.any_marked = 0u; sh.carry = 0u; sh.any_bad = 0u; }
        __syncthreads();
        if (tid == 0 && sh.next_first_bad) { sh.bad2[0] = 1u; sh.next_first_bad = 0u; sh.carry = 1u; }
        // ---- 1a. stale forward chunks of the listed backward chunks
        for (uint64_t i = (uint64_t)tid; i < (uint64_t)n_fail * nfper; i += (uint64_t)nthr) {
            const uint32_t cb = fail_list[i / nfper];
            const uint32_t f = (cb * HML_BWD_CHUNK) / (uint32_t)L + (uint32_t)(i % nfper);
            if (f > (cb * HML_BWD_CHUNK + HML_BWD_CHUNK - 1u) / (uint32_t)L || f >= C || f < w0 || f >= w1) continue;
            // a forward chunk longer than a backward chunk is listed by each of them: the one that holds its first
            // block takes it
            if ((uint32_t)(((uint64_t)f * (uint32_t)L) / HML_BWD_CHUNK) != cb) continue;
            if (hml_fwd_chunk_exact(f, L, W)) continue;
            bool same = true;
#pragma unroll
            for (int s = 0; s < K; ++s) same = same && (hml_f2u(entry[(uint64_t)f * K + s]) == hml_f2u(exitv[(uint64_t)(f - 1) * K + s]));
            if (!same) { atomicOr(&sh.bad[(f - w0) >> 5], 1u << ((f - w0) & 31u)); sh.any_marked = 1u; }
        }
        __syncthreads();
        // a window without a stale chunk (nearly all of them, when a few chunks among millions failed) is done
        if (sh.any_marked == 0u && sh.carry == 0u) continue;   // workgroup-uniform
        // ---- 1b / 2a. recompute the marked chunks after a longer warm-up (one thread per chunk), check them and
        // their successors again (loads that bypass the L1), and escalate what is still inconsistent: 2W, 4W, 8W, 16W
        for (uint32_t factor = 2u; factor <= 16u; factor *= 2u) {
            for (uint32_t c = w0 + (uint32_t)tid; c < w1; c += (uint32_t)nthr) {
                if (((sh.bad[(c - w0) >> 5] >> ((c - w0) & 31u)) & 1u) == 0u) continue;
                if (hml_fwd_chunk_exact(c, L, W)) continue;   // (a successor that was only to be re-checked)
                const uint32_t first = c * (uint32_t)L;
                const uint32_t last = (first + (uint32_t)L < B) ? first + (uint32_t)L : B;
                const uint32_t Wl = (uint32_t)W * factor;
                const uint32_t ws = (first >= Wl) ? first - Wl : 0u;
                float alpha[K];
                uint32_t nfb = 0;
#pragma unroll
                for (int s = 0; s < K; ++s) alpha[s] = (ws == 0u) ? mdl->pi[s] : cx.invK;
                hml_fwd_run<K, false>(cx, alpha, em, gsc, rows, aprobe, ws, first, nfb, lay);
#pragma unroll
                for (int s = 0; s < K; ++s) entry[(uint64_t)c * K + s] = alpha[s];
                hml_fwd_run<K, true>(cx, alpha, em, gsc, rows, aprobe, first, last, nfb, lay);
#pragma unroll
                for (int s = 0; s < K; ++s) exitv[(uint64_t)c * K + s] = alpha[s];
                account(c, nfb);
                atomicAdd(&mdl->forward_refits, 1ull);
            }
            __threadfence_block();
            __syncthreads();
            if (tid == 0) sh.any_bad = 0u;
            __syncthreads();
            for (uint32_t c = w0 + (uint32_t)tid; c < w1; c += (uint32_t)nthr) {
                if (((sh.bad[(c - w0) >> 5] >> ((c - w0) & 31u)) & 1u) == 0u) continue;
                for (uint32_t cc = c; cc <= c + 1u && cc < C; ++cc) {
                    if (hml_fwd_chunk_exact(cc, L, W) || hml_fwd_chunk_consistent<K>(entry, exitv, cc)) continue;
                    if (cc < w1) { atomicOr(&sh.bad2[(cc - w0) >> 5], 1u << ((cc - w0) & 31u)); sh.any_bad = 1u; }
                    else sh.next_first_bad = 1u;
                }
            }
            __syncthreads();
            if (sh.any_bad == 0u || factor == 16u) break;   // workgroup-uniform
            // escalate: the still-inconsistent chunks become the marked ones
            for (int i = tid; i < HML_REPAIR_WINDOW_WORDS; i += nthr) { sh.bad[i] = sh.bad2[i]; sh.bad2[i] = 0u; }
            __syncthreads();
        }
        // ---- 2b. what is still inconsistent, serially
        if (tid == 0 && (sh.any_bad != 0u || sh.carry != 0u)) {
            for (uint32_t wi = 0; wi < (w1 - w0 + 31u) / 32u; ++wi) {
                uint32_t bits = sh.bad2[wi];
                while (bits) {
                    const int bit = __ffs(bits) - 1;
                    bits &= bits - 1u;
                    // repair chunk c and every following chunk that the repair makes inconsistent; the end vector
                    // of a repaired chunk stays in registers as the next chunk's true start vector
                    float alpha[K];
                    bool have_alpha = false;
                    for (uint32_t c = w0 + wi * 32u + (uint32_t)bit; c < C; ++c) {
                        bool same = true;
#pragma unroll
                        for (int s = 0; s < K; ++s) {
                            if (!have_alpha) alpha[s] = hml_u2f(hml_ld_bits_coherent(exitv + (uint64_t)(c - 1) * K + s));
                            same = same && (hml_f2u(alpha[s]) == hml_ld_bits_coherent(entry + (uint64_t)c * K + s));
                        }
                        if (same) break;   // consistent (possibly repaired already by an earlier chain)
                        const uint32_t first = c * (uint32_t)L;
                        const uint32_t last = (first + (uint32_t)L < B) ? first + (uint32_t)L : B;
#pragma unroll
                        for (int s = 0; s < K; ++s) entry[(uint64_t)c * K + s] = alpha[s];
                        uint32_t nfb = 0;
                        hml_fwd_run<K, true>(cx, alpha, em, gsc, rows, aprobe, first, last, nfb, lay);
#pragma unroll
                        for (int s = 0; s < K; ++s) exitv[(uint64_t)c * K + s] = alpha[s];
                        have_alpha = true;
                        account(c, nfb);
                        atomicAdd(&mdl->forward_serial, 1ull);
                    }
                }
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

#endif
