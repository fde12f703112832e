// Models of more than 16 states, filter and backward draws with A CHUNK A LANE (round 5; hml_k_wide.h has a STATE a lane: a
// wavefront per chunk, 20 of its 64 lanes at work on a model of 20 states, and every sum over the states a chain of
// v_readlane broadcasts - 470 cycles of a SIMD per block and chunk step at 20 states).  Here a wavefront runs 64 chunks side
// by side, every lane the whole K-vector of its own chunk - the shape of the kernels for up to 16 states (hml_k_forward.h) with
// the number of states at run time:
//   * the per-block K-vectors (emission terms, rescale factors, trellis rows) are CHUNK-TRANSPOSED, a TILE of 64 chunks at a
//     time (hml_wl_at): element (block b, state s) of chunk c = b / L, row r = b mod L, lies at (((c / 64) L + r) K + s) 64 +
//     c mod 64 - the 64 lanes of a wavefront, 64 consecutive chunks at the same row of their chunk, read and write 64
//     consecutive floats, and a wavefront's steps walk ONE contiguous stream of L K 256 bytes (the layout of hml_bk, whole
//     planes per (row, state), had every one of a step's 3 K accesses in a DRAM page of its own: 2.5 TB/s).  The chunk length L (a
//     power of two) is decided on the device once the sweep's blocks are known (hml_k_wl_prepare: the shortest
//     chunks, at least four blocks, that make at most 65 536 chunks - 131 072 up to 36 states);
//   * the transition matrix comes from SCALAR loads (the same for every lane: a row of a zero-padded 64 x 64 copy, sixteen
//     columns at a time), the lane's own vector of the step before from its column of LDS (the accumulators are registers:
//     static indices), K x K multiply-adds per lane and step and nothing that crosses lanes;
//   * the backward draws: the lane's weights w_i = row_i A(i, q) with A's column q from LDS, the categorical's running double
//     sums in registers, the same screen as hml_compat_categorical_wave (and the literal form for the draw in 10^6 it does not
//     settle).
// Same arithmetic, operation for operation, as the lane-per-state kernels (sums over i and j in index order, IEEE quotients -
// through the double reciprocal where the normaliser is a positive finite float, hml_tr2_quotient), and the same proof of
// equality with the sequential recursion: a chunk starts W blocks early from a guess and is accepted only if what it reached
// at its first block equals, bit for bit, what the chunk before it left there; the rare chunk that fails runs again (by one
// wavefront, a state a lane).  tests/test_gpu_parity.py::test_sweeps_match_checker (17 .. 64 states), the wide fuzz.
// Reference: src/StateSequence/ForwardBackward.hpp:86-162, src/Trellis.hpp:61-66.
#ifndef HML_K_WIDE_LANES_H
#define HML_K_WIDE_LANES_H

#include "hml_k_wide.h"

#if defined(__HIPCC__)

#define HML_WL_MAX_CHUNKS 131072
#define HML_WL_MIN_LSHIFT 2
#define HML_WL_MAP_WORDS (HML_WL_MAX_CHUNKS / 64)
#define HML_WL_PITCH 64   // floats between the rows of the padded transition matrix
// the filter's registers let two wavefronts onto a SIMD up to this many (padded) states - up to 32 as the compiler allocates them, at
// 36 when told to and with the step's loads asked for behind the products (219 registers; 40 states fit too, 240 registers, and gain
// nothing; from 48 on the kernel would spill) - and a sweep is then cut into twice the chunks (hml_k_wl_prepare)
#define HML_WL_TWO_WAVES_KC 36
// hml_compat_chunks::tot on this path: [0] wrong chunks of the filter, [1] the sum of nfb, [2] wrong chunks of the backward draws,
// then a bit per wrong chunk of the filter and of the backward draws (set by the verifying launches, cleared by the checking ones)
#define HML_WL_MAP_F 4
#define HML_WL_MAP_B (HML_WL_MAP_F + HML_WL_MAP_WORDS)
#define HML_WL_TOT_WORDS (HML_WL_MAP_B + HML_WL_MAP_WORDS)

// element (block b, state s) of a chunk-transposed array (chunks of 1 << lshift blocks)
__device__ __forceinline__ uint64_t hml_wl_at(const uint32_t lshift, const int K, const uint32_t b, const int s) {
    const uint32_t c = b >> lshift, r = b & ((1u << lshift) - 1u);
    return ((((uint64_t)(c >> 6) << lshift) + r) * (uint32_t)K + (uint32_t)s) * 64u + (c & 63u);
}
__device__ __forceinline__ hml_layout hml_wl_layout(const hml_model* mdl) {
    hml_layout lay;
    lay.lshift = mdl->wl_lshift;
    lay.cstride = mdl->wl_cstride;
    return lay;
}

// The sweep's geometry and the padded copy of the transition matrix (one workgroup, behind the block enumeration).
// force_lshift >= 0: that chunk length (tests), raised if it would make more than max_chunks chunks.  max_chunks: 65 536 - a
// wavefront for each of the machine's 1024 SIMDs - where the filter's registers let one wavefront onto a SIMD (more than 32 states),
// twice that where they let two (measured at 20 / 40 / 64 states with 65 536, 131 072, 262 144: 0.69 / 1.67 / 3.51, 0.62 / 1.95 / 4.13,
// 0.68 / 2.20 / 4.62 ms per sweep).
HML_KERNEL __launch_bounds__(256) void hml_k_wl_prepare(hml_model* mdl, float* __restrict__ wA, int force_lshift, uint32_t max_chunks) {
    if (mdl->halted != 0u) return;
    const int K = mdl->K;
    for (int idx = threadIdx.x; idx < HML_WL_PITCH * HML_WL_PITCH; idx += 256) {
        const int i = idx / HML_WL_PITCH, j = idx % HML_WL_PITCH;
        wA[idx] = (i < K && j < K) ? mdl->A[i * K + j] : 0.0f;
    }
    if (threadIdx.x == 0) {
        const uint32_t B = mdl->B;
        uint32_t sh = force_lshift >= 0 ? (uint32_t)force_lshift : (uint32_t)HML_WL_MIN_LSHIFT;
        while ((((uint64_t)B + (1ull << sh) - 1ull) >> sh) > (uint64_t)max_chunks) ++sh;
        const uint32_t C = (uint32_t)(((uint64_t)B + (1ull << sh) - 1ull) >> sh);
        mdl->wl_lshift = sh;
        mdl->wl_cstride = (C + 63u) / 64u * 64u + (C == 0u ? 64u : 0u);
    }
}

// The rescale factors g_s(N) = expf((N - 1) log A(s, s)) of the block sizes below HML_WL_GTAB, [s][N]: the emission kernel looks
// a block's K factors up instead of evaluating them (40 of its 142 instructions per block and state; the same expression, the same
// floats) - a model's blocks are mostly short (of 3.4 10^6 blocks at 64 states on config 3's trace none reaches 2048 positions).
#define HML_WL_GTAB 2048
HML_KERNEL __launch_bounds__(256) void hml_k_wl_gtable(const hml_model* __restrict__ mdl, float* __restrict__ gtab) {
    if (mdl->halted != 0u || mdl->self_trans == 0) return;
    const int K = mdl->K;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < (uint32_t)K * HML_WL_GTAB; idx += gridDim.x * blockDim.x) {
        const uint32_t st = idx / HML_WL_GTAB, n = idx % HML_WL_GTAB;
        gtab[idx] = hml_expf_tab(((float)n - 1.0f) * mdl->logA[st], HML_EXP2F_TAB);
    }
}

// Emission terms and rescale factors in the chunk-transposed layout: hml_k_wide_emission's values (same arithmetic, same order),
// a lane per CHUNK: a wavefront takes 64 consecutive chunks and up to four of their rows, and every store is 64 consecutive
// floats.  The rows go through the states together - a state's parameters are read once for four independent chains of
// arithmetic - and twice: first for the rows' maxima, then for the terms themselves (the energies are cheap to compute again,
// and a lane has no room to keep 64 of them per row).
#define HML_WL_EMIT_ROWS 4
HML_KERNEL __launch_bounds__(256) void hml_k_wl_emission(hml_model* __restrict__ mdl, const uint32_t* __restrict__ starts, const float2* __restrict__ bstat,
                                                         float* __restrict__ em, float* __restrict__ g, const float* __restrict__ gtab) {
    __shared__ float s_mu[HML_CAP_K], s_var[HML_CAP_K], s_logNs[HML_CAP_K], s_logA[HML_CAP_K];
    __shared__ double s_rvar[HML_CAP_K];
    __shared__ uint8_t s_map[HML_CAP_K][HML_MAX_D];
    __shared__ uint64_t s_tab[32];
    if (mdl->halted != 0u) return;
    const uint32_t B = mdl->B;
    if (B == 0u) return;
    const int K = mdl->K, D = mdl->D, P = mdl->P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool self = mdl->self_trans != 0;
    const uint64_t dstride = mdl->stat_stride;
    for (int k = tid; k < P; k += 256) { s_mu[k] = mdl->mu[k]; s_var[k] = mdl->var[k]; s_rvar[k] = mdl->rvar2[k]; }
    for (int k = tid; k < K; k += 256) {
        s_logNs[k] = mdl->logNs[k]; s_logA[k] = mdl->logA[k];
        for (int d = 0; d < HML_MAX_D; ++d) s_map[k][d] = mdl->map[k][d];
    }
    for (int k = tid; k < 32; k += 256) s_tab[k] = HML_EXP2F_TAB[k];
    __syncthreads();
    const hml_layout lay = hml_wl_layout(mdl);
    const uint32_t L = 1u << lay.lshift, C = (uint32_t)(((uint64_t)B + L - 1u) >> lay.lshift);
    constexpr int R = HML_WL_EMIT_ROWS;
    const uint32_t RG = L < (uint32_t)R ? L : (uint32_t)R, GPT = L / RG;
    const uint64_t n_items = (uint64_t)((C + 63u) / 64u) * GPT;
    const uint32_t n_waves = gridDim.x * 4u;
    for (uint64_t item = blockIdx.x * 4u + (uint32_t)wave; item < n_items; item += n_waves) {   // wave-uniform
        const uint32_t tile = (uint32_t)(item / GPT), r0 = (uint32_t)(item % GPT) * RG;
        const uint32_t c = tile * 64u + (uint32_t)lane;
        const uint64_t bfirst = ((uint64_t)c << lay.lshift) + r0;
        float N[R], sx[R][HML_MAX_D], sq[R][HML_MAX_D], maxE[R];
        uint32_t Nt[R];   // the block's size as an index into the table of factors (HML_WL_GTAB - 1: not in it)
        bool in[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t b = bfirst + (uint32_t)r;
            in[r] = (uint32_t)r < RG && b < (uint64_t)B;
            const uint32_t bl = in[r] ? (uint32_t)b : B - 1u;   // (rows beyond the last block compute on it and store nothing)
            const uint32_t Ni = starts[bl + 1u] - starts[bl];
            N[r] = (float)Ni;   // (size_t N, converted where it meets a float)
            Nt[r] = Ni < (uint32_t)HML_WL_GTAB - 1u ? Ni : (uint32_t)HML_WL_GTAB - 1u;
#pragma unroll
            for (int d = 0; d < HML_MAX_D; ++d) {
                const float2 v = bstat[(uint64_t)(d < D ? d : 0) * dstride + bl];
                sx[r][d] = v.x; sq[r][d] = v.y;
            }
            maxE[r] = -3.40282346638528859812e+38f;
        }
        // E_s of row r: innerProduct(y, theta.value(), theta.mapping(s)) - a float sum over the dimensions from 0 (EFD.hpp:83-93) -
        // minus N logNormalizer [+ (N - 1) log A(s, s)]
        auto energies = [&](const int st, float (&E)[R], const bool first) {
            float rr[R];
#pragma unroll
            for (int r = 0; r < R; ++r) rr[r] = 0.0f;
#pragma unroll
            for (int d = 0; d < HML_MAX_D; ++d) {
                if (d < D) {
                    const int pp = s_map[st][d];
                    const float mu = s_mu[pp], var = s_var[pp];
                    const double rvar = s_rvar[pp];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float ip = hml_inner_product(mu, var, rvar, sx[r][d], sq[r][d]);
                        if (first && in[r] && !hml_isfinite(ip)) hml_raise(mdl, HML_DEVERR_IP_NOT_FINITE, ip);
                        rr[r] += ip;
                    }
                }
            }
            const float lN = s_logNs[st], lA = s_logA[st];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float e = rr[r] - N[r] * lN;
                if (self) e += (N[r] - 1.0f) * lA;
                E[r] = e;
            }
        };
        for (int st = 0; st < K; ++st) {
            float E[R];
            energies(st, E, true);
#pragma unroll
            for (int r = 0; r < R; ++r) maxE[r] = (E[r] < maxE[r]) ? maxE[r] : E[r];
        }
        const uint64_t a00 = ((((uint64_t)tile << lay.lshift) + r0) * (uint32_t)K) * 64u + (uint32_t)lane;
        for (int st = 0; st < K; ++st) {
            float E[R];
            energies(st, E, false);
            const float lA = s_logA[st];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float v = hml_expf_tab(E[r] - maxE[r], s_tab);
                float gv = 1.0f;
                if (self) {
                    gv = gtab[(uint32_t)st * HML_WL_GTAB + Nt[r]];
                    if (__builtin_expect(Nt[r] == (uint32_t)HML_WL_GTAB - 1u, 0)) gv = hml_expf_tab((N[r] - 1.0f) * lA, s_tab);   // (a long block)
                }
                if (in[r]) {
                    const uint64_t a = a00 + (uint64_t)((uint32_t)r * (uint32_t)K + (uint32_t)st) * 64u;
                    em[a] = v;
                    if (self) g[a] = gv;
                }
            }
        }
    }
}

// one pass of the filter's matrix-vector product over NJ columns from j0: out[j] = sum_i prev_i A(i, j0 + j), i = 0 .. K-1 in
// order from 0.0f (products and sums rounded separately, like `tt += prev_i * A(i, j)` everywhere else); prev from the lane's
// column of LDS, A's rows from the workgroup's padded copy in LDS (every lane the same address: broadcast reads of 16 bytes,
// several rows in flight - scalar loads of the rows were measured first: a wavefront alone on its SIMD waited 100 ns for
// each of them, 40 times a step)
template <int NJ>
__device__ __forceinline__ void hml_wl_mv(const float* sA, const int j0, const float* sp, const int K, float (&out)[NJ]) {
    static_assert(NJ % 4 == 0, "columns in groups of four");
    constexpr int Q = NJ / 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) out[j] = 0.0f;
    // two rows a stage, the next stage's reads issued before this stage's multiply-adds (left to itself the compiler, short of
    // registers, waited for every 16-byte read before the four instructions that use it: an LDS round trip per read)
    float4 a0[Q], a1[Q], b0[Q], b1[Q];
    float p0, p1, q0, q1;
    auto fetch = [&](const int i, float4 (&r0)[Q], float4 (&r1)[Q], float& x0, float& x1) {
        const int i1 = (i + 1 < K) ? i + 1 : i;
        const float4* const row0 = (const float4*)(sA + i * HML_WL_PITCH + j0);
        const float4* const row1 = (const float4*)(sA + i1 * HML_WL_PITCH + j0);
#pragma unroll
        for (int k = 0; k < Q; ++k) { r0[k] = row0[k]; r1[k] = row1[k]; }
        x0 = sp[i * 64]; x1 = sp[i1 * 64];
    };
    auto madd = [&](const float4 (&r)[Q], const float x) {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            out[4 * k + 0] += x * r[k].x;
            out[4 * k + 1] += x * r[k].y;
            out[4 * k + 2] += x * r[k].z;
            out[4 * k + 3] += x * r[k].w;
        }
    };
    fetch(0, a0, a1, p0, p1);
    int i = 0;
    for (; i + 4 <= K; i += 4) {   // rows i .. i + 3; rows i + 4, i + 5 (if any) on their way at the end
        fetch(i + 2, b0, b1, q0, q1);
        madd(a0, p0); madd(a1, p1);
        const int in = (i + 4 < K) ? i + 4 : i;
        fetch(in, a0, a1, p0, p1);
        madd(b0, q0); madd(b1, q1);
    }
    // up to three rows left (a0, a1 hold rows i, i + 1 when i < K)
    if (i < K) {
        if (i + 2 < K) fetch(i + 2, b0, b1, q0, q1);
        madd(a0, p0);
        if (i + 1 < K) madd(a1, p1);
        if (i + 2 < K) madd(b0, q0);
    }
}

// The warm-up of the filter's chunks on this path (mdl->fwd_W; results never depend on it).  It starts at 64 blocks.  A sweep with
// wrong chunks doubles it (eight times, if the whole filter ran again: hml_k_wl_retry_decide); sweeps without one walk it down -
// by half every four of them while it is above 64 (a young chain's first sweeps need a long warm-up once: parameters from the prior),
// by a quarter every sixteen below that, to the floor (32) - but not below one step (8 blocks) above a warm-up that failed on a settled
// chain (from its ninth sweep on) for the 512 sweeps after that.
__device__ __forceinline__ void hml_wl_fwd_adapt(hml_model* mdl, const uint32_t Wspec, const bool failed, const uint32_t W_used, const uint32_t W_next) {
    if (Wspec != HML_CHUNK_W_ADAPTIVE) return;
    if (failed) {
        if (mdl->sweeps >= 8ull) { mdl->wl_W_need = W_used; mdl->wl_need_age = 0u; }
        mdl->fwd_W = W_next;
        mdl->fwd_quiet = 0u;
        return;
    }
    if (mdl->wl_W_need != 0u && ++mdl->wl_need_age > 512u) mdl->wl_W_need = 0u;
    uint32_t floor_w = mdl->fwd_W0;
    if (mdl->wl_W_need != 0u) { const uint32_t keep = ((mdl->wl_W_need + 8u) & ~7u) < 1024u ? ((mdl->wl_W_need + 8u) & ~7u) : 1024u; floor_w = keep > floor_w ? keep : floor_w; }
    uint32_t w = mdl->fwd_W;
    const uint32_t quiet = ++mdl->fwd_quiet;
    if (w > 64u && w / 2u >= floor_w && quiet >= 4u) { w = w / 2u; mdl->fwd_quiet = 0u; }
    else if (quiet >= 16u) {
        const uint32_t lower = (w - w / 4u) & ~7u;
        w = lower > floor_w ? lower : (w < floor_w ? w : floor_w);
        mdl->fwd_quiet = 0u;
    }
    mdl->fwd_W = w;
}

// The filter, a chunk a lane.  KC: the model's number of states rounded up to a multiple of four (4 .. 64).
// Step s of a wavefront's 64 chunks is block lo + s - W of every one of them: the same row of the chunk-transposed arrays for
// all lanes (a scalar offset) and the lane's own chunk index plus a common shift - a load or store is a scalar base and a
// 32-bit lane offset.  Lanes whose block lies outside the trace (the first chunks' warm-up, the last chunk's tail) read some
// element inside the arrays and keep what they have.
template <int KC>
HML_KERNEL __launch_bounds__(64, ((KC > 32 && KC <= HML_WL_TWO_WAVES_KC) ? 2 : 1)) void hml_k_wl_forward(hml_model* __restrict__ mdl, const float* __restrict__ wA, const float* __restrict__ em,
                                                       const float* __restrict__ g, float* __restrict__ rows, const hml_compat_chunks ch, const int retry) {
    __shared__ float s_prev[KC * 64];   // [state][lane]: the lane's row of the step before
    __shared__ __attribute__((aligned(16))) float sA[KC * HML_WL_PITCH];
    if (mdl->halted != 0u) return;
    if (retry && mdl->wl_retry == 0u) return;   // (the second launch of a sweep: only if hml_k_wl_retry_decide asked for it)
    const int lane = threadIdx.x;
    const int K = mdl->K;
    const uint32_t B = mdl->B;
    const hml_layout lay = hml_wl_layout(mdl);
    const uint32_t L = 1u << lay.lshift, C = (uint32_t)(((uint64_t)B + L - 1u) >> lay.lshift);
    if (!retry && blockIdx.x == 0u && lane < 3) ch.tot[lane] = 0ull;   // (the verifying launches behind this one count into them)
    if (blockIdx.x * 64u >= C) return;
    const bool self = mdl->self_trans != 0;
    const uint32_t W = hml_chunk_warmup(mdl, ch.W);
    const float invK = (float)(1.0 / (double)(float)K);
    float* const sp = s_prev + lane;
    for (int idx = lane; idx < K * (HML_WL_PITCH / 4); idx += 64) ((float4*)sA)[idx] = ((const float4*)wA)[idx];
    hml_compat_fence();
    for (uint32_t c0 = blockIdx.x * 64u; c0 < C; c0 += gridDim.x * 64u) {   // wave-uniform
        const uint32_t c = c0 + (uint32_t)lane;
        const bool valid = c < C;
        const int64_t lo = (int64_t)c << lay.lshift;
        const int64_t hi = (lo + (int64_t)L < (int64_t)B) ? lo + (int64_t)L : (int64_t)B;
        const bool exact = lo <= (int64_t)W;   // the warm-up reaches block 0: the chunk starts from pi itself
#pragma unroll
        for (int j = 0; j < KC; ++j) sp[j * 64] = (j < KC - 3 || j < K) ? (exact ? mdl->pi[j] : invK) : 0.0f;
        uint32_t nfb = 0u;
        // step d: the lane's element of state 0 - the row of the chunk (the same for all lanes) in the tile of the lane's chunk
        // index, kept inside the arrays
        auto at = [&](int64_t d) -> uint64_t {
            const int64_t cc = (int64_t)c + (d >> lay.lshift);
            const uint32_t cl = cc < 0 ? 0u : (cc >= (int64_t)C ? C - 1u : (uint32_t)cc);
            return ((((uint64_t)(cl >> 6) << lay.lshift) + ((uint32_t)d & (L - 1u))) * (uint32_t)K) * 64u + (cl & 63u);
        };
        float e[KC];
        {
            const float* const pe = em + at(-(int64_t)W);
#pragma unroll
            for (int j = 0; j < KC; ++j) e[j] = (j < KC - 3 || j < K) ? pe[j * 64] : 0.0f;
        }
        auto step = [&](const int64_t d, auto own_tag) {
            constexpr bool OWN = decltype(own_tag)::value;
            const int64_t b = lo + d;
            const bool active = valid && b >= 0 && b < hi;
            // the next step's terms and this step's factors travel during the step - asked for in front of the products, or (LATE) behind
            // them: during the products their registers are needed, and 36 states fit two wavefronts per SIMD only so
            float en[KC], gc[KC];
            const uint64_t a0 = at(d);   // (an active lane's own element)
            auto ask = [&]() {
            {
                const float* const pe = em + at(d + 1);
#pragma unroll
                for (int j = 0; j < KC; ++j) en[j] = (j < KC - 3 || j < K) ? pe[j * 64] : 0.0f;
            }
            if (OWN && self) {
                const float* const pg = g + a0;
#pragma unroll
                for (int j = 0; j < KC; ++j) gc[j] = (j < KC - 3 || j < K) ? pg[j * 64] : 1.0f;
            }
            };
            constexpr bool LATE = KC > 32 && KC <= HML_WL_TWO_WAVES_KC;
            if constexpr (!LATE) ask();
            float f[KC];
#define HML_WL_PASS(J0, NJ)                                                              \
            if constexpr ((NJ) > 0) {                                                    \
                float o[(NJ) > 0 ? (NJ) : 4];                                            \
                hml_wl_mv<((NJ) > 0 ? (NJ) : 4)>(sA, J0, sp, K, o);                      \
                _Pragma("unroll") for (int j = 0; j < (NJ); ++j) f[(J0) + j] = e[(J0) + j] * o[j]; \
            }
            // (passes of at most sixteen columns: a pass's accumulators and two stages of A's rows are registers next to e, en, g)
            constexpr int NP = (KC + 15) / 16, PW = ((KC + NP - 1) / NP + 3) / 4 * 4;
            constexpr int W0 = PW < KC ? PW : KC, W1 = (KC - W0) < PW ? (KC - W0) : PW, W2 = (KC - W0 - W1) < PW ? (KC - W0 - W1) : PW, W3 = KC - W0 - W1 - W2;
            HML_WL_PASS(0, W0) HML_WL_PASS(W0, W1) HML_WL_PASS(W0 + W1, W2) HML_WL_PASS(W0 + W1 + W2, W3)
#undef HML_WL_PASS
            if constexpr (LATE) ask();
            float Z = 0.0f;
#pragma unroll
            for (int j = 0; j < KC; ++j) Z += f[j];   // (the padded terms are +0.0)
            // f / Z: the IEEE quotient through one double reciprocal where Z is a positive finite float (hml_fwd_step)
            const double Zd = (double)Z;
            const double rz = hml_tr2_reciprocal(Zd);
            float fw[KC];
#pragma unroll
            for (int j = 0; j < KC; ++j) fw[j] = hml_tr2_quotient(f[j], Zd, rz);
            if (__builtin_expect(active && (!(Z > 0.0f) || !(Z < 3.4028234663852886e38f)), 0)) {   // 0: the uniform vector; negative, infinite or NaN: whatever the division says
                const bool ok = (Z != 0.0f);
#pragma unroll
                for (int j = 0; j < KC; ++j) fw[j] = ok ? f[j] / Z : invK;
                if (OWN && !ok) nfb++;
            }
            if (active) {
#pragma unroll
                for (int j = 0; j < KC; ++j) if (j < KC - 3 || j < K) sp[j * 64] = fw[j];
                if (OWN) {
                    const bool scaled = self && (b + 1 < (int64_t)B);
                    float* const pr = rows + a0;
#pragma unroll
                    for (int j = 0; j < KC; ++j) if (j < KC - 3 || j < K) pr[j * 64] = scaled ? fw[j] * gc[j] : fw[j];
                }
            }
#pragma unroll
            for (int j = 0; j < KC; ++j) e[j] = en[j];
        };
        for (int64_t d = -(int64_t)W; d < 0; ++d) step(d, std::false_type{});
        if (valid) {
            for (int j = 0; j < K; ++j) ch.entry[(uint64_t)c * K + j] = sp[j * 64];
        }
        for (int64_t d = 0; d < (int64_t)L; ++d) step(d, std::true_type{});
        if (valid) {
            for (int j = 0; j < K; ++j) ch.exitv[(uint64_t)c * K + j] = sp[j * 64];
            ch.nfb[c] = nfb;
        }
    }
}

// which chunks started from another row than the chunk before them left (hml_k_compat_forward_verify with this path's chunks)
HML_KERNEL __launch_bounds__(256) void hml_k_wl_forward_verify(const hml_model* __restrict__ mdl, const hml_compat_chunks ch, const int retry) {
    if (mdl->halted != 0u) return;
    if (retry && mdl->wl_retry == 0u) return;
    const uint32_t B = mdl->B, K = (uint32_t)mdl->K;
    const uint32_t lshift = mdl->wl_lshift;
    const uint32_t n_chunks = (uint32_t)(((uint64_t)B + (1ull << lshift) - 1ull) >> lshift);
    const uint64_t n = (uint64_t)n_chunks * K;
    const uint32_t W = hml_chunk_warmup(mdl, ch.W);
    // (the one workgroup that walks the chunks in order returns at once when these say there is nothing to do: it took 100 us to
    // look at 47 000 flags)
    unsigned long long n_bad = 0ull, nfb = 0ull;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)(e / K);
        if (c > 0u && ((uint64_t)c << lshift) > W && hml_f2u(ch.entry[e]) != hml_f2u(ch.exitv[e - K])) {
            atomicOr(&ch.tot[HML_WL_MAP_F + (c >> 6)], 1ull << (c & 63u));
            n_bad++;
        }
        if (e == (uint64_t)c * K) nfb += (unsigned long long)ch.nfb[c];
    }
    if (__ballot(n_bad != 0ull || nfb != 0ull) != 0ull) {   // (rare)
        for (int m = 32; m >= 1; m >>= 1) { n_bad += __shfl_xor(n_bad, m); nfb += __shfl_xor(nfb, m); }
        if ((threadIdx.x & 63u) == 0u) {
            if (n_bad) atomicAdd(&ch.tot[0], n_bad);
            if (nfb) atomicAdd(&ch.tot[1], nfb);
        }
    }
}

// A sweep whose filter left MANY chunks wrong (a young chain: parameters from the prior, rows that forget slowly) does not hand them
// to the one wavefront that runs wrong chunks again in order - 40 000 chunks of 64 blocks took it 1.4 s - but runs the whole
// filter once more, a chunk a lane, with eight times the warm-up, which then is the model's for the sweeps that follow (the
// adaptation walks it down again).  One workgroup between the first verification and the second launch, which returns at once
// unless asked for.  A warm-up fixed by the caller (tests) stays as it is.
#define HML_WL_RETRY_MIN 32u
HML_KERNEL __launch_bounds__(256) void hml_k_wl_retry_decide(hml_model* __restrict__ mdl, const hml_compat_chunks ch) {
    if (mdl->halted != 0u) return;
    const unsigned long long n_bad = ch.tot[0];
    const uint32_t W = hml_chunk_warmup(mdl, ch.W);
    const bool take = ch.W == HML_CHUNK_W_ADAPTIVE && n_bad > (unsigned long long)HML_WL_RETRY_MIN && W < 1024u;   // (workgroup-uniform)
    if (take) {
        for (uint32_t w = threadIdx.x; w < (uint32_t)HML_WL_MAP_WORDS; w += 256u) ch.tot[HML_WL_MAP_F + w] = 0ull;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t nw = (8u * W < 1024u) ? 8u * W : 1024u;
            ch.tot[0] = 0ull; ch.tot[1] = 0ull;
            mdl->forward_refits += n_bad;
            hml_wl_fwd_adapt(mdl, ch.W, true, W, nw);
            mdl->wl_retry = nw;
        }
    } else if (threadIdx.x == 0) mdl->wl_retry = 0u;
}

// which chunks of the backward draws started from another state than the chunk above them ended in (or met a negative weight)
__device__ __forceinline__ bool hml_wl_backward_wrong(const hml_compat_chunks& ch, uint32_t cl, uint32_t n_chunks) {
    const int in = ch.in_state[cl];
    if (in == -2) return true;
    return cl + 1u < n_chunks && in >= 0 && in != ch.out_state[cl + 1u];
}
HML_KERNEL __launch_bounds__(256) void hml_k_wl_backward_verify(const hml_model* __restrict__ mdl, const hml_compat_chunks ch) {
    if (mdl->halted != 0u) return;
    const uint32_t B = mdl->B;
    const uint32_t lshift = mdl->wl_lshift;
    const uint32_t n_chunks = (uint32_t)(((uint64_t)B + (1ull << lshift) - 1ull) >> lshift);
    unsigned long long n_bad = 0ull;
    for (uint32_t cl = blockIdx.x * blockDim.x + threadIdx.x; cl < n_chunks; cl += gridDim.x * blockDim.x) {
        if (hml_wl_backward_wrong(ch, cl, n_chunks)) { atomicOr(&ch.tot[HML_WL_MAP_B + (cl >> 6)], 1ull << (cl & 63u)); n_bad++; }
    }
    if (__ballot(n_bad != 0ull) != 0ull) {
        for (int m = 32; m >= 1; m >>= 1) n_bad += __shfl_xor(n_bad, m);
        if ((threadIdx.x & 63u) == 0u) atomicAdd(&ch.tot[2], n_bad);
    }
}

// blocks [lo, hi) of the filter again from `prev`, a state a lane (the rare chunk that started from the wrong row): the
// arithmetic of hml_compat_forward_range over the chunk-transposed arrays.  KS: 32 or 64 (loops in groups of four up to K).
template <int KS>
__device__ __forceinline__ void hml_wl_forward_again(const int K, const uint32_t B, const bool self, const hml_layout lay, const float* __restrict__ em,
                                                     const float* __restrict__ g, float* __restrict__ rows, const float (&acol)[KS], const uint32_t lo,
                                                     const uint32_t hi, float& prev, uint32_t& nfb, const int lane) {
    const bool act = lane < K;
    const int sl = act ? lane : 0;
    for (uint32_t b = lo; b < hi; ++b) {
        const uint64_t a = hml_wl_at(lay.lshift, K, b, sl);
        const float ev = em[a];
        const float gv = self ? g[a] : 1.0f;
        float tt = 0.0f;
#pragma unroll
        for (int i0 = 0; i0 < KS; i0 += 4) {
            if (i0 < K) {
#pragma unroll
                for (int i = i0; i < i0 + 4; ++i) tt += hml_lane_f32(prev, i) * acol[i];
            }
        }
        const float f = act ? ev * tt : 0.0f;
        float Z = 0.0f;
#pragma unroll
        for (int j0 = 0; j0 < KS; j0 += 4) {
            if (j0 < K) {
#pragma unroll
                for (int j = j0; j < j0 + 4; ++j) Z += hml_lane_f32(f, j);
            }
        }
        float fw;
        if (Z != 0.0f) fw = f / Z;
        else { nfb++; fw = (float)(1.0 / (double)(float)K); }
        if (act) rows[a] = (self && (uint64_t)b + 1u < (uint64_t)B) ? fw * gv : fw;
        prev = act ? fw : 0.0f;
    }
}

// the filter's chunks in order: a chunk whose first row is not what the chunk before it left runs again from that row
// (hml_k_compat_forward_check with this path's chunks and arrays)
template <int KS>
HML_KERNEL __launch_bounds__(256) void hml_k_wl_forward_check(hml_model* __restrict__ mdl, const float* __restrict__ em, const float* __restrict__ g,
                                                              float* __restrict__ rows, const hml_compat_chunks ch) {
    __shared__ unsigned long long map[HML_WL_MAP_WORDS];
    if (mdl->halted != 0u) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = mdl->K;
    const uint32_t B = mdl->B;
    const hml_layout lay = hml_wl_layout(mdl);
    const uint32_t L = 1u << lay.lshift;
    const uint32_t n_chunks = (uint32_t)(((uint64_t)B + L - 1u) >> lay.lshift);
    if (ch.tot[0] == 0ull) {   // every chunk started from what the chunk before it left (hml_k_wl_forward_verify)
        if (tid == 0) { mdl->uniform_fallbacks += ch.tot[1]; if (mdl->wl_retry == 0u) hml_wl_fwd_adapt(mdl, ch.W, false, 0u, 0u); }
        return;
    }
    for (uint32_t w = (uint32_t)tid; w < (n_chunks + 63u) / 64u; w += 256u) {   // the wrong chunks' bits (and zeros behind them for the next sweep)
        const unsigned long long v = ch.tot[HML_WL_MAP_F + w];
        map[w] = v;
        if (v) ch.tot[HML_WL_MAP_F + w] = 0ull;
    }
    __syncthreads();
    if (tid >= 64) return;
    const bool self = mdl->self_trans != 0;
    const bool act = lane < K;
    float acol[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) acol[i] = (act && i < K) ? mdl->A[i * K + lane] : 0.0f;
    const uint32_t W = hml_chunk_warmup(mdl, ch.W);   // (the warm-up this sweep's chunks ran with: adapted at the very end)
    unsigned long long total_nfb = ch.tot[1], redone = 0ull;
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64u) {
        while (true) {   // wave-uniform
            const unsigned long long todo = map[c0 >> 6];
            if (todo == 0ull) break;
            const uint32_t c = c0 + (uint32_t)(__ffsll((long long)todo) - 1);
            hml_compat_flag_clear(map, c, lane);
            for (uint32_t cc = c; cc < n_chunks; ++cc) {   // chunks c, c + 1, ... until one leaves what its successor started from
                const uint32_t lo = cc << lay.lshift, hi = ((uint64_t)lo + L < (uint64_t)B) ? lo + L : B;
                float prev = act ? ch.exitv[(uint64_t)(cc - 1u) * K + lane] : 0.0f;
                uint32_t nfb = 0u;
                const uint32_t old_nfb = ch.nfb[cc];
                hml_wl_forward_again<KS>(K, B, self, lay, em, g, rows, acol, lo, hi, prev, nfb, lane);
                if (act) ch.exitv[(uint64_t)cc * K + lane] = prev;
                if (lane == 0) ch.nfb[cc] = nfb;
                total_nfb += (unsigned long long)nfb - (unsigned long long)old_nfb;
                redone++;
                if (cc + 1u >= n_chunks) break;
                hml_compat_flag_clear(map, cc + 1u, lane);   // (the successor is compared right here)
                const float nx = act ? ch.entry[(uint64_t)(cc + 1u) * K + lane] : 0.0f;
                const bool exact_next = ((uint64_t)(cc + 1u) << lay.lshift) <= W;
                if (exact_next || __ballot(act && hml_f2u(nx) != hml_f2u(prev)) == 0ull) break;
            }
        }
    }
    if (lane == 0) {
        mdl->uniform_fallbacks += total_nfb; mdl->forward_refits += redone;
        if (redone != 0ull) hml_wl_fwd_adapt(mdl, ch.W, true, W, (2u * W < 1024u) ? 2u * W : 1024u);
        else if (mdl->wl_retry == 0u) hml_wl_fwd_adapt(mdl, ch.W, false, 0u, 0u);
    }
}

// hml_categorical's literal form over a lane's own weights (hml_compat_categorical_exact, a chunk a lane)
template <int KC>
__device__ __forceinline__ int hml_wl_categorical_exact(const float (&w)[KC], const int K, const double u) {
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < KC; ++i) if (i < KC - 3 || i < K) sum += (double)w[i];
    double cp = 0.0;
    int res = K - 1;
    bool found = false;
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        if (i < KC - 3 || i < K) {
            cp += (double)w[i] / sum;
            const double c = (i == K - 1) ? 1.0 : cp;
            if (!found && !(c < u)) { res = i; found = true; }
        }
    }
    return res;
}

// the warm-up of the backward draws' chunks (rows): the model's own unless the caller fixed one - it starts at 64, doubles when a sweep
// had chunks that ran again, falls by half after four sweeps without one while above 16 and by a quarter after sixteen below, floor 8
__device__ __forceinline__ uint32_t hml_wl_bwd_warmup(const hml_model* mdl, const uint32_t W) { return W == HML_CHUNK_W_ADAPTIVE ? mdl->wl_bwd_W : W; }
__device__ __forceinline__ void hml_wl_bwd_warmup_adapt(hml_model* mdl, const uint32_t W, const unsigned long long ran_again) {
    if (W != HML_CHUNK_W_ADAPTIVE) return;
    uint32_t w = mdl->wl_bwd_W;
    if (ran_again != 0ull) { w = (2u * w < 1024u) ? 2u * w : 1024u; mdl->wl_bwd_quiet = 0u; }
    else {
        const uint32_t quiet = ++mdl->wl_bwd_quiet;
        if (w > 16u && quiet >= 4u) { w = w / 2u; mdl->wl_bwd_quiet = 0u; }
        else if (quiet >= 16u) {
            const uint32_t lower = w - w / 4u;
            w = lower > 8u ? lower : 8u;
            mdl->wl_bwd_quiet = 0u;
        }
    }
    mdl->wl_bwd_W = w;
}

// The backward draws, a chunk a lane: rows (c + 1) L .. c L + 1 of the trellis (blocks b = t - 1 from the chunk's last down),
// from state 0 W rows above (or from the trellis's last row, whose weights do not depend on a state above).
template <int KC>
HML_KERNEL __launch_bounds__(64) void hml_k_wl_backward(hml_model* __restrict__ mdl, const float* __restrict__ rows, int16_t* __restrict__ q,
                                                        const hml_compat_chunks ch) {
    constexpr int PITCH = KC + 4;
    __shared__ __attribute__((aligned(16))) float sAT[KC * PITCH];   // [q][i] = A(i, q)
    if (mdl->halted != 0u) return;
    const int lane = threadIdx.x;
    const int K = mdl->K;
    const uint32_t B = mdl->B;
    const hml_layout lay = hml_wl_layout(mdl);
    const uint32_t L = 1u << lay.lshift, C = (uint32_t)(((uint64_t)B + L - 1u) >> lay.lshift);
    for (int idx = lane; idx < KC * KC; idx += 64) {
        const int qq = idx / KC, i = idx - qq * KC;
        sAT[qq * PITCH + i] = (i < K && qq < K) ? mdl->A[i * K + qq] : 0.0f;
    }
    hml_compat_fence();
    const uint32_t W = hml_wl_bwd_warmup(mdl, ch.W);
    const hml_key key = mdl->key;
    const unsigned long long epoch = mdl->epoch;
    for (uint32_t c0 = blockIdx.x * 64u; c0 < C; c0 += gridDim.x * 64u) {   // wave-uniform
        const uint32_t c = c0 + (uint32_t)lane;
        const bool valid = c < C;
        const int64_t lo = (int64_t)c << lay.lshift;
        const int64_t top = lo + (int64_t)L - 1 + (int64_t)W;   // the block the lane's chain of draws starts at (if inside the trace)
        const bool from_last = top >= (int64_t)B - 1;            // ... or at the trellis's last row: the true start
        auto addr = [&](int64_t b) -> uint64_t {
            const uint32_t bc = b < 0 ? 0u : (b >= (int64_t)B ? B - 1u : (uint32_t)b);
            return hml_wl_at(lay.lshift, K, bc, 0);
        };
        int64_t b = top;
        int j = 0, in_rec = 0;
        bool clean = true;
        float r[KC];
        {
            const uint64_t a = addr(b);
#pragma unroll
            for (int i = 0; i < KC; ++i) r[i] = (i < KC - 3 || i < K) ? rows[a + (uint64_t)i * 64u] : 0.0f;
        }
        const uint32_t steps = W + L;
        for (uint32_t s = 0; s < steps; ++s, --b) {
            const bool active = valid && b <= (int64_t)B - 1 && b >= lo;
            const bool own = s >= W;   // (wave-uniform)
            // the next row travels during the draw
            float rn[KC];
            {
                const uint64_t a = addr(b - 1);
#pragma unroll
                for (int i = 0; i < KC; ++i) rn[i] = (i < KC - 3 || i < K) ? rows[a + (uint64_t)i * 64u] : 0.0f;
            }
            if (s == W) in_rec = j;
            // the row's uniform from its own Philox address (hml_cat_uniform: nothing to read, no kernel ahead of this one)
            const uint32_t bu = b < 0 ? 0u : (b >= (int64_t)B ? B - 1u : (uint32_t)b);
            const double u = hml_cat_uniform(key, epoch, bu + 1u);
            const bool last_row = b >= (int64_t)B - 1;   // (row B: the weights are the row itself, no check - Trellis::sample)
            float w[KC];
            bool neg = false;
            {
                const float4* const at = (const float4*)(sAT + j * PITCH);
#pragma unroll
                for (int i0 = 0; i0 < KC; i0 += 4) {
                    const float4 a4 = at[i0 / 4];
                    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float prod = r[i0 + k] * av[k];
                        w[i0 + k] = last_row ? r[i0 + k] : prod;
                        neg = neg || (w[i0 + k] < 0.0f);
                    }
                }
            }
            if (neg && !last_row) clean = false;   // (ForwardBackward.hpp:147-149: the checking launch runs the chunk again and raises)
            // hml_compat_categorical_wave's screen, the running double sums t_i = (..(w_0 + w_1) + ..) + w_i in registers
            double t[KC];
            t[0] = (double)w[0];
#pragma unroll
            for (int i = 1; i < KC; ++i) t[i] = (i < KC - 3 || i < K) ? t[i - 1] + (double)w[i] : t[i - 1];
            const double sum = t[KC - 1];
            const double us = u * sum, margin = sum * 9.31322574615478515625e-10;   // 2^-30
            bool unclear = false;
            int jn = K - 1;
#pragma unroll
            for (int i = KC - 2; i >= 0; --i) {
                if (i < KC - 4 || i < K - 1) {
                    const double d = t[i] - us;
                    unclear = unclear || !(d > margin || d < -margin);
                    jn = (d > 0.0) ? i : jn;   // !(cp_i < u)
                }
            }
            const bool ok = sum > 0.0 && sum < 1.7976931348623157e308 && !neg;
            if (__builtin_expect(unclear || !ok, 0)) jn = hml_wl_categorical_exact<KC>(w, K, u);
            if (active) {
                j = jn;
                if (own) q[b] = (int16_t)jn;
            }
#pragma unroll
            for (int i = 0; i < KC; ++i) r[i] = rn[i];
        }
        if (valid) {
            ch.out_state[c] = j;
            // (a chunk that met a negative weight is run again by the checking launch, which raises; -2 never equals a state)
            ch.in_state[c] = clean ? (from_last ? -1 : in_rec) : -2;
        }
    }
}

// the chunks of the backward draws from the top: a chunk that started from another state than the chunk above it ended in (or
// met a negative weight) runs again from that state, a state a lane (hml_k_compat_backward_check with this path's chunks)
template <int KS>
HML_KERNEL __launch_bounds__(256) void hml_k_wl_backward_check(hml_model* __restrict__ mdl, const float* __restrict__ rows, int16_t* __restrict__ q,
                                                               const hml_compat_chunks ch) {
    __shared__ unsigned long long map[HML_WL_MAP_WORDS];
    if (mdl->halted != 0u) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = mdl->K;
    const uint32_t B = mdl->B;
    const hml_layout lay = hml_wl_layout(mdl);
    const uint32_t L = 1u << lay.lshift;
    const uint32_t n_chunks = (uint32_t)(((uint64_t)B + L - 1u) >> lay.lshift);
    if (ch.tot[2] == 0ull) {   // every chunk started from the state the chunk above it ended in (hml_k_wl_backward_verify)
        if (tid == 0) hml_wl_bwd_warmup_adapt(mdl, ch.W, 0ull);
        return;
    }
    for (uint32_t w = (uint32_t)tid; w < (n_chunks + 63u) / 64u; w += 256u) {
        const unsigned long long v = ch.tot[HML_WL_MAP_B + w];
        map[w] = v;
        if (v) ch.tot[HML_WL_MAP_B + w] = 0ull;
    }
    __syncthreads();
    if (tid >= 64) return;
    const bool act = lane < K;
    const int sl = act ? lane : 0;
    const hml_key key = mdl->key;
    const unsigned long long epoch = mdl->epoch;
    unsigned long long redone = 0ull;
    for (uint32_t wi = (n_chunks + 63u) / 64u; wi-- > 0u; ) {
        while (true) {   // wave-uniform
            const unsigned long long todo = map[wi];
            if (todo == 0ull) break;
            const uint32_t c = wi * 64u + (uint32_t)(63 - __clzll((long long)todo));
            hml_compat_flag_clear(map, c, lane);
            int j = (c + 1u < n_chunks) ? ch.out_state[c + 1u] : 0;   // (the top chunk starts at the trellis's last row: no state above it)
            for (uint32_t cc = c; ; --cc) {   // chunks c, c - 1, ... until one ends in the state its successor started from
                const uint32_t lo = cc << lay.lshift, hi = ((uint64_t)lo + L < (uint64_t)B) ? lo + L : B;
                for (uint32_t b = hi; b-- > lo; ) {
                    const uint32_t t = b + 1u;
                    const float row = rows[hml_wl_at(lay.lshift, K, b, sl)];
                    const double u = hml_cat_uniform(key, epoch, t);
                    float w;
                    if (t == B) w = act ? row : 0.0f;
                    else {
                        w = act ? row * mdl->A[sl * K + j] : 0.0f;
                        const unsigned long long neg = __ballot(w < 0.0f);
                        if (neg != 0ull) {   // ForwardBackward.hpp:147-149 (the first negative weight in state order is the one reported)
                            const int first = __ffsll((long long)neg) - 1;
                            if (lane == first) hml_raise(mdl, HML_DEVERR_NEG_BACKWARD, w);
                            hml_compat_fence();
                            if (w < 0.0f && lane != first) hml_raise(mdl, HML_DEVERR_NEG_BACKWARD, w);
                        }
                    }
                    j = hml_compat_categorical_wave<KS, true>(w, K, u, lane);
                    if (lane == 0) q[b] = (int16_t)j;
                }
                hml_compat_fence();
                if (lane == 0) ch.out_state[cc] = j;
                hml_compat_fence();
                redone++;
                if (cc == 0u) break;
                hml_compat_flag_clear(map, cc - 1u, lane);   // (the successor is compared right here)
                const int in = ch.in_state[cc - 1u];
                if (in == -1 || in == j) break;
            }
        }
    }
    if (lane == 0) { mdl->forward_refits += redone; hml_wl_bwd_warmup_adapt(mdl, ch.W, redone); }   // (the statistic counts chunks of either pass that ran again)
}

#endif
#endif
