// Device-resident state of one Gibbs chain and the fixed geometry shared by the kernels.
#ifndef HML_STATE_H
#define HML_STATE_H

#include "hml_common.h"
#include "hml_philox.h"

// ---- fixed geometry (the CPU checker mirrors these numbers) ----
// longest warm-up of the fused trellis path's first pass (hml_k_trellis.h; many states forget slowly: K = 10 on uncompressed
// input needs more than the 64 rows of rounds 2-3)
#define HML_TRE_HALO_MAX 128
#define HML_SPAN 4096
#define HML_GROUP_SPANS 16   // spans per group of the two-level block offset (one scatter workgroup)          // positions scanned by one wavefront in blocks_compact
#define HML_REDUCE_CHUNK 256   // blocks per reduction chunk (one workgroup)
#define HML_REDUCE_GROUPS 1024 // chunk c is accumulated by group c % HML_REDUCE_GROUPS
#define HML_BWD_CHUNK 64       // trellis rows per backward map chunk (one wavefront)
#define HML_FWD_GROUP 16       // lanes cooperating on one forward chunk (>= HML_MAX_K)
// Layout of the per-block K-vectors (emission terms, rescale factors, trellis rows): "chunk-transposed".
// The forward kernel gives chunk c = b / L to one lane, so element (block b, state s) lives at
//   ((b % L) * K + s) * cstride + b / L          (L a power of two, cstride >= number of chunks)
// and the 64 lanes of a wavefront read 64 consecutive floats at every step.
struct hml_layout {
    uint32_t lshift;     // log2(L)
    uint32_t cstride;    // floats between consecutive (row-in-chunk, state) planes
};
HML_HD uint64_t hml_bk(const hml_layout lay, uint32_t b, int K, int s) {
    const uint32_t r = b & ((1u << lay.lshift) - 1u), c = b >> lay.lshift;
    return ((uint64_t)r * (uint32_t)K + (uint32_t)s) * lay.cstride + c;
}

#define HML_CNT_SPLIT 16       // the integer count accumulators are split 16 ways to spread atomic contention

// error codes raised on the device (first one wins); mirrored into messages by the host
enum {
    HML_DEVERR_NONE = 0,
    HML_DEVERR_IP_NOT_FINITE = 1,     // EFD.hpp:28-30
    HML_DEVERR_NEG_BACKWARD = 2,      // ForwardBackward.hpp:147-149
    HML_DEVERR_NEG_SUMSQ = 3,         // Conjugate.hpp:137-139
    HML_DEVERR_NIG_ALPHA = 4,         // Observation.hpp:374-385
    HML_DEVERR_NIG_BETA = 5,
    HML_DEVERR_NIG_NU = 6,
    HML_DEVERR_NIG_MU0 = 7,
    HML_DEVERR_MEAN_NOT_FINITE = 8,   // Observation.hpp:148-160
    HML_DEVERR_VAR_NOT_FINITE = 9,
    HML_DEVERR_VAR_NOT_POSITIVE = 10,
    HML_DEVERR_TOO_MANY_RECORDS = 11,
    HML_DEVERR_LAUNCH_GEOMETRY = 12   // a kernel that relies on one wavefront per workgroup was launched with another shape (a host bug)
};

#define HML_MAX_D 4            // data dimensions (K = P^D <= 16 with P >= 2; up to 64 in the reference-compatible mode)

struct hml_model {
    // ---- configuration ----
    int32_t K;
    // multivariate / shared parameters ("-s C P D", reference src/Mapping.hpp:53-137): K = P^D states over D interleaved
    // data dimensions; state s uses parameter map[s][d] = (s / P^d) % P for dimension d.  D = 1: P = K, map[s][0] = s.
    int32_t D, P;
    uint8_t map[HML_CAP_K][HML_MAX_D];
    float logNs[HML_CAP_K];      // theta.logNormalizer(state): sum over the state's parameters (Theta.hpp:148-158)
    uint64_t stat_stride;        // elements between the per-dimension planes of the integral array / block statistics
    int32_t self_trans;
    int32_t dynamic;
    uint32_t T;
    float nig_prior[4];          // alpha, beta, mu0, nu (same tuple for every state, main.cpp:348-352)
    float a_off, a_diag, pi_alpha;
    hml_key key;
    // ---- current parameters ----
    float mu[HML_CAP_K], var[HML_CAP_K], sd[HML_CAP_K];
    double rvar2[HML_CAP_K];     // 1 / (2 var): the emission kernels multiply by it instead of dividing (hml_inner_product)
    float logN[HML_CAP_K];       // theta.logNormalizer(s)          (EFD.hpp:35-38)
    float logA[HML_CAP_K];       // log A(s,s)                       (ForwardBackward.hpp:47-52)
    float A[HML_CAP_K * HML_CAP_K];   // row-major, stride K
    float pi[HML_CAP_K];
    float thr;                   // current wavelet threshold
    float thr_theta;             // threshold implied by the current theta (createBlocks(theta))
    // ---- posteriors (reset to the priors after every draw) ----
    float nig_post[HML_CAP_K][4];
    float dirA[HML_CAP_K * HML_CAP_K];
    float dirPi[HML_CAP_K];
    // ---- block structure ----
    uint32_t B;                  // number of blocks
    uint32_t n_spans;
    // Block capacity of the chain's per-block buffers (starts, statistics, emission terms, trellis, maps, states): T for an
    // ordinary context (the worst case: every position a block), less for a context that was given less (option "max_blocks";
    // chains attached to another chain's observations by default: eight chains of 10^8 positions would otherwise reserve 80 GB
    // for 2 10^5 blocks each).  An enumeration that finds MORE blocks writes nothing beyond the capacity, leaves B = 0 - every
    // kernel of the sweep then finds nothing to do - and HALTS the chain: `halted` = the number of blocks it found; the
    // parameter and recording kernels return at once while it is set, so the chain's state stays that of the last completed
    // sweep.  The host grows the buffers and runs the missing sweeps again (hml_capi.hip: hml_settle) - same results.
    uint32_t cap;
    uint32_t halted;
    // ---- per-sweep accumulators (zeroed by the parameter kernel) ----
    unsigned long long trans[HML_CNT_SPLIT][HML_MAX_K * HML_MAX_K];   // (default path only)
    unsigned long long occ[HML_CNT_SPLIT][HML_MAX_K];
    // copies of the last sweep's sufficient statistics (probe)
    unsigned long long last_trans[HML_CAP_K * HML_CAP_K];
    unsigned long long last_occ[HML_CAP_K];
    float last_sum[HML_CAP_K], last_sumsq[HML_CAP_K];
    // ---- counters ----
    unsigned long long epoch;
    unsigned long long sweeps, block_updates, uniform_fallbacks, forward_refits, forward_serial;
    unsigned long long fused_fallbacks;   // words of the fused block kernel that a waiting workgroup had to compute itself
    unsigned long long n_recorded;
    int32_t max_state_recorded;
    // ---- errors ----
    uint32_t err_code;
    float err_value;
    unsigned long long err_count;
    // ---- scratch for the forward fix-up ----
    uint32_t fwd_mismatch;       // set by a verification round that found a stale chunk
    uint32_t fwd_mismatch2;      // the fused trellis path alternates between two lists of stale chunks (refit rounds)
    // adaptive warm-up length of the speculative forward pass (results never depend on it)
    uint32_t fwd_W, fwd_W0, fwd_serial_ran, fwd_quiet;
    uint32_t fwd_W_burnin;       // floor during the first fwd_burnin_sweeps sweeps of a chain (parameters still far from settled)
    uint32_t fwd_burnin_sweeps;  // (HML_FWD_BURNIN_SWEEPS)
    uint32_t fwd_quiet_need;     // sweeps without a single refit before the warm-up shrinks by a quarter (HML_FWD_QUIET)
    uint32_t tre_fused;          // weakly compressed FB sweeps take the fused trellis kernels (hml_k_trellis.h): stale chunks are
                                 // refitted in parallel there, so the warm-up follows a different rule (hml_k_params)
    uint32_t tre_hi_shift, tre_lo_shift;   // ... the warm-up grows above B >> hi refits per sweep and shrinks below B >> lo
    uint32_t tre_W_floor, tre_floor_age;   // ... and does not shrink below the length that last let the refits explode (forgotten slowly)
    unsigned long long fwd_refits_seen, fwd_serial_seen;
    uint32_t params_ticket;      // arrivals of the parameter kernel's workgroups (hml_k_params.h: the last of every 16 goes on)
    unsigned long long dbg_t[12];   // wall_clock64 stamps of the parameter kernel's stages (printed by hml_sync with HML_PARAMS_DEBUG)
    // more than 16 states, a chunk a lane (hml_k_wide_lanes.h): this sweep's chunk length (log2) and the chunk-transposed arrays' stride
    uint32_t wl_lshift, wl_cstride;
    // ... the warm-up of its backward draws' chunks, adapted on its own (chains of draws from different states coalesce within a few
    // rows, long before a filter forgets its start): rows, sweeps without a chunk that ran again
    uint32_t wl_bwd_W, wl_bwd_quiet;
    uint32_t wl_W_need, wl_need_age;   // ... a warm-up that failed on a settled chain, and the sweeps since (the adaptation stays above twice that for a while)
    uint32_t wl_retry;           // ... this sweep's filter runs once more with a longer warm-up (hml_k_wl_retry_decide): the warm-up, or 0
};

#if defined(__HIPCC__)
// the enumeration found `found` > cap blocks (host_words: the context's host-mapped words [block count, fused-kernel trouble, halted])
__device__ __forceinline__ void hml_halt(hml_model* mdl, uint32_t found, uint32_t* host_words) {
    mdl->B = 0u;
    mdl->halted = found;
    if (host_words) {
        __hip_atomic_store(host_words, found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);       // (the grids and buffers the next attempt needs)
        __hip_atomic_store(host_words + 2, found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// A young chain over millions of blocks (weakly compressed input: each block carries little evidence, the filter forgets
// slowly) starts its forward pass with four times the usual warm-up instead of finding that level through repairs:
// on C5 the first ten sweeps cost 203 ms each (3.6e5 refits) until the adaptation had raised W from 24 to 96.
__device__ __forceinline__ void hml_warmup_for_many_blocks(hml_model* mdl, uint32_t B) {
    if (B >= (1u << 22) && !mdl->tre_fused && mdl->sweeps < 4ull && mdl->fwd_W < 4u * mdl->fwd_W_burnin) mdl->fwd_W = 4u * mdl->fwd_W_burnin;
    // the fused trellis path starts a young chain at its longest first-pass warm-up and lets the refit count walk it down
    // (hml_k_params): starting low would put a million chunks through the refit rounds in the first sweeps
    if (B >= (1u << 22) && mdl->tre_fused && mdl->sweeps < 4ull && mdl->fwd_W < 64u) mdl->fwd_W = 64u;
}
#endif

#endif
