// Shared by the two kinds of translation unit of libhammlet_hip.so's chain code (hammlet_amd/build.py):
//   hml_capi.hip  - the C ABI (include/hml.h) and everything that does not depend on the number of states K: one object;
//   hml_sweep.hip - the sweep for K states: the kernels templated on K and the host code that launches them, behind a table
//                   of function pointers (hml_ktab): fifteen objects, -DHML_TU_K=2 ... 16, compiled in parallel.
// Every object carries its own code object, and the HIP runtime loads a code object when the first kernel of it is launched:
// a run with K states loads the core's (construction, block scan, marginals) and the one of its K (1-2 MB each) instead of
// one 19 MB object with every kernel for 2 ... 16 states - 45 ms of a short run's start-up - and the library
// builds in a fifth of the time.  Kernels that are not templates have internal linkage (HML_KERNEL) so that the objects may
// each hold the ones they launch.  (Round 4: this header and the two files were one file sliced by the preprocessor.)
// Development builds for one K (tools/dev_build.py): both files with -DHML_ONLY_K=k.
#ifndef HML_CAPI_SHARED_HPP
#define HML_CAPI_SHARED_HPP

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/hml.h"
#include "hml_k_backward.h"
#include "hml_k_blocks.h"
#include "hml_k_blocks_fused.h"
#include "hml_k_blocks_fused_many.h"
#include "hml_k_build.h"
#include "hml_k_forward.h"
#include "hml_k_marginals.h"
#include "hml_k_segment.h"
#include "hml_k_trellis.h"
#include "hml_k_trellis_rows.h"
#include "hml_k_compat.h"
#include "hml_k_wide.h"
#include "hml_k_wide_lanes.h"
#include "hml_k_blocks_split_many.h"
#include "hml_k_many.h"
#include "hml_k_params.h"
#include "hml_state.h"
#include "hml_synth_host.hpp"

#include "hml_host_common.hpp"
#include "hml_ctx.hpp"

int hml_set_err(int code, const std::string& msg);   // hml_capi.hip (the thread's last error message)
static int set_err(int code, const std::string& msg) { return hml_set_err(code, msg); }


// what the core calls of the K-dependent part: one table per number of states (defined at the end of hml_sweep.hip)
struct hml_ktab {
    int (*sweep)(hml_ctx* c, char method, bool record);
    int (*iterate_many)(hml_ctx* const* cs, int n, uint64_t first, uint64_t iterations, uint64_t thinning, uint64_t* done);
    void (*params)(hml_ctx* c, int mode);        // hml_k_params<K>: 1 = draw from the priors, 2 = Theta's constructor draw
    void (*derive)(hml_ctx* c);                  // hml_k_derive<K>
};

// Live contexts per device.  The fused block kernel hands block offsets from workgroup to workgroup inside one launch
// (a workgroup spins on the words of lower-numbered ones); that is safe while all lower-numbered workgroups are resident
// or finished, which in-order dispatch guarantees for ONE kernel on the GPU.  With two chains sweeping the same GPU at
// once the eight XCDs can fill up with the late workgroups of one launch and the early ones of the other, each waiting
// for workgroups that cannot be dispatched (observed once under the profiler: two launches stalled for 27 s until the
// firmware's time slicing untangled them).  So while more than one context is alive on a device every sweep takes the
// scan + scatter pair, which has no such hand-off.
extern std::atomic<int> hml_live_ctx[64];   // hml_capi.hip
#define g_live_ctx hml_live_ctx
static bool shares_device(const hml_ctx* c) { return c->device < 64 && g_live_ctx[c->device].load() > 1; }


// ------------------------------------------------------------------------------------------------
static int ctx_bind(hml_ctx* c) {
    HIPCHK(hipSetDevice(c->device));
    return 0;
}

static hipEvent_t ev_get(hml_ctx* c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

struct ProfScope {
    hml_ctx* c; const char* name; hipEvent_t a = nullptr, b = nullptr;
    bool on;
    ProfScope(hml_ctx* c_, const char* n, int level = 2) : c(c_), name(n), on(c_->profiling >= level) {
        // level 1 (the bench's timed region): bracket every 32nd launch only - two event records cost ~6 us of
        // stream time, a visible share of an 80 us sweep
        // (a counter per family: the weakly compressed sweep has two level-1 families, block scan and first trellis pass)
        if (on && c->profiling == 1 && (c->prof[name].tick++ & 31u) != 0u) on = false;
        if (on) { a = ev_get(c); b = ev_get(c); hipEventRecord(a, c->stream); }
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(b, c->stream);
        c->prof[name].pending.push_back({a, b});
        if (c->profiling == 1) {
            // an empty bracket right behind: what two event records measure with nothing in between ("event_null")
            hipEvent_t n0 = ev_get(c), n1 = ev_get(c);
            hipEventRecord(n0, c->stream);
            hipEventRecord(n1, c->stream);
            c->prof["event_null"].pending.push_back({n0, n1});
        }
    }
};

// device scratch that is released on every path out of a function
struct DevScratch {
    void* p = nullptr;
    ~DevScratch() { if (p) hipFree(p); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

static int grid_for(uint64_t items, int per_block, int lo, int hi) {
    uint64_t g = (items + per_block - 1) / per_block;
    if (g < (uint64_t)lo) g = lo;
    if (g > (uint64_t)hi) g = hi;
    return (int)g;
}

static const char* deverr_text(uint32_t code, float v, char* buf, size_t n) {
    switch (code) {
        case HML_DEVERR_IP_NOT_FINITE: snprintf(buf, n, "Result of Normal inner product is not finite!"); break;
        case HML_DEVERR_NEG_BACKWARD: snprintf(buf, n, "Negative backward variable!"); break;
        case HML_DEVERR_NEG_SUMSQ: snprintf(buf, n, "Sum of squares is negative (%s)!", std::to_string(v).c_str()); break;
        case HML_DEVERR_NIG_ALPHA: snprintf(buf, n, "Alpha (%s) must be positive!", std::to_string(v).c_str()); break;
        case HML_DEVERR_NIG_BETA: snprintf(buf, n, "Beta (%s) must be positive!", std::to_string(v).c_str()); break;
        case HML_DEVERR_NIG_NU: snprintf(buf, n, "Nu (%s)must be positive!", std::to_string(v).c_str()); break;
        case HML_DEVERR_NIG_MU0: snprintf(buf, n, "Mu0 (%s)  must be finite!", std::to_string(v).c_str()); break;
        case HML_DEVERR_MEAN_NOT_FINITE: snprintf(buf, n, "Mean (%s) must be set to a finite value!", std::to_string(v).c_str()); break;
        case HML_DEVERR_VAR_NOT_FINITE: snprintf(buf, n, "Variance(%s) must be set to a finite value!", std::to_string(v).c_str()); break;
        case HML_DEVERR_VAR_NOT_POSITIVE: snprintf(buf, n, "Variance (%s) must be positive!", std::to_string(v).c_str()); break;
        case HML_DEVERR_TOO_MANY_RECORDS: snprintf(buf, n, "Too many recorded iterations for the marginal counters!"); break;
        case HML_DEVERR_LAUNCH_GEOMETRY: snprintf(buf, n, "internal error: a one-wavefront kernel was launched with %d threads per workgroup", (int)v); break;
        default: snprintf(buf, n, "device error %u", code);
    }
    return buf;
}

static int fetch_model(hml_ctx* c, hml_model* out) {
    HIPCHK(hipMemcpyAsync(out, c->d_mdl, sizeof(hml_model), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

static int check_device_error(hml_ctx* c) {
    // (the two error words only: the model is 100 KB since its arrays hold HML_CAP_K states)
    struct { uint32_t code; float value; } e;
    static_assert(offsetof(hml_model, err_value) == offsetof(hml_model, err_code) + sizeof(uint32_t), "err_code and err_value are read together");
    HIPCHK(hipMemcpyAsync(&e, &c->d_mdl->err_code, sizeof e, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (e.code != 0) {
        char buf[256];
        return set_err(HML_ERR_MODEL, deverr_text(e.code, e.value, buf, sizeof buf));
    }
    return 0;
}



// ---------------------------------------------------------------------------------------- blocks
// K4: scan (the HBM-bound kernel) + scatter with in-kernel offsets
static void launch_compact_pair(hml_ctx* c, int mode, float thr) {
    // mode 0: model threshold; 1: explicit threshold
    const uint32_t n_groups = (c->n_spans + HML_GROUP_SPANS - 1) / HML_GROUP_SPANS;
    // the summary scan skips unopened groups; when most groups would be opened (weak compression) the plain
    // float stream is the better access pattern - both give the same blocks
    const bool dense = !c->summary_always && c->B_hint && (uint64_t)c->B_hint * 24u > c->T;
    // ... and with B ~ T the scan stages the flags themselves (512 bytes per span) instead of 16-bit offsets (2 bytes per block)
    const bool bits = dense && c->stage_bits;
    {
        ProfScope ps(c, "blocks_compact", 1);
        if (c->use_keys && !dense) {
            hipLaunchKernelGGL(hml_k_compact_scan_summary, dim3(n_groups), dim3(256), 0, c->stream, c->d_summary, c->d_w,
                               (uint32_t)c->T, c->d_mdl, thr, mode, c->key_base, c->d_stage, c->d_span_count, c->d_coarse1);
        } else if (bits) {
            hipLaunchKernelGGL(hml_k_compact_scan_bits, dim3((c->n_spans + 3) / 4), dim3(256), 0, c->stream, c->d_w, (uint32_t)c->T,
                               c->d_mdl, thr, mode, (unsigned long long*)c->d_stage, c->d_span_count);
        } else {
            hipLaunchKernelGGL(hml_k_compact_scan, dim3((c->n_spans + 3) / 4), dim3(256), 0, c->stream, c->d_w, (uint32_t)c->T,
                               c->d_mdl, thr, mode, c->d_stage, c->d_span_count);
        }
    }
    {
        ProfScope ps(c, "blocks_scatter");
        if (!(c->use_keys && !dense))
            hipLaunchKernelGGL(hml_k_group_totals, dim3((n_groups + 255) / 256), dim3(256), 0, c->stream, c->d_span_count,
                               c->n_spans, c->d_coarse1);
        if (bits)
            hipLaunchKernelGGL(hml_k_compact_scatter_bits, dim3(n_groups), dim3(256), 0, c->stream, (const unsigned long long*)c->d_stage,
                               c->d_span_count, c->d_coarse1, c->n_spans, (uint32_t)c->T, c->d_mdl, c->d_starts, c->d_hB);
        else
        hipLaunchKernelGGL(hml_k_compact_scatter, dim3(n_groups), dim3(256), 0, c->stream, c->d_stage, c->d_span_count,
                           c->d_coarse1, c->n_spans, (uint32_t)c->T, c->d_mdl, c->d_starts, c->d_hB);
    }
}

static int launch_compact(hml_ctx* c, bool use_override, float thr) {
    launch_compact_pair(c, use_override ? 1 : 0, thr);
    KLAUNCH_CHECK();
    {
        ProfScope ps(c, "block_stats");
        const uint32_t hint = c->B_hint ? c->B_hint : (uint32_t)std::min<uint64_t>(c->T, 1u << 20);
        for (int d = 0; d < c->D; ++d)   // the same enumeration for every dimension (dimension-major planes)
            hipLaunchKernelGGL(hml_k_block_stats, dim3(grid_for(hint, 256, 64, 16384)), dim3(256), 0, c->stream,
                               c->d_ia + (uint64_t)d * (c->T + 1), c->d_starts, c->d_mdl, c->d_bstat + (uint64_t)d * c->T);
    }
    KLAUNCH_CHECK();
    return 0;
}

// capacity-limited contexts (hml_ctx.hpp): every enqueued sweep is noted, so that sweeps a halted chain skipped can be run again
static inline bool cap_limited(const hml_ctx* c) { return c->cap != 0 && c->cap < c->T; }
static inline bool chain_halted(const hml_ctx* c) { return cap_limited(c) && ((volatile uint32_t*)c->h_B)[2] != 0u; }
static inline void log_sweep(hml_ctx* c, char method, bool record) {
    c->requested++;
    if (cap_limited(c)) c->sweep_log.push_back((uint8_t)((method == HML_METHOD_MIXTURE ? 1 : 0) | (record ? 2 : 0)));
}
static inline int settle_if_limited(hml_ctx* c) { return cap_limited(c) ? hml_settle(c) : 0; }

static void refresh_hint(hml_ctx* c) {
    const uint32_t b = *(volatile uint32_t*)c->h_B;
    if (b) c->B_hint = b + b / 4 + 1024;
}


static int ensure_marginal_buffers(hml_ctx* c) {
    if (c->d_diff) return 0;
    const uint64_t n = (uint64_t)c->K * (c->T + 1);
    HIPCHK(hipMalloc(&c->d_diff, n * sizeof(int32_t)));
    HIPCHK(hipMemsetAsync(c->d_diff, 0, n * sizeof(int32_t), c->stream));
    const uint64_t words = (c->T + 1 + 31) / 32 + 1;
    HIPCHK(hipMalloc(&c->d_boundary, words * sizeof(uint32_t)));
    HIPCHK(hipMemsetAsync(c->d_boundary, 0, words * sizeof(uint32_t), c->stream));
    return 0;
}




// ---- chunk length of the fused trellis path (hml_ctx.hpp: tre_autotune)
#define HML_TRE_TUNE_AFTER 48u   // sweeps before the measurement: the filter's warm-up length has settled by then
// Candidates.  The wavefronts of the first pass (64 chunks each) all do the same work and stay resident from launch to
// exit, so the pass takes a whole number of ROUNDS over the machine's wavefront slots: the chunk lengths worth measuring
// are those that fill 1, 2, 3 ... rounds almost completely (longer chunks = a smaller warm-up share, but longer refits of
// the chunks that fail verification).  hml_k_trellis_tile (HML_TRELLIS_ROWS=0) keeps round 2's list.
static uint32_t tre_default_L_old(uint32_t hint) { return hint >= (1u << 26) ? 128u : hint >= (1u << 24) ? 64u : (uint32_t)HML_TRE_MIN_L; }
static int tre_candidates(const hml_ctx* c, uint32_t hint, uint32_t* out) {
    int n = 0;
    if (!c->tre_rows || c->tre_slots <= 0) {
        const uint32_t L0 = tre_default_L_old(hint);
        for (uint32_t q = 4; q <= 8; ++q) {   // L0 * {1, 1.25, 1.5, 1.75, 2}, multiples of 32
            const uint32_t l = L0 * q / 4u;
            if (l % 32u == 0u && l <= 256u) out[n++] = l;
        }
        return n;
    }
    // (the hint is the last sweep's block count with a quarter of headroom - what the grids are sized for; the rounds are
    // counted over the blocks themselves)
    const uint32_t blocks = hint > 1024u ? (uint32_t)(((uint64_t)hint - 1024u) * 4u / 5u) : hint;
    for (uint32_t rounds = 1; rounds <= 8u && n < 6; ++rounds) {
        const double waves = 0.985 * (double)c->tre_slots * rounds;              // (a little air: one wavefront too many costs a round)
        uint32_t l = (uint32_t)((double)blocks / (64.0 * waves)) + 1u;
        l = (l + 31u) / 32u * 32u;
        if (l < (uint32_t)HML_TRE_MIN_L) l = HML_TRE_MIN_L;
        // (without checkpoints a refit walks its whole chunk: beyond 512 rows that costs more than the warm-up saves)
        if (l > (c->tre_ckpt ? (uint32_t)HML_TRE_MAX_L : 512u)) continue;
        bool seen = false;
        for (int i = 0; i < n; ++i) seen = seen || out[i] == l;
        if (!seen) out[n++] = l;
        if (l == (uint32_t)HML_TRE_MIN_L) break;
    }
    if (n == 0) out[n++] = 512u;
    return n;
}
static uint32_t tre_default_L(const hml_ctx* c, uint32_t hint) {   // until the measurement: two rounds where there are blocks for them
    if (!c->tre_rows || c->tre_slots <= 0) return tre_default_L_old(hint);
    uint32_t cand[8];
    const int n = tre_candidates(c, hint, cand);
    return cand[n > 1 ? 1 : 0];
}
static bool tre_tuned_for(const hml_ctx* c, uint32_t hint) {
    return c->tre_tuned_L && hint <= c->tre_tuned_hint + c->tre_tuned_hint / 8u && hint + hint / 8u >= c->tre_tuned_hint;
}
// does the next fused-trellis sweep measure a candidate?  (it then runs outside any graph and waits for its own events)
static bool tre_wants_measurement(const hml_ctx* c, uint32_t hint) {
    return !c->tre_L && c->tre_autotune && !tre_tuned_for(c, hint) && c->tre_dense_sweeps >= HML_TRE_TUNE_AFTER;
}
// the chunk length of the next sweep; *measure: bracket the trellis kernels with events and report (tre_tune_report)
static uint32_t tre_pick_L(hml_ctx* c, uint32_t hint, bool capturing, bool* measure) {
    *measure = false;
    if (c->tre_L) return c->tre_L;
    if (tre_tuned_for(c, hint)) return c->tre_tuned_L;
    if (capturing || !tre_wants_measurement(c, hint)) return c->tre_tuned_L ? c->tre_tuned_L : tre_default_L(c, hint);
    if (c->tre_tune_step < 0) {
        // a measurement starts: its candidates are fixed now (the block count drifts over the 2 n measuring sweeps, and the
        // list with it: timings of different lengths would mix, and the winner might never have been measured)
        c->tre_cand_n = tre_candidates(c, hint, c->tre_cand);
        if (c->tre_cand_n < 2) { c->tre_tuned_L = c->tre_cand[0]; c->tre_tuned_hint = hint; return c->tre_cand[0]; }
        c->tre_tune_step = 0;
        for (float& v : c->tre_tune_ms) v = 3.4e38f;
    }
    *measure = true;
    return c->tre_cand[c->tre_tune_step % c->tre_cand_n];
}
static void tre_tune_report(hml_ctx* c, uint32_t hint, float ms) {
    const uint32_t* const cand = c->tre_cand;
    const int n = c->tre_cand_n;
    const int i = c->tre_tune_step % n;
    c->tre_tune_ms[i] = std::min(c->tre_tune_ms[i], ms);
    if (++c->tre_tune_step < 2 * n) return;
    int best = 0;
    for (int k = 1; k < n; ++k) if (c->tre_tune_ms[k] < c->tre_tune_ms[best]) best = k;
    c->tre_tuned_L = cand[best];
    c->tre_tuned_hint = hint;
    c->tre_tune_step = -1;
    if (getenv("HML_TRELLIS_TUNE_DEBUG")) {
        fprintf(stderr, "[trellis tune] %u blocks:", hint);
        for (int k = 0; k < n; ++k) fprintf(stderr, " L=%u %.3f ms", cand[k], c->tre_tune_ms[k]);
        fprintf(stderr, " -> L=%u\n", cand[best]);
    }
}


// ---- several chains of one device in one set of launches (hml_k_many.h) ----
static bool many_eligible(hml_ctx* const* cs, int n, char method) {
    if (n < 2 || n > 64 || method != HML_METHOD_FB) return false;
    const hml_ctx* a = cs[0];
    for (int i = 0; i < n; ++i) {
        const hml_ctx* c = cs[i];
        if (!c->model_set || c->device != a->device || c->K != a->K || c->T != a->T || c->D != 1 || c->compat || c->wide || !c->dynamic || !c->use_keys ||
            c->probes || c->profiling || c->fwdL != a->fwdL || c->fwdL_many != a->fwdL_many || c->late_rescale != a->late_rescale || c->n_spans != a->n_spans)
            return false;
        for (int j = 0; j < i; ++j) if (cs[j] == c) return false;
    }
    return true;
}
// weakly compressed sweeps (either criterion of sweep_k / launch_compact_pair) keep their own kernels: not batched
static bool many_sparse(hml_ctx* c) {
    refresh_hint(c);
    return !(c->B_hint >= c->dense_min_blocks) && !(!c->summary_always && c->B_hint && (uint64_t)c->B_hint * 24u > c->T);
}


#endif
