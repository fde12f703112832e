// The chain context behind `hml_ctx*` (include/hml.h), shared by the translation units of libhammlet_hip.so
// (hml_capi.hip: the chain; hml_pool.hip: chain-parallel pooling over RCCL).
#ifndef HML_CTX_HPP
#define HML_CTX_HPP

#include <hip/hip_runtime.h>

#include <atomic>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/hml.h"
#include "hml_host_common.hpp"
#include "hml_state.h"

// The read-only construction of one observation trace on one device - breakpoint weights, their group summary, maxlet
// coefficients, integral array(s) - shared by every chain that was attached to it (hml_attach_observations): the chains of a
// run read the same trace, their block sets are nested by threshold, and private copies (0.8 GB of integral array per chain at
// 10^8 positions) only defeat the caches.  Freed with the last context that holds it.
struct hml_trace {
    std::atomic<int> refs{1};
    float* d_w = nullptr;
    uint8_t* d_summary = nullptr;
    float* d_coeff = nullptr;
    float2* d_ia = nullptr;
};

struct ProfAcc { double ms = 0; uint64_t n = 0; uint32_t tick = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; };

struct hml_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint64_t seed = 0;
    uint32_t chain = 0;
    uint64_t T = 0;
    int K = 0;
    int D = 1, P = 0;              // data dimensions / emission parameters ("-s C P D"; P = 0: univariate, P = K)
    bool loaded = false, model_set = false;
    bool dynamic = true;
    bool blocks_valid = false;     // starts/bstat describe the current threshold
    double sigma = 0;
    // construction (owned by `trace`; the pointers below are this context's view of it)
    hml_trace* trace = nullptr;
    float* d_w = nullptr;
    uint8_t* d_summary = nullptr;  // largest key of every 16-position group (what the scan streams)
    int32_t key_base = 0;
    double key_scale = 1.0;         // product of the weight multipliers applied so far
    bool use_keys = true;
    bool fused_blocks = true;       // option fused_blocks: 0 forces the scan + scatter + statistics launches
    uint64_t host_draws = 0;        // hml_categorical_draw calls so far (Philox sub-stream HOST)
    bool summary_always = false;    // option weight_keys = 2: never fall back to the float stream (tests)
    float* d_coeff = nullptr;
    float2* d_ia = nullptr;
    // block structure
    uint16_t* d_stage = nullptr;
    uint32_t *d_span_count = nullptr, *d_starts = nullptr;
    float2* d_bstat = nullptr;
    uint32_t n_spans = 0;
    uint32_t* d_coarse1 = nullptr;   // block count per group of HML_GROUP_SPANS spans
    unsigned long long* d_group_word = nullptr;   // fused block kernel: {generation, starts, last start} per tile
    uint32_t* d_wave_total = nullptr;             // split many-chain block kernels: starts per wavefront of a tile (hml_k_blocks_split_many.h)
    int fused_slots = 0;                          // workgroups of it that are resident at once (occupancy x compute units; 0: not asked yet)
    uint32_t fused_spin_limit = 4096;             // polls of a tile word before the waiting thread computes the word itself
    bool fused_keep = false;                      // option fused_blocks = 2 (tests): keep the kernel after it reported trouble
    unsigned long long* d_dbg = nullptr;
    // hipGraph replay of a non-recording sweep (launch-bound inner loop); re-captured when the grid hint moves
    bool use_graph = false;
    hipGraphExec_t graph_exec = nullptr;
    char graph_method = 0;
    uint32_t graph_hint = 0;
    bool graph_dynamic = false, graph_valid_blocks = false;
    bool graph_fused = false;       // the captured sweep was free to take the fused block kernel
    // sweep buffers (allocated by set_model)
    float *d_em = nullptr, *d_gsc = nullptr, *d_rows = nullptr, *d_eprobe = nullptr, *d_aprobe = nullptr;
    float *d_entry = nullptr, *d_exitA = nullptr;
    uint32_t* d_redo = nullptr;    // backward chunks that failed the forward verification (list for the repair step)
    uint32_t* d_redo2 = nullptr;   // second list of stale chunks and the bitmap of the sequential finisher (fused trellis path)
    uint32_t* d_tre_bitmap = nullptr;
    uint32_t* d_tre_ckpt = nullptr;      // the first pass's forward vectors every 64 rows of a chunk (hml_k_trellis_rows.h): where a refit may stop
    bool late_rescale = true;      // strongly compressed univariate sweeps: no plane of rescale factors (HML_LATE_RESCALE=0: keep it)
    uint32_t tre_L = 0;            // its chunk length (0: chosen from the number of blocks, then by measurement; HML_TRELLIS_L, option "trellis_L")
    // The best chunk length depends on how the wavefronts of hml_k_trellis_tile (64 chunks each, resident for the whole
    // launch) divide among the CUs' slots - a rounding effect no formula of ours predicted - so it is MEASURED: once the
    // warm-up policy has settled, every candidate length runs two sweeps between a pair of events and the fastest
    // stays.  The chain's results do not depend on the chunk length (rows, maps and draws are addressed by block).
    bool tre_autotune = true;      // HML_TRELLIS_TUNE=0: keep the length chosen from the number of blocks
    uint32_t tre_tuned_L = 0, tre_tuned_hint = 0;
    int tre_tune_step = -1;        // >= 0 while measuring: candidate step % n, pass step / n
    float tre_tune_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t tre_cand[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the candidate lengths of the measurement under way (frozen when it starts:
    int tre_cand_n = 0;                               // the list depends on the drifting block count)
    uint64_t tre_dense_sweeps = 0; // fused-trellis sweeps of this chain so far
    uint32_t graph_tre_L = 0;      // chunk length of the captured sweep
    int tre_slots = 0;             // wavefronts of hml_k_trellis_rows the device holds at once (0: not asked yet, -1: unknown)
    bool stage_bits = true;        // weakly compressed sweeps stage block-start FLAGS between scan and scatter (HML_STAGE_BITS=0: 16-bit offsets)
    uint32_t tre_refit_rounds = 2; // parallel refit rounds before the sequential finisher (HML_TRELLIS_REFIT_ROUNDS, 0 ... 6: the rounds tag the chunks they list with 3 bits).
                                   // Four until the end of round 5: on configs 3u and 5 rounds 2-4 never had a chunk to refit in 240 sweeps and cost 9 us each
                                   // (two launches at the 4.6 us floor of a kernel in a stream; profiles/round5_refit_rounds.txt)
    bool tre_ckpt = true;          // refits stop where they meet the first pass's checkpoint again (HML_TRELLIS_CKPT=0: always the whole chunk)
    bool tre_rows = true;          // its first pass is hml_k_trellis_rows (round 3); HML_TRELLIS_ROWS=0: hml_k_trellis_tile (round 2)
    bool tre_fused = true;         // weakly compressed FB sweeps take the fused trellis kernels (HML_TRELLIS_FUSED=0: the separate ones)
    uint32_t* d_touched = nullptr; // backward chunks whose rows the repair recomputed, tagged with the sweep
    uint32_t* d_fb = nullptr;
    unsigned long long *d_smap = nullptr, *d_cmap = nullptr;
    unsigned long long *d_scmap = nullptr, *d_super = nullptr;   // two-level chain (weakly compressed sweeps)
    uint8_t *d_bentry = nullptr, *d_bentry2 = nullptr;
    int16_t* d_q = nullptr;
    double* d_partial = nullptr;
    bool params_spread = false;    // single-chain sweeps: the parameter kernel's tree over 16 workgroups (hml_k_params.h; HML_PARAMS_SPREAD)
    int32_t* d_diff = nullptr;
    uint32_t* d_boundary = nullptr;
    hml_model* d_mdl = nullptr;
    uint32_t* h_B = nullptr;        // pinned + mapped, four words: [0] the block count of the latest enumeration (grid sizing hint),
                                    // [1] set by the fused block kernel when a bounded wait expired, [2] the chain is HALTED: the number of
                                    // blocks an enumeration found beyond the capacity of the per-block buffers (hml_state.h)
    // Block capacity of the per-block buffers (0 until the observations are loaded; T = the worst case, every position a block).
    // Less for a context with option "max_blocks" and, by default, for a context ATTACHED to another one's observations; a sweep
    // that needs more halts the chain on the device, and the host grows the buffers and runs the missing sweeps again
    // (hml_settle) - `sweep_log` holds what was enqueued since the last point at which the stream was known to be idle.
    uint64_t cap = 0;
    uint64_t cap_opt = 0;           // option "max_blocks" / HML_MAX_BLOCKS (0: the default of the context's kind)
    std::vector<uint8_t> sweep_log; // per enqueued sweep: bit 0 mixture, bit 1 recorded
    unsigned long long log_base = 0;   // the model's sweep counter when the log started (= sweeps requested up to then)
    unsigned long long requested = 0;  // sweeps requested of this chain so far (its sweep counter once everything has run)
    unsigned long long call_base = 0;  // `requested` when the current hml_iterate / hml_iterate_many call began: a sweep's index in
                                       // its call (what the recording callback is told) = its ordinal - call_base
    uint64_t grown = 0;             // times the buffers were grown (hml_stats)
    uint32_t* d_hB = nullptr;       // device view of h_B
    uint32_t B_hint = 0;
    bool hint_stale = true;         // the hint predates the current parameters (new model, prior draw, mode switch)
    // forward geometry
    int fwdL = 4, fwdW = 12;   // W is the floor of the adaptive warm-up (measured: 12 beats 16 and 24 on C1-C4; 8 does not)
    // the warm-up policy of hml_k_params (HML_FWD_BURNIN_SWEEPS, HML_FWD_QUIET): the floor of a young chain holds for 64
    // sweeps and the warm-up shrinks by a quarter after 16 sweeps without a refit.  Measured over the first 400 sweeps
    // of configs 2 / 3 / 4 (tools/burnin_sweep.sh): 19.9 / 25.4 / 50.4 ms against 21.0 / 26.1 / 52.2 ms with (256, 32);
    // (32, 8) and faster let the warm-up fall while the parameters still move - bursts of 150+ refits, 26.6 ms on config 3
    uint32_t fwd_burnin_sweeps = 64, fwd_quiet_need = 16;
    int fwdW_init = 24;        // where a chain starts and the floor of its first fwd_burnin_sweeps sweeps: while the parameters are
                               // still far from settled the filter forgets slowly (131 refits in sweeps 20-220 of C3 with a floor of 12)
    hml_layout lay = {2, 0};
    // weakly compressed sweeps (B_hint >= dense_min_blocks): longer forward chunks - the warm-up is a smaller share
    // of the work - in their own chunk-transposed layout; which geometry a sweep uses never changes its results
    int fwdL_dense = 16;
    hml_layout lay_dense = {4, 0};
    // in between (B_hint >= mid_min_blocks = 2^18: 33 000 chunks of 8 still put a wavefront on every second SIMD): chunks of 8 - 32 filter
    // steps per 8 blocks instead of 28 per 4.  Measured on config 3's trace with 8 / 10 / 12 / 16 states in the model (3.0 / 5.2 / 7.2 /
    // 9.7 10^5 blocks per sweep): 0.188 / 0.228 / 0.489 / 0.498 -> 0.165 / 0.224 / 0.429 / 0.437 ms per sweep; with 5 states (1.8 10^5 blocks:
    // below the threshold) chunks of 8 cost 0.063 against 0.060 ms (profiles/round5_wide_path.txt).  HML_FWD_CHUNK_MID, HML_MID_MIN_BLOCKS.
    int fwdL_mid = 8;
    hml_layout lay_mid = {3, 0};
    uint32_t mid_min_blocks = 1u << 18;
    // chains batched by hml_iterate_many are bound by throughput, not latency: chunks of 8 pay the warm-up over twice as many
    // blocks (20 filter steps per 8 blocks instead of 16 per 4; measured with eight chains of config 3: 0.181 against 0.188 ms
    // per round; 16 and 32 are slower again - too few wavefronts).  HML_FWD_CHUNK_MANY.
    int fwdL_many = 8;
    hml_layout lay_many = {3, 0};
    uint32_t dense_min_blocks = 1u << 22;
    bool graph_dense = false, graph_mid = false;
    bool probes = false;
    bool rec_marginals = true;
    hml_record_cb cb = nullptr;
    void* cb_user = nullptr;
    int fm_slots = 0;              // workgroups of the many-chain block kernel that are resident at once (0: not asked yet)
    void* d_many = nullptr;        // hml_iterate_many: the chains' pointers (hml_chain_dev), kept by the first chain of a batch
    int many_cap = 0;
    bool compat = false;           // option "compat": sweeps exactly as the reference computes them (hml_k_compat.h)
    void* d_mt = nullptr;          // its engine (hml_mt_state)
    float* d_crows = nullptr;      // its trellis, (T + 1) x K
    void* d_cchunk = nullptr;      // chunks of its filter / backward draws (hml_compat_chunks, hml_k_compat.h)
    uint32_t* d_cdraws = nullptr;  // the engine's outputs of a sweep's categorical draws
    void* d_clists = nullptr;      // its count pass's lists by state (hml_compat_lists)
    int compat_chunks = 0;         // 0: chosen from the block count; 1: the sequential form (HML_COMPAT_CHUNKS)
    int compat_warmup = 0;         // blocks a chunk runs ahead of its first; 0: 64 (128 beyond 16 states) (HML_COMPAT_WARMUP)
    bool wide = false;             // more than 16 states on the default path: the number of states is a run-time value, a state a lane (hml_k_wide.h)
    void* d_wacc = nullptr;        // its integer counts of a sweep (hml_wide_acc)
    float* d_wA = nullptr;         // ... the transition matrix padded to 64 x 64 (hml_k_wide_lanes.h)
    int wide_lanes = 1;            // ... filter and backward draws with a chunk a lane (0: a state a lane, hml_k_compat.h's kernels) (HML_WIDE_LANES)
    int wide_w0 = 32;              // ... the floor of its chunks' adaptive warm-up, blocks (HML_WIDE_W0)
    uint32_t wide_max_chunks = 0;  // ... the most chunks a sweep is cut into (HML_WIDE_MAX_CHUNKS; 0: HML_WL_MAX_CHUNKS)
    uint64_t wide_chunks_cap = 0;  // ... chunks its per-chunk arrays hold (alloc_sweep_buffers)
    int wide_lshift = -1;          // ... log2 of a forced chunk length (tests: HML_WIDE_L), -1: from the block count
    bool pooled = false;           // the marginals are a pooled payload (hml_pool_install): common labels, counts of several chains
    std::vector<int32_t> pool_perm;   // perm[pooled label] = this chain's label, from the export that preceded the pooling
    int profiling = 0;             // 0 off, 1 dominant kernel only (blocks_compact, every 32nd launch), 2 every kernel family
    uint32_t prof_tick = 0;
    std::map<std::string, ProfAcc> prof;
    std::vector<hipEvent_t> ev_pool;
};

// hml_capi.hip
int hml_ctx_bind(hml_ctx* c);                         // hipSetDevice(ctx's device)
int hml_ctx_fetch_model(hml_ctx* c, hml_model* out);  // synchronising copy of the device-resident model
int hml_ctx_ensure_marginal_buffers(hml_ctx* c);
int hml_settle(hml_ctx* c);                           // stream idle, no halted sweep left behind (block capacity, above)
// marginal segments on the device: starts d_seg[M] and count differences at the starts d_g[M * K] (caller frees both)
int hml_ctx_gather_marginal_segments(hml_ctx* c, uint64_t* M, uint32_t** d_seg, int32_t** d_g);

#endif
