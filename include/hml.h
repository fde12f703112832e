/* hml.h - C ABI of the MI355X-native HaMMLET hot path (libhammlet_hip.so).
 *
 * The reference (wiedenhoeft/HaMMLET) is one header-only C++ program with no FFI seam; the
 * types it instantiates in src/main.cpp:338-362,437,444 are what a drop-in has to stand behind.
 * Every entry point below names the reference interface it replaces (file:line relative to the
 * reference's repository root).  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a non-zero
 * code on failure, with the message available from hml_last_error() (the text of the
 * std::runtime_error the reference would have thrown, where one exists).  A context owns all
 * device memory of one chain on one GPU and is not thread-safe (the reference is single-threaded
 * with one shared RNG, src/main.cpp:108).  All work is enqueued on the context's HIP stream;
 * functions that return data to the host synchronise that stream.
 */
#ifndef HML_H
#define HML_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hml_ctx hml_ctx;

/* sampling methods of hml_iterate (src/main.cpp:432-445: "F" and "M" scheme tokens) */
#define HML_METHOD_FB 'F'
#define HML_METHOD_MIXTURE 'M'

/* error codes */
#define HML_OK 0
#define HML_ERR_ARG 1      /* invalid argument / call order                                  */
#define HML_ERR_HIP 2      /* HIP runtime failure                                            */
#define HML_ERR_MODEL 3    /* a model invariant the reference enforces by throwing was hit   */

const char* hml_last_error(void);

/* Library/ABI version and the GPU architecture the kernels were compiled for ("gfx950"). */
uint32_t hml_abi_version(void);
const char* hml_device_arch(void);
/* number of GPUs this process can use (hipGetDeviceCount) */
int hml_device_count(int* n);

/* rng_t RNG(seed) (src/main.cpp:107-108) + one chain's device state.  `stream` may be NULL
 * (a private stream is created) or a hipStream_t owned by the caller. */
int hml_create(hml_ctx** out, int device, uint64_t seed, uint32_t chain_id, void* stream);
void hml_destroy(hml_ctx* ctx);

/* MaxletTransform + noise estimate + HaarBreakpointWeights + Statistics<IntegralArray,Normal> +
 * Blocks<BreakpointArray> constructors (src/wavelet.hpp:68-188, src/main.cpp:303-318,340-341,
 * src/Statistics/IntegralArray.hpp:136-191, src/Blocks/BreakpointArray.hpp:130-184).
 * hml_load_observations: n_values = T host floats - T * D after hml_set_dimensions(D, .), the D values of a position one
 * after the other.  hml_load_observations_device: T device floats (univariate; sigma-hat is then computed from a host copy
 * made internally). */
int hml_load_observations(hml_ctx* ctx, const float* x, uint64_t n_values);
int hml_load_observations_device(hml_ctx* ctx, const void* x_dev, uint64_t T);

/* A further chain over the SAME observations on the same GPU (chains outnumbering GPUs: `hammlet -chains N`, several
 * chains per rank): `ctx` shares `source`'s read-only construction - breakpoint weights and their summary, maxlet
 * coefficients, integral arrays; reference counted, freed with the last context - instead of uploading and building a copy
 * of its own; block structure, sweep buffers, model and marginals stay per chain.  Chains of one run read the same trace
 * and their block sets are nested by threshold, so their gathers then hit the same cache lines (hml_iterate_many batches
 * such chains through ONE block kernel).  Replaces hml_load_observations for `ctx`; dimensions (hml_set_dimensions) are taken
 * from the source.  hml_scale_weights / hml_set_weights are refused while the weights are shared: apply them to the source
 * first.  The reference builds the construction once per process for its single chain (src/main.cpp:286-343). */
int hml_attach_observations(hml_ctx* ctx, hml_ctx* source);

/* Text input: the values that `while ( input >> v )` extracts from a whitespace-separated decimal stream
 * (src/wavelet.hpp:131, called from src/main.cpp:266-291), converted on the GPU chunk by chunk.  The result is
 * bit-identical to the stream extraction, including where it stops: tokens the device cannot decide with proof
 * (more than 19 significant digits, ties, sub-normal or overflowing values, anything that is not one plain decimal
 * number from blank to blank) are re-read by the host with that same extraction.
 *   hml_text_open(&r, device, chunk_bytes);          (chunk_bytes 0 = 64 MiB staging)
 *   loop: hml_text_buffer(r, &buf, &cap); n = read(fd, buf, cap); hml_text_commit(r, n);   (or hml_text_feed)
 *   hml_text_finish(r, &n_values, &stopped);  hml_text_values(r, out);  hml_text_close(r);
 * `stopped` = 1 if an extraction failed before the end of the text (the reference silently stops reading there). */
typedef struct hml_text hml_text;
int hml_text_open(hml_text** out, int device, uint64_t chunk_bytes);
void hml_text_close(hml_text* reader);
int hml_text_buffer(hml_text* reader, char** buf, uint64_t* capacity);
int hml_text_commit(hml_text* reader, uint64_t nbytes);
int hml_text_feed(hml_text* reader, const char* bytes, uint64_t nbytes);
/* optional: an upper estimate of the number of values (the reference's reserveT, src/wavelet.hpp:103,123-124) */
int hml_text_reserve(hml_text* reader, uint64_t n_values);
int hml_text_finish(hml_text* reader, uint64_t* n_values, int* stopped);
int hml_text_values(hml_text* reader, float* out /* n_values */);
/* bytes consumed, tokens resolved by the host, chunks that went through the host extraction entirely */
int hml_text_counters(hml_text* reader, uint64_t* bytes_in, uint64_t* irregular_tokens, uint64_t* host_chunks);

/* "-s C P D" (src/main.cpp:114-137, src/Mapping.hpp:53-137): D data dimensions whose values follow each other in the
 * observation stream and P emission parameters shared by K = P^D states (state s uses parameter (s / P^d) % P for
 * dimension d).  Call before hml_load_observations (which then takes T * D values); hml_set_model's K must be P^D;
 * hml_get_theta / hml_set_parameters carry P (mean, variance) pairs.  P = 0 here: taken from hml_set_model's K.
 * Default: D = 1, P = K. */
int hml_set_dimensions(hml_ctx* ctx, int D, int P);
int hml_get_dimensions(hml_ctx* ctx, int* D, int* P);

/* Blocks<BreakpointArray>(vector<real_t>& weights) (src/Blocks/BreakpointArray.hpp:130-184, src/main.cpp:341): replaces
 * the breakpoint weights the device computed by the caller's T host values - for drivers that, like the reference's,
 * hold the weights on the host between HaarBreakpointWeights and the Blocks constructor (and may have changed them). */
int hml_set_weights(hml_ctx* ctx, const float* w, uint64_t T);

/* stdEstimate of src/main.cpp:303-311 */
int hml_noise_sigma(hml_ctx* ctx, double* sigma);

/* `for (auto& w : inputValues) w *= weightMultiplier` (src/main.cpp:332-334) */
int hml_scale_weights(hml_ctx* ctx, float multiplier);

/* autoPrior (src/AutoPriors.hpp:86-110, :18-80): s2 = VAR and p = P of "-e normal VAR P".
 * out4 = {alpha, beta, mu0, nu}.  Leaves the universal threshold as the current threshold. */
int hml_autoprior(hml_ctx* ctx, float s2, float p, float out4[4]);

/* Mapping/Transitions/Initial/TransitionHyperParam/InitialHyperParam/ThetaHyperParam/Theta
 * construction (src/main.cpp:133-166,354-362).  nig4 = {alpha,beta,mu0,nu} shared by all K
 * emission parameters; a_off/a_diag = "-t" tokens; pi_alpha = "-I"; self_trans = !"-S".
 * Like Theta's constructor (src/Theta.hpp:126-127) this draws theta once from the prior.
 * K: 2 .. 64 (the reference takes any -s K, src/main.cpp:112-137).  Up to 16 states the sweep runs kernels instantiated for that
 * number of states; from 17 on the number of states is a run-time value and a state is a lane (hml_k_wide.h) - the same chain
 * semantics (Philox addresses, hml_math.h, the count tree), about three times the time per block of a 16-state model. */
int hml_set_model(hml_ctx* ctx, int K, const float nig4[4], float a_off, float a_diag, float pi_alpha,
                  int self_trans);

/* useSelfTransitions, the last argument of StateSequence::sample / sampleHMM (src/StateSequence.hpp:64, src/HMM.hpp:75):
 * a driver in the reference's shape only states it when it samples, after the model objects exist. */
int hml_set_self_transitions(hml_ctx* ctx, int on);

/* theta.sample(tau_theta); pi.sample(tau_pi); A.sample(tau_A) from the priors
 * (src/main.cpp:393-401, and again after a "P" token). */
int hml_sample_prior(hml_ctx* ctx);

/* "S" token: y.createBlocks(theta); dynamic = false (src/main.cpp:407-414).
 * "D" token: dynamic = true (src/main.cpp:415-421). */
int hml_set_static_blocks(hml_ctx* ctx);
int hml_set_dynamic(hml_ctx* ctx, int on);

/* Emissions::createBlocks(real_t) (src/Emissions.hpp:49-51) + a full enumeration with block
 * statistics (Emissions::next, src/Emissions.hpp:88-95): parity probe for block structures. */
int hml_create_blocks(hml_ctx* ctx, float threshold);

/* sampleHMM (src/HMM.hpp:60-125) with StateSequence<ForwardBackward> or <Mixture>
 * (src/StateSequence/ForwardBackward.hpp:16-213, Mixture.hpp:31-144): `iterations` Gibbs sweeps,
 * recording when thinning > 0 && (i+1) % thinning == 0.  Device-resident; returns after the
 * sweeps are enqueued unless per-sweep side files were requested with hml_set_recording. */
int hml_iterate(hml_ctx* ctx, char method, uint64_t iterations, uint64_t thinning);

/* Records::setRecord* (src/Records.hpp:121-144).  marginals: accumulate state marginals on the
 * device.  The other four make hml_iterate call `cb` after every recorded sweep (with the stream
 * synchronised) so the host can pull blocks/states/theta and append its files
 * (src/Records.hpp:147-235). */
typedef void (*hml_record_cb)(hml_ctx* ctx, uint64_t sweep_in_call, void* user);
int hml_set_recording(hml_ctx* ctx, int marginals, hml_record_cb cb, void* user);

/* Options.  "weight_keys" (before the observations are loaded), 1 (default): the per-sweep block scan reads a
 * one-byte-per-16-positions summary of the breakpoint weights (largest monotone 8-bit key of the group) and opens only
 * the groups that can hold a block start, comparing their float weights exactly (sweeps whose compression is
 * below 24 positions per block stream the floats instead); 2: the summary at any compression; 0: always stream all T
 * float weights.  Same block structures every way (exact). */
int hml_set_option(hml_ctx* ctx, const char* name, int value);
/* "max_blocks" (before the observations are loaded; univariate models): the BLOCK CAPACITY of the context's per-block buffers
 * (block starts, statistics, emission terms, trellis rows, maps, states) - by default T, the worst case of every position a
 * block (about 100 bytes per position: 10 GB at 10^8 positions and 5 states), and max(2^20, T / 16) for a context attached
 * to another one's observations (hml_attach_observations).  A sweep whose enumeration finds more blocks writes nothing beyond
 * the capacity and halts the chain on the device; the host then allocates larger buffers and runs the missing sweeps again,
 * with the same results (hml_stats.buffer_growths counts how often).  0: the default.  Environment: HML_MAX_BLOCKS. */
/* "fused_blocks" (any time), 1 (default): dynamic sweeps take the fused block kernel (block scan + block statistics +
 * emission terms in one launch; its workgroups hand block offsets to each other inside the launch, with a bounded
 * wait: when the GPU is shared and a wait expires the waiting workgroup computes the missing word itself, and the chain
 * takes the other path from the next sweep on); 0: always the scan + scatter + statistics launches, which share
 * nothing inside a launch - the setting for a GPU that several processes use; 2: like 1, but keep the kernel after an
 * expired wait (tests).  Same results every way.  Environment: HML_FUSED_BLOCKS.
 * "trellis_L" (any time): chunk length of the fused trellis kernels that weakly compressed univariate FB sweeps take
 * (millions of blocks; hml_k_trellis.h).  0 (default): chosen from the number of blocks and then, after 48 such sweeps,
 * by measurement - every candidate length runs two sweeps between a pair of events and the fastest stays; a multiple of
 * 32 up to 1024: that length.  A launch geometry only: rows, maps and draws are addressed by block, so the chain's
 * results are the same for every length.  Environment: HML_TRELLIS_L, HML_TRELLIS_TUNE=0 (no measurement).
 * "compat" (before hml_set_model), 0 (default) / 1: the REFERENCE-COMPATIBLE mode.  The default path addresses every
 * random decision by a Philox counter, uses its own logf / powf, sums block statistics over a fixed tree and counts in
 * exact integers: same posterior, but a chain of its own for every seed.  With compat = 1 a
 * sweep is computed exactly as the reference's single thread computes it - one std::mt19937 seeded like
 * `rng_t RNG(seed)` (src/main.cpp:107-108) and consumed in its order, glibc's expf / logf / powf bit for bit
 * (hml_math_glibc.h), float Kahan sums in block order, `size_t += float` counts - so that a run with the reference's
 * seed leaves the reference's states, parameters and marginals.  The order-dependent part of a sweep keeps the reference's
 * order but runs in chunks that are checked against each other (hml_k_compat.h; a few milliseconds per sweep of config 3's
 * 1.8 10^5 blocks - the default path is the fast one: 0.056 ms).  2 .. 64 states, like the default path (which takes the
 * number of states as a run-time value from 17 states on: hml_k_wide.h).  Environment: HML_COMPAT. */

/* hml_iterate for SEVERAL chains at once: `iterations` sweeps of every chain, sweep i of all chains before sweep i + 1.
 * Chains that live on one device, have the same shape (positions, states) and are in the strongly compressed regime of a
 * univariate Forward-Backward sweep are BATCHED: every kernel of the sweep is launched once for all of them (the chain
 * is the grid's second dimension), so the host pays for one chain's launches - a single such chain is bound by latency
 * and leaves most of the GPU idle.  Chains that share ONE construction (hml_attach_observations) additionally take one
 * block kernel for up to eight of them (block starts, statistics and emission terms of all chains from one pass over the
 * shared summary / weights / integral array): eight chains of 10^8 positions and 5 states reach 2.5 times one chain's rate.  Everything else (other shapes or devices, mixture sweeps, weakly compressed,
 * multivariate or reference-compatible chains) is run chain by chain inside the same call.  A chain's results are the
 * same bit for bit as under hml_iterate; recorded sweeps call each chain's callback in chain order. */
int hml_iterate_many(hml_ctx* const* ctxs, int n, char method, uint64_t iterations, uint64_t thinning);

/* wait for all enqueued work; surfaces model errors raised on the device */
int hml_sync(hml_ctx* ctx);

/* ---- probes: state of the last sweep (y.start()/end()/blockSize()/suffStat(), q.states(),
 * theta/A/pi values; src/Emissions.hpp:66-99, src/StateSequence.hpp:71-74) ---- */
int hml_get_num_blocks(hml_ctx* ctx, uint64_t* B);
int hml_get_blocks(hml_ctx* ctx, uint32_t* starts /* B+1, last = T */);
int hml_get_block_stats(hml_ctx* ctx, float* sum /*D*B, dimension-major*/, float* sum_sq /*D*B*/);
int hml_get_states(hml_ctx* ctx, int16_t* q /*B*/);
int hml_get_theta(hml_ctx* ctx, float* mean_var /*2K: mean0,var0,mean1,...*/);
int hml_get_transitions(hml_ctx* ctx, float* A /*K*K row-major*/, float* pi /*K*/);
int hml_set_parameters(hml_ctx* ctx, const float* mean_var, const float* A, const float* pi);
int hml_get_threshold(hml_ctx* ctx, float* thr);
/* E_s of src/StateSequence/ForwardBackward.hpp:74-81 for every block of the last sweep and the
 * normalised forward rows alpha_t (row 0 = pi); only filled when probes are enabled. */
int hml_enable_probes(hml_ctx* ctx, int on);
int hml_get_block_loglik(hml_ctx* ctx, float* E /*B*K*/);
int hml_get_forward_rows(hml_ctx* ctx, float* rows /*(B+1)*K*/);
/* sufficient statistics of the last sweep's count pass (ForwardBackward.hpp:170-200) */
int hml_get_counts(hml_ctx* ctx, uint64_t* trans /*K*K*/, uint64_t* occ /*K*/, float* sum /*K*/,
                   float* sum_sq /*K*/, uint64_t* nterms /*K*/);
/* construction probes */
int hml_get_weights(hml_ctx* ctx, float* w /*T*/);
int hml_get_coefficients(hml_ctx* ctx, float* c /*T*/);   /* maxlet coefficients (before weights) */
int hml_get_integral_array(hml_ctx* ctx, float* sum /*T+1*/, float* sum_sq /*T+1*/);

/* ---- results ---- */
/* StateMarginals (src/StateMarginals.hpp:51-137,268-310): run-length form.  Call with
 * seg_len == NULL to obtain the number of segments and of printed state columns. */
int hml_marginals_rle(hml_ctx* ctx, uint64_t* n_segments, int* n_columns, uint64_t* seg_len /*n_segments*/,
                      int32_t* counts /*n_segments * n_columns*/);
/* Maximum-posterior-margin segmentation of the recorded marginals, computed on the device - the post-processing
 * step of src/tools/maxSegmentation.cpp:53-82 without the round trip through the marginals file: arg-max state of
 * every marginal segment (first maximum; state 0 if all counts are zero), adjacent segments of equal state merged.
 * Call with run_len == NULL to obtain the number of runs. */
int hml_max_segmentation(hml_ctx* ctx, uint64_t* n_runs, uint64_t* run_len /*n_runs*/, int32_t* run_state /*n_runs*/);
/* dense per-position counts, [K+1][T] int32 on the DEVICE (row K = 1 at segment boundaries), with
 * the state rows permuted by `perm` (perm[new] = old; NULL = identity): the buffer that the
 * chain-parallel pooling all-reduces over RCCL. */
int hml_marginals_dense_device(hml_ctx* ctx, void* out_dev, const int32_t* perm);
int hml_recorded_sweeps(hml_ctx* ctx, uint64_t* n);

/* Trellis::sample(t) (src/Trellis.hpp:61-66): one draw of std::discrete_distribution over K weights - p_i = w_i / sum in
 * double, first i whose cumulative probability reaches u - with u from the chain's Philox key (sub-stream HOST,
 * one counter step per call).  Runs on the host (shared arithmetic of the kernels, hml_dist.h). */
int hml_categorical_draw(hml_ctx* ctx, const float* weights, int K, uint32_t* index);

/* ---- chain-parallel pooling.  Chains shard across GPUs and never communicate while sampling;
 * one ncclAllReduce(sum, int32) over xGMI pools their recorded state marginals at the end.  The reference has no
 * counterpart (one process, one thread, src/main.cpp:108). ---- */
/* Common labels: perm[new] = old with the states ordered by ascending emission mean of the current theta - for
 * "-s C P D" by the tuple of the means of their mapped parameters, in dimension order; ties keep their order.  (The
 * idea of bin/sortStates:1-6, which orders states by their last sampled mean.) */
int hml_relabel_permutation(hml_ctx* ctx, int32_t* perm /*K*/);
/* The payload one chain contributes: int32 [K+1][T+1] - rows 0..K-1 the relabelled per-state difference arrays of the
 * recorded marginals, row K = 1 at recorded segment boundaries - followed by [recorded sweeps, used[0..K-1]].
 * hml_pool_export fills a device buffer of hml_pool_payload_size elements, hml_pool_install makes a (summed) payload
 * the context's marginals: hml_marginals_rle, hml_max_segmentation, hml_marginals_dense_device and hml_recorded_sweeps
 * then describe the pooled chains.  Any transport may sum the payloads in between. */
int hml_pool_payload_size(hml_ctx* ctx, uint64_t* n_int32);
int hml_pool_export(hml_ctx* ctx, void* payload_dev, int32_t* perm_out_or_null);
int hml_pool_install(hml_ctx* ctx, const void* payload_dev);
/* RCCL transport, one process per GPU: rank 0 creates the id (ncclGetUniqueId) and the launcher hands its
 * HML_POOL_ID_BYTES bytes to every rank; hml_pool_create is ncclCommInitRank (collective).  hml_pool_marginals =
 * export + ncclAllReduce(sum, int32) on the pool's stream + install (collective; every rank ends with the same
 * pooled marginals).  RCCL is loaded on the first of these calls (librccl.so.1). */
typedef struct hml_pool hml_pool;
#define HML_POOL_ID_BYTES 128
int hml_pool_unique_id(void* id /*HML_POOL_ID_BYTES*/);
int hml_pool_create(hml_pool** out, int device, int rank, int n_ranks, const void* id);
void hml_pool_destroy(hml_pool* pool);
int hml_pool_marginals(hml_pool* pool, hml_ctx* ctx, int32_t* perm_out_or_null);
int hml_pool_info(hml_pool* pool, int* rank, int* n_ranks, double* last_allreduce_ms, uint64_t* last_bytes, int* rccl_version);
/* The collective of hml_pool_marginals has two forms with the same result.  DENSE: the payload above through
 * ncclAllReduce(sum) - 4 (K+1)(T+1) bytes whatever it holds (2.4 GB at 10^8 positions and 5 states).  LISTS: the difference
 * arrays are zero except at recorded segment boundaries, so every rank sends the list of its marginal segments - header
 * [M, recorded sweeps, used[0..K-1]], then M x [position, relabelled deltas 0..K-1] - through ncclAllGather and adds all
 * lists into zeroed arrays (config 3 after 100 recorded sweeps: 23 000 segments, 0.6 MB per rank).  form 0 (default): the
 * lists when the gathered slots are at most an eighth of the dense payload - every rank decides alike from a handshake
 * that carries the ranks' segment counts; 1: always dense; 2: always lists.  Environment: HML_POOL_FORM.  The form is part
 * of the handshake: when the ranks hold different ones hml_pool_marginals returns HML_ERR_ARG on ALL of them.
 * hml_pool_last: the form the last call took (1 / 2) and, for the lists, the slot size in segments. */
int hml_pool_set_form(hml_pool* pool, int form);
int hml_pool_last(hml_pool* pool, int* form, uint64_t* entries);
/* One process driving n chains, e.g. one per GPU (`hammlet -chains N`): contexts sharing a device are summed there,
 * the per-device sums go through one grouped ncclAllReduce (ncclCommInitAll over the distinct devices), every context
 * receives the pooled marginals. */
int hml_allreduce_marginals(hml_ctx* const* ctxs, int n);
/* The same with the relabelling it applied: perms[i * K + j] = chain i's own label of pooled state j (may be null).  Chains
 * that all share one device are summed on it without RCCL. */
int hml_allreduce_marginals_perm(hml_ctx* const* ctxs, int n, int32_t* perms /* n * K */);
/* LABEL SPACES.  Pooling puts the context's MARGINALS (hml_marginals_rle, hml_max_segmentation, hml_marginals_dense_device)
 * into the common labels - states by ascending mean - while its parameters, state sequences and transition counts keep the
 * chain's own labels: perm[j] = the chain's label of pooled state j (the identity before any pooling).  A pooled context
 * refuses a second pooling (it would relabel and count twice) and refuses to record further sweeps into its marginals. */
int hml_pool_permutation(hml_ctx* ctx, int32_t* perm /*K*/);

/* ---- counters for measurement ---- */
typedef struct {
    uint64_t sweeps;            /* Gibbs sweeps executed                                   */
    uint64_t block_updates;     /* sum over sweeps of the number of blocks                 */
    uint64_t uniform_fallbacks; /* "[WARNING] Uniform sampling of forward variables!" count */
    uint64_t forward_refits;    /* chunks whose speculative forward pass had to be redone  */
    uint64_t forward_serial;    /* chunks finished by the sequential fallback              */
    uint64_t forward_warmup;     /* current (adaptive) warm-up length of the speculative forward pass */
    uint64_t fused_fallbacks;    /* tile words of the fused block kernel computed by a waiting workgroup (bounded wait expired) */
    uint64_t buffer_growths;     /* times the per-block buffers were grown (option "max_blocks", attached contexts) */
    uint64_t block_capacity;     /* blocks per sweep the per-block buffers hold at present */
} hml_stats;
int hml_get_stats(hml_ctx* ctx, hml_stats* out);

/* HIP-event timing of one named kernel family accumulated since the last reset (milliseconds and
 * launches); name is one of "blocks_compact", "blocks_scatter", "block_stats", "stats_emission", "emission", "forward",
 * "backward_maps", "backward_chain", "mixture", "counts", "params", "marginals", "event_null".  level 0 = off, 1 = only the dominant kernel
 * ("blocks_compact", two events per sweep), 2 = every family. */
int hml_profile_enable(hml_ctx* ctx, int level);
int hml_profile_get(hml_ctx* ctx, const char* name, double* total_ms, uint64_t* launches);

/* parity probe for the arithmetic shared between the kernels and the CPU checker (hml_math.h, hml_dist.h):
 * evaluates function `fn` elementwise on the GPU (0 expf, 1 logf, 2 pow(a,b) on (0,1], 3 sqrtf, 4 a/b,
 * 5 gamma(alpha=a, beta=b) draw, 6 normal(mean=a, sd=b) draw, 7-10 double-precision kernels). */
int hml_debug_eval(int device, int fn, const float* a, const float* b_or_null, float* out, uint64_t n, uint64_t seed);

/* synthetic piecewise-constant Gaussian trace, host buffer */
int hml_synth_gauss(float* x, int16_t* states_or_null, uint64_t T, int K, const float* mu, float sigma,
                    double mean_dwell, uint64_t seed, int nthreads);

/* simulated read-depth trace: copy-number segments, Poisson-lognormal counts as floats */
int hml_synth_depth(float* x, int16_t* states_or_null, uint64_t T, double depth, double ln_sigma, uint64_t seed, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
