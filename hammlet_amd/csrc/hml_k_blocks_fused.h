// K4+K5+K6a in one launch: block starts from the weight summary, their global order, the block statistics and
// the emission terms - the per-sweep block structure of a dynamic chain without a dependent dispatch in
// between (a dependent boundary costs 1-4 us on this part, tools/dispatch_bench.hip - as much as the work it separates).
#ifndef HML_K_BLOCKS_FUSED_H
#define HML_K_BLOCKS_FUSED_H

#include "hml_k_blocks.h"
#include "hml_k_forward.h"

// ------------------------------------------------------------------------------------------
// One workgroup of 8 wavefronts per TILE of n_sub * 2^17 positions; wavefront w owns the tile's w-th eighth and walks it
// in n_sub batches of four 4096-position spans (the geometry of hml_k_compact_scan_summary).  n_sub (1..4) is chosen by
// the host so that the whole grid is resident at once (hml_fused_geometry): the kernel's time then grows smoothly with T
// instead of doubling when a second round of workgroups has to wait for the first.
//   phase A  per batch a wavefront lists its opened groups (LDS), opens them and keeps their 16-bit start masks (LDS);
//            the workgroup then publishes {generation, its number of block starts, the position of its last one} as
//            ONE 64-bit word (relaxed agent-scope store).  The masks are expanded into the wavefront's list of starts
//            - 16-bit offsets into its eighth, in position order, the first HML_FUSED_WAVE_LIST in LDS, any further
//            ones (compression below 64) in a staging array in memory - AFTER the word is out (for the last batch;
//            earlier batches of a long tile are expanded when their LDS arrays are needed again).
//   items    every thread takes its first item - the k-th start of the workgroup, which ENDS a block - and gathers
//            that block's statistics from the integral array (K5): loads only, they travel during the wait below.
//   offsets  every workgroup reads the words of ALL workgroups before it (763 at T = 10^8, two loads per thread),
//            polling words that do not carry this launch's generation yet: their sum is the index of its first block,
//            the last non-empty one gives the start of the block that ends at its first start.
//   phase B  emission terms (K6a) and the stores (starts, statistics, terms); any further items of the thread (more
//            than 512 starts in a tile: compression below 256 n_sub).  The last workgroup has one more item, the end marker:
//            starts[B] = T, the final block, and B.
// Nothing but the 64-bit words crosses workgroups, so no cache write-back is needed inside the launch.
//
// Progress.  A workgroup waits only for lower-numbered ones, and every workgroup publishes before it waits; with the
// whole grid resident (one chain per GPU - the supported layout - and the geometry above) nobody waits for a workgroup
// that cannot run.  HIP promises no dispatch order, though, and another process may share the GPU, so the wait is
// BOUNDED: a thread whose poll of word i has not succeeded after `spin_limit` tries computes that word itself from the
// summary (hml_fused_tile_word: the same deterministic function of the weights and the threshold), publishes it for
// everyone, and raises a flag in host-mapped memory; the host then takes the scan + scatter launches for this chain
// from the next sweep on (hml_capi.hip).  Slow when it happens, never wrong, never stuck.
// Same results as hml_k_compact_scan(+_summary) + hml_k_compact_scatter + hml_k_stats_emission, bit for bit.
// ------------------------------------------------------------------------------------------
#define HML_FUSED_WAVES 8                                                        // wavefronts per workgroup
#define HML_FUSED_WAVE_BATCH (HML_SUM_SPANS * HML_SPAN)                           // positions a wavefront scans per batch (16384)
#define HML_FUSED_SUB_POSITIONS (HML_FUSED_WAVES * HML_FUSED_WAVE_BATCH)          // positions per workgroup and batch (2^17)
#define HML_FUSED_MAX_SUB 4                                                      // batches per workgroup (16-bit offsets into a wavefront's eighth)
#define HML_FUSED_POS_BITS 19                                                    // log2(HML_FUSED_MAX_SUB * HML_FUSED_SUB_POSITIONS)
#define HML_FUSED_GEN_MASK ((1u << (64 - 2 * HML_FUSED_POS_BITS - 1)) - 1u)      // 25 bits
#define HML_FUSED_WAVE_LIST 256                                                  // starts per wavefront kept in LDS
__device__ __forceinline__ unsigned long long hml_group_word(uint32_t gen, uint32_t total, uint32_t last_rel) {
    // gen: 25 bits | total: POS_BITS + 1 bits (0..2^POS_BITS) | position of the last start in the tile: POS_BITS bits
    return ((unsigned long long)(gen & HML_FUSED_GEN_MASK) << (2 * HML_FUSED_POS_BITS + 1)) |
           ((unsigned long long)total << HML_FUSED_POS_BITS) | (unsigned long long)last_rel;
}
// the generation of a launch: the chain's epoch (one fused launch per sweep, the parameter kernel advances the epoch at
// a kernel boundary), so every workgroup of a launch derives the same value whenever it starts
__device__ __forceinline__ uint32_t hml_fused_generation(const hml_model* mdl) { return ((uint32_t)mdl->epoch + 1u) & HML_FUSED_GEN_MASK; }

// The word tile `tile` publishes, computed by ONE thread from the summary and the weights (the bounded wait's fallback).
__device__ __forceinline__ unsigned long long hml_fused_tile_word(const uint8_t* __restrict__ summary, const float* __restrict__ w, uint32_t T,
                                                               float thr, int32_t base, uint32_t tile, uint32_t n_sub, uint32_t gen) {
    const uint32_t n_spans = (uint32_t)(((uint64_t)T + HML_SPAN - 1) / HML_SPAN);
    const uint32_t kthr = (thr != thr) ? 0u : hml_weight_key(thr, base);
    const uint32_t first_span = tile * (uint32_t)HML_FUSED_WAVES * n_sub * HML_SUM_SPANS;
    const uint32_t spans = (uint32_t)HML_FUSED_WAVES * n_sub * HML_SUM_SPANS;
    uint32_t total = 0u, last1 = 0u;
    for (uint32_t s = 0; s < spans && first_span + s < n_spans; ++s) {
        const uint32_t span = first_span + s;
        for (uint32_t g = 0; g < 256u; ++g) {
            // byte j of word l holds group 64 j + l (hml_k_build_summary)
            const uint32_t key = summary[(uint64_t)span * 256u + (uint64_t)(g & 63u) * 4u + (g >> 6)];
            const bool origin = (span == 0u && g == 0u);
            if (key < kthr && !origin) continue;
            uint32_t m16 = hml_group_mask16(w, (uint64_t)span * HML_SPAN + (uint64_t)g * 16u, T, thr);
            if (origin) m16 |= 1u;
            if (m16) {
                total += (uint32_t)__popc(m16);
                last1 = s * HML_SPAN + g * 16u + (31u - (uint32_t)__clz((int)m16)) + 1u;   // positions grow with (s, g)
            }
        }
    }
    return hml_group_word(gen, total, last1 ? last1 - 1u : 0u);
}

template <int K>
HML_KERNEL __launch_bounds__(HML_FUSED_WAVES * 64, 6) void hml_k_blocks_fused(const uint8_t* __restrict__ summary, const float* __restrict__ w,
                                                          const float2* __restrict__ ia, uint32_t T,
                                                          hml_model* __restrict__ mdl, int32_t base,
                                                          unsigned long long* __restrict__ group_word,
                                                          uint16_t* __restrict__ stage, uint32_t* __restrict__ starts,
                                                          float2* __restrict__ bstat, float* __restrict__ em,
                                                          float* __restrict__ gsc, float* __restrict__ eprobe, int mixture,
                                                          const hml_layout lay, uint32_t* __restrict__ host_words,
                                                          uint32_t n_sub, uint32_t spin_limit, unsigned long long* __restrict__ dbg,
                                                          const hml_model* __restrict__ mdl_ro) {
    // mdl_ro: the same model through a read-only pointer - the parameters this kernel only reads (threshold, epoch, theta,
    // log-terms; written by the parameter kernel of the sweep before) then come through the scalar unit into scalar
    // registers instead of occupying 4 K + 3 vector registers per lane (the kernel is held to 80 VGPRs for residency)
    if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 4 + 0] = wall_clock64();
    static_assert(HML_FUSED_MAX_SUB * HML_FUSED_SUB_POSITIONS == (1 << HML_FUSED_POS_BITS), "tile geometry (bit fields of hml_group_word)");
    static_assert(HML_FUSED_MAX_SUB * HML_FUSED_WAVE_BATCH <= 65536, "16-bit offsets into a wavefront's eighth");
    constexpr int NW = HML_FUSED_WAVES;
    constexpr uint32_t NT = NW * 64;
    __shared__ uint16_t listed_all[NW][HML_SUM_SPANS * 256];   // per wavefront: opened groups of the current batch (span << 8 | group), position order
    __shared__ uint16_t mask_all[NW][HML_SUM_SPANS * 256];     // their start masks
    __shared__ uint16_t wave_list[NW][HML_FUSED_WAVE_LIST];    // per wavefront: its first starts (offsets into its eighth)
    __shared__ uint32_t wave_total[NW], wave_last[NW];         // block starts per wavefront; 1 + tile-relative position of the last
    __shared__ unsigned long long red_sum[NW], red_near[NW];
    __shared__ uint64_t sm_exp_tab[32];   // hml_expf's table: phase B looks it up at the end of the launch's critical path
    if (threadIdx.x < 32u) sm_exp_tab[threadIdx.x] = HML_EXP2F_TAB[threadIdx.x];   // (visible behind the barriers of phase A)
    constexpr bool LOOPED = K > 6;        // many states: emission parameters from LDS, states walked in a loop (hml_emit_block_looped)
    __shared__ hml_emit_lds<LOOPED ? K : 1> sm_emit;
    if (LOOPED) hml_emit_lds_fill<LOOPED ? K : 1>(sm_emit, mdl_ro, (int)threadIdx.x);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x;
    const uint32_t n_spans = (uint32_t)(((uint64_t)T + HML_SPAN - 1) / HML_SPAN);
    const uint32_t eighth = n_sub * (uint32_t)HML_FUSED_WAVE_BATCH;                   // positions per wavefront
    const uint32_t tile_positions = eighth * (uint32_t)NW;
    const uint64_t wave_base = ((uint64_t)g * NW + (uint32_t)wave) * eighth;          // first position of this wavefront's eighth
    const uint32_t gen = hml_fused_generation(mdl_ro);
    uint16_t* listed = listed_all[wave];
    uint16_t* masks = mask_all[wave];
    const float thr = mdl_ro->thr;
    const uint32_t cap = mdl_ro->cap;

    // ---------------- phase A
    uint32_t total = 0u, last1 = 0u;
    uint32_t placed = 0u;          // starts of this wavefront already in its list (ranks of the batches expanded so far)
    uint32_t n_listed = 0u, last_batch = 0u;
    // the starts of one batch, from its masks, appended to the wavefront's list in position order
    auto expand = [&](uint32_t j, uint32_t n_l) {
        for (uint32_t i0 = 0; i0 < n_l; i0 += 64u) {   // wave-uniform
            const uint32_t i = i0 + (uint32_t)lane;
            uint32_t m16 = 0u, in_eighth = 0u;
            if (i < n_l) {
                const uint32_t sg = listed[i];
                m16 = masks[i];
                in_eighth = j * (uint32_t)HML_FUSED_WAVE_BATCH + (sg >> 8) * HML_SPAN + (sg & 255u) * 16u;
            }
            const uint32_t c = (uint32_t)__popc(m16);
            uint32_t r, sum;
            const unsigned long long some = __ballot(c != 0u);
            if (__ballot(c > 1u) == 0ull) {   // wave-uniform: at most one start per group (the rule at strong compression)
                r = placed + hml_mbcnt(some);
                sum = (uint32_t)__popcll(some);
            } else {
                uint32_t incl = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(incl, d);
                    if (lane >= d) incl += o;
                }
                r = placed + incl - c;
                sum = __shfl(incl, 63);
            }
            while (m16) {
                const int bit = __ffs(m16) - 1;
                m16 &= m16 - 1u;
                const uint16_t off = (uint16_t)(in_eighth + (uint32_t)bit);
                if (r < (uint32_t)HML_FUSED_WAVE_LIST) wave_list[wave][r] = off;
                else stage[wave_base + r] = off;
                ++r;
            }
            placed += sum;
        }
    };
    {
        // NaN threshold: !(w < thr) holds everywhere, every position starts a block; key 0 opens every group
        const uint32_t kthr = (thr != thr) ? 0u : hml_weight_key(thr, base);
        const hml_swar_ge sw_ge = hml_swar_ge_make(kthr);
        auto load_batch = [&](uint32_t j, uint32_t (&gw)[HML_SUM_SPANS]) {
            const uint32_t span0 = (uint32_t)(wave_base / HML_SPAN) + j * HML_SUM_SPANS;
#pragma unroll
            for (int s = 0; s < HML_SUM_SPANS; ++s)
                gw[s] = (j < n_sub && span0 + s < n_spans)
                            ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(summary) + (uint64_t)(span0 + s) * 64u + lane)
                            : 0u;
        };
        uint32_t gw[HML_SUM_SPANS];
        load_batch(0u, gw);
        for (uint32_t j = 0; j < n_sub; ++j) {   // wave-uniform
            const uint32_t span0 = (uint32_t)(wave_base / HML_SPAN) + j * HML_SUM_SPANS;
            if (span0 >= n_spans) break;
            // the batch before this one leaves the LDS arrays: its starts go to the wavefront's list first.  (Only the LAST
            // batch is expanded after the workgroup has published - nobody should wait for this bookkeeping.)
            if (j > 0u) expand(j - 1u, n_listed);
            n_listed = 0u;
            last_batch = j;
#pragma unroll
            for (int s = 0; s < HML_SUM_SPANS; ++s) {
                const uint32_t fl = (span0 + s < n_spans) ? hml_swar_ge_apply(sw_ge, gw[s]) : 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // group 0 of span 0 is always opened: position 0 starts a block whatever its weight
                    const bool open = ((fl >> (8 * q + 7)) & 1u) || (span0 + s == 0u && q == 0 && lane == 0);
                    const unsigned long long m = __ballot(open);
                    if (open) listed[n_listed + hml_mbcnt(m)] = (uint16_t)((s << 8) | (64 * q + lane));
                    n_listed += (uint32_t)__popcll(m);
                }
            }
            load_batch(j + 1u, gw);   // the next batch's summary words travel while this batch's groups are opened
            // (LDS operations of one wavefront complete in order: the reads below see the writes above)
            for (uint32_t i0 = 0; i0 < n_listed; i0 += 64u) {   // wave-uniform; one pass unless > 64 groups are open
                const uint32_t i = i0 + (uint32_t)lane;
                uint32_t m16 = 0u;
                if (i < n_listed) {
                    const uint32_t sg = listed[i];
                    const uint32_t in_batch = (sg >> 8) * HML_SPAN + (sg & 255u) * 16u;
                    m16 = hml_group_mask16(w, (uint64_t)span0 * HML_SPAN + in_batch, T, thr);
                    if (span0 == 0u && i == 0u) m16 |= 1u;   // position 0 (group 0 of span 0 is listed first)
                    masks[i] = (uint16_t)m16;
                    if (m16) last1 = (uint32_t)wave * eighth + j * (uint32_t)HML_FUSED_WAVE_BATCH + in_batch + (31u - (uint32_t)__clz((int)m16)) + 1u;
                }
                total += (uint32_t)__popc(m16);
            }
        }
        total = hml_wave_sum_u32(total);
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { const uint32_t o = __shfl_xor(last1, m); last1 = o > last1 ? o : last1; }
    }
    if (lane == 0) { wave_total[wave] = total; wave_last[wave] = last1; }
    __syncthreads();
    if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 4 + 1] = wall_clock64();
    uint32_t wg_total = 0u, wg_last1 = 0u;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        wg_total += wave_total[k];
        wg_last1 = wave_last[k] > wg_last1 ? wave_last[k] : wg_last1;
    }
    if (threadIdx.x == 0) {
        const uint32_t l1 = wg_last1;
        __hip_atomic_store(&group_word[g], hml_group_word(gen, wg_total, l1 ? l1 - 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the last batch's starts join the list now that the word is out; the lists are read across wavefronts below
    if ((uint32_t)(wave_base / HML_SPAN) < n_spans) expand(last_batch, n_listed);
    __syncthreads();   // (also makes staged starts visible to the other wavefronts of the workgroup)

    // the k-th start of the workgroup, as a global position (k < wg_total)
    auto start_at = [&](uint32_t k) -> uint32_t {
        // the wavefront whose list holds rank k: the last one whose first rank is <= k (prefix of the eight totals, all
        // lanes read the same LDS words)
        uint32_t wv = 0u, first = 0u, run = 0u;
#pragma unroll
        for (int q = 0; q < NW - 1; ++q) {
            run += wave_total[q];
            if (k >= run) { wv = (uint32_t)q + 1u; first = run; }
        }
        const uint32_t idx = k - first;
        const uint64_t wb = ((uint64_t)g * NW + wv) * eighth;
        const uint32_t off = (idx < (uint32_t)HML_FUSED_WAVE_LIST) ? (uint32_t)wave_list[wv][idx] : (uint32_t)stage[wb + idx];
        return (uint32_t)(wb + off);
    };
    const bool last_wg = (g == gridDim.x - 1u);
    // items: one per start (write it, finish the block that ends there); the last workgroup has one more, the
    // end marker T, which finishes the final block
    const uint32_t n_items = wg_total + (last_wg ? 1u : 0u);

    // ---------------- first items: every thread's first item but the workgroup's very first (whose block begins in an
    // earlier workgroup): the block's statistics only need positions, so their gathers travel during the wait.  The
    // arithmetic (emission terms) waits for the offsets on purpose: run earlier it competes with the phase A of
    // workgroups that started later - and everybody waits for the slowest of those (measured: +4.7 us per launch).
    const uint32_t k_first = threadIdx.x;
    bool have_first = false;
    uint32_t first_t = 0u, first_n = 0u;
    float first_sx = 0.0f, first_sq = 0.0f;
    if (k_first > 0u && k_first < n_items) {
        first_t = (k_first < wg_total) ? start_at(k_first) : T;
        const uint32_t prev_t = start_at(k_first - 1u);
        first_n = first_t - prev_t;
        hml_block_stats_one(ia, prev_t, first_t, first_sx, first_sq);
        have_first = true;
    }

    // ---------------- offsets: sum over all earlier tiles, and the last start before this tile
    unsigned long long acc = 0ull, near = 0ull;   // near: (1 + tile index) << POS_BITS | last_rel of the last non-empty earlier one
    {
        for (uint32_t i = threadIdx.x; i < g; i += NT) {
            // (spin_limit 0 is the tests' setting: do not even look - every word is computed here)
            unsigned long long d = spin_limit ? __hip_atomic_load(&group_word[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                              : hml_group_word(gen + 1u, 0u, 0u);
            uint32_t tries = 0u;
            while ((uint32_t)(d >> (2 * HML_FUSED_POS_BITS + 1)) != gen) {
                if (tries++ >= spin_limit) {
                    // the owner of tile i has not published in time (not resident? see "Progress" above): its word is a
                    // function of the weights and the threshold, so compute it here and publish it for everyone
                    d = hml_fused_tile_word(summary, w, T, thr, base, i, n_sub, gen);
                    __hip_atomic_store(&group_word[i], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (host_words) __hip_atomic_store(host_words + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    atomicAdd(&mdl->fused_fallbacks, 1ull);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                d = __hip_atomic_load(&group_word[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const uint32_t tot = (uint32_t)(d >> HML_FUSED_POS_BITS) & ((2u << HML_FUSED_POS_BITS) - 1u);
            acc += tot;
            if (tot) near = ((unsigned long long)(i + 1u) << HML_FUSED_POS_BITS) | (d & ((1ull << HML_FUSED_POS_BITS) - 1ull));   // i grows within a thread
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        acc += __shfl_xor(acc, m);
        const unsigned long long o = __shfl_xor(near, m);
        near = o > near ? o : near;
    }
    if (lane == 0) { red_sum[wave] = acc; red_near[wave] = near; }
    __syncthreads();
    if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 4 + 2] = wall_clock64();
    unsigned long long sum_all = 0ull, nr = 0ull;
#pragma unroll
    for (int k = 0; k < NW; ++k) { sum_all += red_sum[k]; nr = red_near[k] > nr ? red_near[k] : nr; }
    const uint32_t before_group = (uint32_t)sum_all;
    // global position of the last start before this tile (tile 0 holds position 0, so it exists for g > 0)
    const uint32_t prev_group_start =
        nr ? (uint32_t)(((nr >> HML_FUSED_POS_BITS) - 1ull) * tile_positions + (nr & ((1ull << HML_FUSED_POS_BITS) - 1ull))) : 0u;

    // ---------------- phase B: emission terms and the writes
    hml_emit_params<LOOPED ? 1 : K> p;
    if (!LOOPED) hml_emit_load<LOOPED ? 1 : K>(p, mdl_ro, mixture);
    const bool self_trans = mdl_ro->self_trans != 0;
    {
        uint32_t t = first_t, n = first_n;
        float sx = first_sx, sq = first_sq;
        bool gathered = have_first;
        for (uint32_t k = threadIdx.x; k < n_items; k += NT) {
            const uint32_t b = before_group + k;
            if (!gathered) {
                t = (k < wg_total) ? start_at(k) : T;
                if (t != 0u) {
                    const uint32_t prev_t = (k > 0u) ? start_at(k - 1u) : prev_group_start;
                    n = t - prev_t;
                    hml_block_stats_one(ia, prev_t, t, sx, sq);
                }
            }
            gathered = false;
            if (b > cap) continue;   // beyond the chain's block capacity (hml_state.h): the last workgroup halts the chain below
            starts[b] = t;   // (item wg_total of the last workgroup: starts[B] = T)
            if (t == 0u) continue;   // no block ends at position 0
            bstat[b - 1u] = make_float2(sx, sq);
            if constexpr (LOOPED)
                hml_emit_block_looped<K>(reinterpret_cast<const hml_emit_lds<K>&>(sm_emit), mdl, self_trans, b - 1u, sx, sq, (float)n, em, gsc, eprobe, mixture, lay, sm_exp_tab);
            else
                hml_emit_block<LOOPED ? 1 : K, true>(p, mdl, b - 1u, sx, sq, (float)n, em, gsc, eprobe, mixture, lay, sm_exp_tab);
        }
    }
    if (dbg) { __syncthreads(); if (threadIdx.x == 0) dbg[blockIdx.x * 4 + 3] = wall_clock64(); }
    // the block count
    if (last_wg && threadIdx.x == 0) {
        const uint32_t Bn = before_group + wg_total;
        if (Bn > cap) { hml_halt(mdl, Bn, host_words); return; }
        mdl->B = Bn;
        hml_warmup_for_many_blocks(mdl, Bn);
        // host-mapped word: lets the host size later grids without a copy in the stream
        if (host_words) __hip_atomic_store(host_words, Bn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#endif
