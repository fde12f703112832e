// K4+K5+K6a for SEVERAL chains over ONE shared trace in one launch (hml_iterate_many over contexts attached to the same
// observations, hml_attach_observations): a workgroup owns a tile of the trace for ALL chains of the batch.
//
// Why not the single-chain kernel with the chain as a second grid dimension: that kernel is three dependent memory round
// trips (summary -> weights -> integral array) and one inter-workgroup hand-off per tile, and a chain's launch keeps the
// machine's workgroup slots for the whole of it - eight chains are eight times the latency chain through the same slots.
// The chains of one run read the same trace, and their thresholds differ by a few percent, so here
//   * a wavefront reads its summary words ONCE and opens the groups the LOWEST threshold of the batch opens (a superset of
//     every chain's groups: the key is monotone);
//   * a lane loads the 16 weights of its group ONCE and compares them with every chain's threshold - the chains' start
//     masks, ranks and lists of starts fall out of the same registers;
//   * the gathers of the integral array for the blocks of all chains are issued together, before the wait for the
//     offsets, and most of them hit the lines a sibling chain's block fetched (the block sets are nested by threshold);
//   * one word per (tile, chain) crosses workgroups; a thread polls tile i's words of all chains together.
// Per chain the results are those of hml_k_blocks_fused (hence of scan + scatter + statistics + emission), bit for bit:
// the same block starts in the same order, hml_block_stats_one's operations, hml_emit_block's terms.
// Reference: Blocks<BreakpointArray>::next src/Blocks/BreakpointArray.hpp:216-235, addBlockStats
// src/Statistics/IntegralArray.hpp:104-124, emission terms src/StateSequence/ForwardBackward.hpp:67-84 - once per chain
// (the reference runs one chain, src/main.cpp:108).
#ifndef HML_K_BLOCKS_FUSED_MANY_H
#define HML_K_BLOCKS_FUSED_MANY_H

#include "hml_k_blocks_fused.h"
#include "hml_k_trellis_rows.h"   // hml_tr2_stats: the statistics of a block inside one cell of the integral array

#define HML_FM_MAX_CHAINS 8      // chains per launch (more: the host launches groups of eight)
#define HML_FM_LIST 128          // starts per wavefront and chain kept in LDS (further ones go through the chain's staging array)
#ifndef HML_FM_MIN_WAVES
#define HML_FM_MIN_WAVES 4       // wavefronts per SIMD the register allocation aims at: 91 VGPRs, no scratch, two workgroups per CU (tiles of 2^18
                                 // positions at 10^8).  6 (three workgroups per CU, tiles of 2^17) spills 13 registers and is as fast: 0.179-0.181
                                 // against 0.178 ms per round of eight chains
#endif

// what the kernel needs of one chain
struct hml_fm_chain {
    hml_model* mdl;
    unsigned long long* group_word;
    uint16_t* stage;
    uint32_t* starts;
    float2* bstat;
    float* em;
    float* gsc;           // nullptr: no plane of rescale factors (late_rescale)
    uint32_t* host_words;
    hml_layout lay;       // the chain's chunk-transposed layout (its stride follows its own block capacity)
};
struct hml_fm_args {
    hml_fm_chain c[HML_FM_MAX_CHAINS];
};

// The emission terms of one block, the chain's parameters from LDS (the chain differs from item to item).  Up to 6 states:
// the forms of the weakly compressed sweep's first pass (hml_k_trellis_rows.h) - every E_s through the double reciprocal with
// ONE test per block for "could round differently" (then, and only then, hml_inner_product's literal form), max E by v_max3
// with the reference's comparison chain only where a NaN shows, expf for arguments <= 0 with its case analysis folded - half
// the vector instructions of hml_emit_block for the same bits (the phase is bound by vector issue: eight chains' blocks).
// More states: the looped form (few registers).
template <int K>
struct hml_fm_params {
    hml_tr2_params<K> fast;   // 2 mu, 1 / (2 var), logN, logA (0 without self-transitions)
    hml_emit_lds<K> plain;    // mu, var, logN, logA, 1 / (2 var)
};
// e_s = expf(E_s - max E) of one block (K <= 6: registers)
template <int K>
__device__ __forceinline__ void hml_fm_terms(const hml_fm_params<K>& l, bool self, hml_model* mdl, float sx, float sq, float N, const uint64_t* exp_tab,
                                             float (&e)[K]) {
    float E[K];
    if (__builtin_expect(hml_tr2_energies<K>(l.fast, self, sx, sq, N, E), 0)) hml_tr2_energies_literal<K>(mdl, mdl, self, sx, sq, N, E);
    // std::max in state order from numeric_limits<float>::lowest() (ForwardBackward.hpp:78-81) is the largest term unless
    // one is a NaN (which the comparison chain lets through and then forgets); v_max3 skips NaNs, and the NaN shows in
    // the sum below, where the chain is then walked literally
    float maxE = -3.40282346638528859812e+38f;
#pragma unroll
    for (int s = 0; s < K; ++s) maxE = __builtin_fmaxf(maxE, E[s]);
    float xs[K], xsum = 0.0f;
#pragma unroll
    for (int s = 0; s < K; ++s) { xs[s] = E[s] - maxE; xsum += xs[s]; }   // every term <= 0 or NaN: the sum is a NaN only if a term is
    if (__builtin_expect(xsum != xsum, 0)) {
        maxE = -3.40282346638528859812e+38f;
#pragma unroll
        for (int s = 0; s < K; ++s) maxE = (E[s] < maxE) ? maxE : E[s];
#pragma unroll
        for (int s = 0; s < K; ++s) e[s] = hml_expf_tab(E[s] - maxE, exp_tab);
    } else {
#pragma unroll
        for (int s = 0; s < K; ++s) e[s] = hml_tr2_expf_nonpos(xs[s], exp_tab);
    }
}
template <int K>
__device__ __forceinline__ void hml_fm_store(const hml_fm_params<K>& l, bool self, uint32_t b, float N, const float (&e)[K], float* __restrict__ em,
                                             float* __restrict__ gsc, const hml_layout lay, const uint64_t* exp_tab) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
        em[hml_bk(lay, b, K, s)] = e[s];
        if (gsc) gsc[hml_bk(lay, b, K, s)] = self ? hml_expf_tab((N - 1.0f) * l.plain.logA[s], exp_tab) : 1.0f;
    }
}
template <int K>
__device__ __forceinline__ void hml_fm_emit(const hml_fm_params<K>& l, bool self, hml_model* mdl, uint32_t b, float sx, float sq, float N,
                                            float* __restrict__ em, float* __restrict__ gsc, const hml_layout lay, const uint64_t* exp_tab) {
    if constexpr (K > 6) {
        hml_emit_block_looped<K>(l.plain, mdl, self, b, sx, sq, N, em, gsc, nullptr, 0, lay, exp_tab);
    } else {
        float e[K];
        hml_fm_terms<K>(l, self, mdl, sx, sq, N, exp_tab, e);
        hml_fm_store<K>(l, self, b, N, e, em, gsc, lay, exp_tab);
    }
}

template <int K>
HML_KERNEL __launch_bounds__(HML_FUSED_WAVES * 64, HML_FM_MIN_WAVES) void hml_m_blocks_fused(const uint8_t* __restrict__ summary, const float* __restrict__ w,
                                                                             const float2* __restrict__ ia, uint32_t T, int32_t base,
                                                                             const hml_fm_args args, int n,
                                                                             uint32_t n_sub, uint32_t spin_limit, unsigned long long* __restrict__ dbg) {
    // dbg (HML_FUSED_DEBUG): wall-clock stamps of the workgroup's phases, 8 words per workgroup (printed by hml_sync)
    if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 8 + 0] = wall_clock64();
    // The loops over the chains are NOT unrolled (a first version was: 21 000 instructions, twice the instruction cache of a
    // CU pair - 99 us for eight chains): per-chain values live in LDS and are indexed by the chain.
    constexpr int NW = HML_FUSED_WAVES;
    constexpr uint32_t NT = NW * 64;
    constexpr int NC = HML_FM_MAX_CHAINS;
    __shared__ uint16_t listed_all[NW][HML_SUM_SPANS * 256];   // per wavefront: opened groups of the current batch (span << 8 | group), position order
    __shared__ uint16_t wave_list[NC][NW][HML_FM_LIST];        // per chain and wavefront: its first starts (offsets into its eighth)
    __shared__ uint32_t wave_total[NC][NW], wave_last[NC][NW]; // block starts per wavefront; 1 + tile-relative position of the last
    __shared__ uint32_t s_before[NC], s_prev_start[NC];
    __shared__ uint32_t s_item0[NC + 1];                       // items of the chains before chain c (one item per start; + the end marker in the last workgroup)
    __shared__ float s_thr[NC];
    __shared__ uint32_t s_gen[NC];
    __shared__ uint64_t sm_exp_tab[32];
    __shared__ hml_fm_params<K> sm_emit[NC];
    __shared__ int s_self[NC];
    __shared__ uint32_t s_cap[NC];   // block capacity of the chain's buffers (hml_state.h)
    __shared__ hml_fm_chain s_ch[NC];
    if (threadIdx.x < 32u) sm_exp_tab[threadIdx.x] = HML_EXP2F_TAB[threadIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the first batch's summary words do not depend on the chains' parameters: requested ahead of them, so that the two round
    // trips overlap (beside another group's kernels a round trip takes three times its idle latency)
    const uint32_t g = blockIdx.x;
    const uint32_t n_spans = (uint32_t)(((uint64_t)T + HML_SPAN - 1) / HML_SPAN);
    const uint32_t eighth = n_sub * (uint32_t)HML_FUSED_WAVE_BATCH;                   // positions per wavefront
    const uint32_t tile_positions = eighth * (uint32_t)NW;
    const uint64_t wave_base = ((uint64_t)g * NW + (uint32_t)wave) * eighth;          // first position of this wavefront's eighth
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
    // chain c's parameters (written by the parameter kernel of the sweep before) by wavefront c, c + 8, ...
    for (int c = wave; c < n; c += NW) {
        const hml_model* m = args.c[c].mdl;
        hml_emit_lds_fill<K>(sm_emit[c].plain, m, lane);
        hml_tr2_params_fill<K>(sm_emit[c].fast, m, lane);
        if (lane == 0) { s_thr[c] = m->thr; s_gen[c] = hml_fused_generation(m); s_self[c] = m->self_trans; s_cap[c] = m->cap; s_ch[c] = args.c[c]; }
    }
    if (lane == 0) for (int c = 0; c < n; ++c) { wave_total[c][wave] = 0u; wave_last[c][wave] = 0u; }
    __syncthreads();
    if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 8 + 5] = wall_clock64();

    uint16_t* listed = listed_all[wave];

    // ---------------- phase A: the starts of every chain in this wavefront's eighth
    {
        // the lowest key any chain's threshold maps to opens a superset of every chain's groups (NaN threshold: key 0, all groups)
        uint32_t kmin = 256u;
        for (int c = 0; c < n; ++c) {
            const float th = s_thr[c];
            const uint32_t k = (th != th) ? 0u : hml_weight_key(th, base);
            kmin = k < kmin ? k : kmin;
        }
        const hml_swar_ge sw_ge = hml_swar_ge_make(kmin);
        for (uint32_t j = 0; j < n_sub; ++j) {   // wave-uniform
            const uint32_t span0 = (uint32_t)(wave_base / HML_SPAN) + j * HML_SUM_SPANS;
            if (span0 >= n_spans) break;
            uint32_t n_listed = 0u;
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
            hml_wave_lds_fence();     // (LDS operations of one wavefront complete in order: the reads below see the writes above)
            for (uint32_t i0 = 0; i0 < n_listed; i0 += 64u) {   // wave-uniform; one pass unless > 64 groups are open
                const uint32_t i = i0 + (uint32_t)lane;
                float wv[16];
                uint32_t valid = 0u, in_eighth = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) wv[r] = 0.0f;
                if (i < n_listed) {
                    const uint32_t sg = listed[i];
                    const uint32_t in_batch = (sg >> 8) * HML_SPAN + (sg & 255u) * 16u;
                    const uint64_t t0 = (uint64_t)span0 * HML_SPAN + in_batch;
                    in_eighth = j * (uint32_t)HML_FUSED_WAVE_BATCH + in_batch;
                    if (t0 + 16u <= T) {
                        const float4* __restrict__ p = reinterpret_cast<const float4*>(w + t0);
                        const float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
                        wv[0] = v0.x; wv[1] = v0.y; wv[2] = v0.z; wv[3] = v0.w; wv[4] = v1.x; wv[5] = v1.y; wv[6] = v1.z; wv[7] = v1.w;
                        wv[8] = v2.x; wv[9] = v2.y; wv[10] = v2.z; wv[11] = v2.w; wv[12] = v3.x; wv[13] = v3.y; wv[14] = v3.z; wv[15] = v3.w;
                        valid = 0xffffu;
                    } else {
                        // the group that straddles T (groups wholly beyond T hold nothing)
#pragma unroll
                        for (uint32_t r = 0; r < 16u; ++r)
                            if (t0 + r < T) { wv[r] = w[t0 + r]; valid |= 1u << r; }
                    }
                }
                const bool origin = (span0 == 0u && i == 0u);   // position 0 (group 0 of span 0 is listed first)
                // The two largest weights of the group decide almost every chain: below the second largest nothing, or
                // exactly the position of the largest, starts a block (at strong compression a group holds at most one
                // start) - two comparisons per chain instead of sixteen.  A NaN anywhere (weight or threshold: !(w < thr)
                // is then true whatever the other is) and a threshold that reaches the second largest weight take the
                // sixteen comparisons.
                float w1 = -HML_INF_F, w2 = -HML_INF_F;
                uint32_t p1 = 0u;
                bool has_nan = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float x = wv[r];
                    has_nan = has_nan || (x != x);
                    const bool gt1 = x > w1;
                    w2 = gt1 ? w1 : ((x > w2) ? x : w2);
                    p1 = gt1 ? (uint32_t)r : p1;
                    w1 = gt1 ? x : w1;
                }
                if (dbg && threadIdx.x == 0 && i0 == 0u && j == 0u) dbg[blockIdx.x * 8 + 6] = wall_clock64() + (w1 > 1e30f ? 1ull : 0ull);   // (behind the weights' arrival)
#pragma unroll 1
                for (int c = 0; c < n; ++c) {
                    const float thr = s_thr[c];
                    uint32_t m16;
                    if (__builtin_expect(has_nan || (thr != thr) || !(w2 < thr), 0)) {
                        m16 = 0u;
#pragma unroll
                        for (int r = 0; r < 16; ++r) m16 |= (uint32_t)!(wv[r] < thr) << r;
                    } else {
                        m16 = (w1 < thr) ? 0u : (1u << p1);   // w2 < thr: every weight but the largest is below the threshold
                    }
                    m16 &= valid;
                    if (origin) m16 |= 1u;
                    const uint32_t cnt = (uint32_t)__popc(m16);
                    const unsigned long long some = __ballot(cnt != 0u);
                    if (some == 0ull) continue;   // wave-uniform: this chain's threshold opens none of these groups
                    const uint32_t placed = wave_total[c][wave];   // (wave-uniform: what lane 0 stored behind the pass before)
                    uint32_t r, sum;
                    if (__ballot(cnt > 1u) == 0ull) {   // wave-uniform: at most one start per group
                        r = placed + hml_mbcnt(some);
                        sum = (uint32_t)__popcll(some);
                    } else {
                        // groups next to a true jump often hold two or three starts: the exclusive prefix of the counts (<= 16) bit
                        // by bit from ballots - no cross-lane traffic through LDS (a shuffle scan is six dependent LDS round trips,
                        // and all wavefronts of the machine are in this loop at the same time)
                        r = placed; sum = 0u;
#pragma unroll
                        for (int bit = 0; bit < 5; ++bit) {
                            const unsigned long long mb = __ballot(((cnt >> bit) & 1u) != 0u);
                            r += hml_mbcnt(mb) << bit;
                            sum += (uint32_t)__popcll(mb) << bit;
                        }
                    }
                    // the last start of this pass: highest set bit of the highest lane that holds one
                    const int src = 63 - __clzll((long long)some);
                    const uint32_t hi = in_eighth + (31u - (uint32_t)__clz((int)(m16 | 1u)));
                    const uint32_t last_off = (uint32_t)__builtin_amdgcn_readlane((int)hi, src);
                    uint16_t* const wl = wave_list[c][wave];
                    uint16_t* const stg = s_ch[c].stage;
                    uint32_t mm = m16;
                    while (mm) {
                        const int bit = __ffs(mm) - 1;
                        mm &= mm - 1u;
                        const uint16_t off = (uint16_t)(in_eighth + (uint32_t)bit);
                        if (r < (uint32_t)HML_FM_LIST) wl[r] = off;
                        else stg[wave_base + r] = off;
                        ++r;
                    }
                    if (lane == 0) { wave_total[c][wave] = placed + sum; wave_last[c][wave] = (uint32_t)wave * eighth + last_off + 1u; }
                    hml_wave_lds_fence();
                }
            }
        }
    }
    if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 8 + 7] = wall_clock64();
    __syncthreads();   // (also makes staged starts visible to the other wavefronts of the workgroup)
    if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 8 + 1] = wall_clock64();
    // the tile's word of every chain, and the chains' item ranges
    const bool last_wg = (g == gridDim.x - 1u);
    if (threadIdx.x < (uint32_t)n) {
        const int c = (int)threadIdx.x;
        uint32_t tot = 0u, l1 = 0u;
#pragma unroll
        for (int k = 0; k < NW; ++k) { tot += wave_total[c][k]; l1 = wave_last[c][k] > l1 ? wave_last[c][k] : l1; }
        __hip_atomic_store(&s_ch[c].group_word[g], hml_group_word(s_gen[c], tot, l1 ? l1 - 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 64u) {
        uint32_t run = 0u;
        for (int c = 0; c < n; ++c) {
            s_item0[c] = run;
            uint32_t tot = 0u;
#pragma unroll
            for (int k = 0; k < NW; ++k) tot += wave_total[c][k];
            run += tot + (last_wg ? 1u : 0u);   // one item per start; the last workgroup has one more: the end marker T, which finishes the final block
        }
        s_item0[n] = run;
    }
    __syncthreads();

    // the k-th start of the workgroup in chain c, as a global position (k < the chain's total); *idx_out: its index in
    // its wavefront's list (> 0: the start before it is the list's entry before it)
    auto start_at2 = [&](int c, uint32_t k, uint32_t& wv_out, uint32_t& idx_out) -> uint32_t {
        uint32_t wv = 0u, first = 0u, run = 0u;
#pragma unroll
        for (int q = 0; q < NW - 1; ++q) {
            run += wave_total[c][q];
            if (k >= run) { wv = (uint32_t)q + 1u; first = run; }
        }
        const uint32_t idx = k - first;
        const uint64_t wb = ((uint64_t)g * NW + wv) * eighth;
        const uint32_t off = (idx < (uint32_t)HML_FM_LIST) ? (uint32_t)wave_list[c][wv][idx] : (uint32_t)s_ch[c].stage[wb + idx];
        wv_out = wv; idx_out = idx;
        return (uint32_t)(wb + off);
    };
    auto start_at = [&](int c, uint32_t k) -> uint32_t { uint32_t a, b; return start_at2(c, k, a, b); };
    // Items: the starts of all chains in one sequence, chain after chain (j -> chain c, k-th start of the workgroup in c): a
    // thread takes items tid, tid + NT, ...; the item with start t finishes the block that ends at t.
    const uint32_t total_items = s_item0[n];
    struct item_t { uint32_t c, k, t, prev, pending; float ax, ay, zx, zy; };   // (32-bit fields only: the copies stay in registers)
    auto locate = [&](uint32_t j, item_t& it) {
        int c = 0;
        for (int q = 1; q < n; ++q) c += (j >= s_item0[q]) ? 1 : 0;
        it.c = (uint32_t)c;
        it.k = j - s_item0[c];
        const uint32_t wg_total = s_item0[c + 1] - s_item0[c] - (last_wg ? 1u : 0u);
        it.pending = 0u;
        // the block's first position: the start before this one - nearly always the entry before it in the same list
        uint32_t wv = 0u, idx = 0u;
        it.t = (it.k < wg_total) ? start_at2(c, it.k, wv, idx) : T;
        if (it.k < wg_total && idx > 0u) {
            const uint64_t wb = ((uint64_t)g * NW + wv) * eighth;
            const uint32_t i1 = idx - 1u;
            it.prev = (uint32_t)(wb + ((i1 < (uint32_t)HML_FM_LIST) ? (uint32_t)wave_list[c][wv][i1] : (uint32_t)s_ch[c].stage[wb + i1]));
        } else {
            it.prev = (it.k > 0u) ? start_at(c, it.k - 1u) : 0xffffffffu;   // (a chain's first item of the workgroup: known behind the offsets)
        }
    };
    // the gathers of an item's block (loads only): possible as soon as the block's first position is known - for every
    // item but a chain's first of the workgroup, whose block begins in an earlier tile (known behind the offsets)
    auto request = [&](item_t& it, bool offsets_known) {
        it.ax = it.ay = it.zx = it.zy = 0.0f;
        if (it.t == 0u) return;                              // no block ends at position 0
        if (it.k == 0u && !offsets_known) { it.pending = 1u; return; }
        if (it.k == 0u) it.prev = s_prev_start[it.c];
        const float2 a = ia[it.prev], z = ia[it.t];
        it.ax = a.x; it.ay = a.y; it.zx = z.x; it.zy = z.y;
    };
    auto finish = [&](const item_t& it) {
        const hml_fm_chain& ch = s_ch[it.c];
        const uint32_t b = s_before[it.c] + it.k;
        if (b > s_cap[it.c]) return;   // beyond the chain's block capacity: the chain is halted where its block count is set
        ch.starts[b] = it.t;   // (the end marker of the last workgroup: starts[B] = T)
        if (it.t == 0u) return;
        float sx, sq;
        bool inside;
        hml_tr2_stats(it.prev, it.t, make_float2(it.ax, it.ay), make_float2(it.zx, it.zy), sx, sq, inside);
        if (inside) hml_block_stats_one(ia, it.prev, it.t, sx, sq);   // a cell boundary of the integral array inside the block
        ch.bstat[b - 1u] = make_float2(sx, sq);
        hml_fm_emit<K>(sm_emit[it.c], s_self[it.c] != 0, ch.mdl, b - 1u, sx, sq, (float)(it.t - it.prev), ch.em, ch.gsc, ch.lay, sm_exp_tab);
    };
    // ---------------- first round: this thread's first two items - their gathers travel during the wait for the offsets
    item_t it0, it1;
    const uint32_t j0 = threadIdx.x, j1 = threadIdx.x + NT;
    it0.t = 0u; it0.k = 0u; it0.c = 0u; it0.pending = 0u; it0.prev = 0u; it0.ax = it0.ay = it0.zx = it0.zy = 0.0f;
    it1 = it0;
    if (j0 < total_items) { locate(j0, it0); request(it0, false); }
    if (j1 < total_items) { locate(j1, it1); request(it1, false); }
    // (Measured and rejected, round 4: COMPUTING the first item's terms during the wait as well - only their stores need the
    // block's index.  The extra live registers spill at six wavefronts per SIMD, 0.201-0.206 against 0.179-0.181 ms per round of
    // eight chains; at four per SIMD it is a wash, 0.1761 against 0.1778.)
    if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 8 + 2] = wall_clock64();

    // ---------------- offsets: per chain the sum of the words of all earlier tiles and the last start before this tile.
    // Wavefront w collects chains w, w + 8, ...: a lane takes tiles lane, lane + 64, ..., four loads in flight at a time.
    static_assert(NC <= NW, "one wavefront per chain");
#pragma unroll 1
    for (int c = wave; c < n; c += NW) {
        const uint32_t gen = s_gen[c];
        unsigned long long* const gword = s_ch[c].group_word;
        uint32_t acc = 0u;
        unsigned long long near = 0ull;   // (1 + tile index) << POS_BITS | last_rel of the last non-empty earlier tile
        for (uint32_t i0 = (uint32_t)lane; i0 < g; i0 += 256u) {
            unsigned long long d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + 64u * (uint32_t)u;
                // (spin_limit 0 is the tests' setting: do not even look - every word is computed here)
                d[u] = (i < g && spin_limit) ? __hip_atomic_load(&gword[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : hml_group_word(gen + 1u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + 64u * (uint32_t)u;
                if (i >= g) break;   // (i grows with u)
                unsigned long long dd = d[u];
                uint32_t tries = 0u;
                while ((uint32_t)(dd >> (2 * HML_FUSED_POS_BITS + 1)) != gen) {
                    if (tries++ >= spin_limit) {
                        // the owner of tile i has not published in time: its word is a function of the weights and the
                        // threshold, so compute it here and publish it for everyone (hml_k_blocks_fused.h, "Progress")
                        dd = hml_fused_tile_word(summary, w, T, s_thr[c], base, i, n_sub, gen);
                        __hip_atomic_store(&gword[i], dd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (s_ch[c].host_words) __hip_atomic_store(s_ch[c].host_words + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        atomicAdd(&s_ch[c].mdl->fused_fallbacks, 1ull);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    dd = __hip_atomic_load(&gword[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const uint32_t tot = (uint32_t)(dd >> HML_FUSED_POS_BITS) & ((2u << HML_FUSED_POS_BITS) - 1u);
                acc += tot;
                if (tot) near = ((unsigned long long)(i + 1u) << HML_FUSED_POS_BITS) | (dd & ((1ull << HML_FUSED_POS_BITS) - 1ull));   // i grows within a lane
            }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            acc += __shfl_xor(acc, m);
            const unsigned long long o = __shfl_xor(near, m);
            near = o > near ? o : near;
        }
        if (lane == 0) {
            s_before[c] = acc;
            // global position of the last start before this tile (tile 0 holds position 0, so it exists for g > 0)
            s_prev_start[c] = near ? (uint32_t)(((near >> HML_FUSED_POS_BITS) - 1ull) * tile_positions + (near & ((1ull << HML_FUSED_POS_BITS) - 1ull))) : 0u;
            // the block count
            if (last_wg) {
                const uint32_t Bn = acc + (s_item0[c + 1] - s_item0[c] - 1u);
                if (Bn > s_cap[c]) hml_halt(s_ch[c].mdl, Bn, s_ch[c].host_words);
                else {
                    s_ch[c].mdl->B = Bn;
                    hml_warmup_for_many_blocks(s_ch[c].mdl, Bn);
                    if (s_ch[c].host_words) __hip_atomic_store(s_ch[c].host_words, Bn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
    __syncthreads();
    if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 8 + 3] = wall_clock64();

    // ---------------- phase B: emission terms and the writes; two items per round, the next round's gathers under way
    if (it0.pending) request(it0, true);
    if (it1.pending) request(it1, true);
#pragma unroll 1
    for (uint32_t j = threadIdx.x; j < total_items; j += 2u * NT) {
        item_t n0, n1;
        n0 = it0; n1 = it1;
        const uint32_t ja = j + 2u * NT, jb = j + 3u * NT;
        const bool ha = ja < total_items, hb = jb < total_items;
        if (ha) { locate(ja, n0); request(n0, true); }
        if (hb) { locate(jb, n1); request(n1, true); }
        finish(it0);
        if (j + NT < total_items) finish(it1);
        it0 = n0; it1 = n1;
    }
    if (dbg) { __syncthreads(); if (threadIdx.x == 0) dbg[blockIdx.x * 8 + 4] = wall_clock64(); }
}

#endif
