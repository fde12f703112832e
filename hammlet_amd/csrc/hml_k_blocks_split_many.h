// K4 | K5+K6a for several chains over one shared trace in TWO launches (round 5) - the many-chain block kernel of
// hml_k_blocks_fused_many.h cut where its workgroups wait for each other.
//
// The fused kernel's workgroups hand the number of their block starts from tile to tile inside ONE launch, so its whole grid
// must be resident, and every workgroup keeps its slots through three dependent memory round trips and the hand-off: a launch
// for eight chains holds the machine for 57-66 us of which a workgroup computes for ~30 (profiles/round4_chains_attached.txt:
// lists done at 17 us, offsets known at 30-47, end at 46-65).  For ONE chain that is the fastest form - a launch boundary
// costs as much as the work it separates.  For MANY chains the machine is the bottleneck (a chain and sweep cost 15.5 us of the
// whole GPU at saturation, the block kernel 8.5 of them), and slots that wait are throughput lost.  Here
//   hml_m_blocks_list      (the number of states does not enter): per tile and chain the block starts - 16-bit offsets into the
//                          wavefront's part of the tile, all of them in the chain's staging array - their count per wavefront and
//                          the tile's word {generation, starts, last start}; up to SIXTEEN chains share one pass over the summary
//                          and the opened groups' weights;
//   hml_m_blocks_offsets   one workgroup per chain: the exclusive prefix of the tiles' words - blocks before every tile, the last start
//                          before it - and the chain's block count (or its halt, hml_state.h);
//   hml_m_blocks_emit<K>   per tile the items - one per block - of all chains in one sequence, FOUR per thread and round: their
//                          starts (staging array) requested together, then their integral-array gathers together, then
//                          statistics, emission terms, stores.  A workgroup's life is five memory round trips however many
//                          chains it serves (the fused kernel: the hand-off, then two dependent round trips per pair of items).
// No workgroup waits for another, no grid has to be resident, tiles are always 2^17 positions.  Per chain the same block
// starts, statistics and terms as hml_m_blocks_fused / hml_k_blocks_fused, bit for bit (tests: test_gpu_parity.py
// test_attached_chains_batched..., test_gpu_fuzz.py test_bounded_fuzz_of_batched_chains; HML_FM_SPLIT=0 takes the fused kernel).
// Reference: Blocks<BreakpointArray>::next src/Blocks/BreakpointArray.hpp:216-235, addBlockStats
// src/Statistics/IntegralArray.hpp:104-124, emission terms src/StateSequence/ForwardBackward.hpp:67-84.
#ifndef HML_K_BLOCKS_SPLIT_MANY_H
#define HML_K_BLOCKS_SPLIT_MANY_H

#include "hml_k_blocks_fused_many.h"

#define HML_FS_MAX_CHAINS 16     // chains per launch

struct hml_fs_chain {
    hml_model* mdl;
    unsigned long long* group_word;   // [tiles] {generation, starts, last start} (hml_group_word)
    uint32_t* wave_total;             // [tiles * 8] starts per wavefront of a tile
    uint32_t* tile_before;            // [tiles] block starts in the tiles before (hml_m_blocks_offsets)
    uint32_t* tile_prev;              // [tiles] position of the last start before the tile
    uint16_t* stage;                  // [T] the starts of wavefront v of tile g: offsets into its part, from (g * 8 + v) * part on
    uint32_t* starts;
    float2* bstat;
    float* em;
    float* gsc;                       // nullptr: no plane of rescale factors (late_rescale)
    uint32_t* host_words;
    hml_layout lay;
};
struct hml_fs_args { hml_fs_chain c[HML_FS_MAX_CHAINS]; };
static_assert(sizeof(hml_fs_args) + 64 <= 4096, "kernel arguments");

HML_KERNEL __launch_bounds__(HML_FUSED_WAVES * 64) void hml_m_blocks_list(const uint8_t* __restrict__ summary, const float* __restrict__ w, uint32_t T,
                                                                           int32_t base, const hml_fs_args args, int n, uint32_t n_sub) {
    constexpr int NW = HML_FUSED_WAVES;
    constexpr int NC = HML_FS_MAX_CHAINS;
    __shared__ uint16_t listed_all[NW][HML_SUM_SPANS * 256];   // per wavefront: opened groups of the current batch (span << 8 | group), position order
    __shared__ uint32_t wave_total[NC][NW], wave_last[NC][NW]; // block starts per wavefront; 1 + tile-relative position of the last
    __shared__ float s_thr[NC];
    __shared__ uint32_t s_gen[NC];
    __shared__ uint16_t* s_stage[NC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x;
    const uint32_t n_spans = (uint32_t)(((uint64_t)T + HML_SPAN - 1) / HML_SPAN);
    const uint32_t eighth = n_sub * (uint32_t)HML_FUSED_WAVE_BATCH;                   // positions per wavefront
    const uint64_t wave_base = ((uint64_t)g * NW + (uint32_t)wave) * eighth;          // first position of this wavefront's part
    auto load_batch = [&](uint32_t j, uint32_t (&gw)[HML_SUM_SPANS]) {
        const uint32_t span0 = (uint32_t)(wave_base / HML_SPAN) + j * HML_SUM_SPANS;
#pragma unroll
        for (int s = 0; s < HML_SUM_SPANS; ++s)
            gw[s] = (j < n_sub && span0 + s < n_spans)
                        ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(summary) + (uint64_t)(span0 + s) * 64u + lane)
                        : 0u;
    };
    uint32_t gw[HML_SUM_SPANS];
    load_batch(0u, gw);   // (does not depend on the chains' thresholds: requested ahead of them)
    if (threadIdx.x < (uint32_t)n) {
        const int c = (int)threadIdx.x;
        const hml_model* m = args.c[c].mdl;
        s_thr[c] = m->thr; s_gen[c] = hml_fused_generation(m); s_stage[c] = args.c[c].stage;
    }
    if (lane == 0) for (int c = 0; c < n; ++c) { wave_total[c][wave] = 0u; wave_last[c][wave] = 0u; }
    __syncthreads();
    uint16_t* listed = listed_all[wave];
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
            hml_wave_lds_fence();
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
                // the two largest weights of the group decide almost every chain (hml_m_blocks_fused): two comparisons per chain
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
                    uint16_t* const stg = s_stage[c];
                    uint32_t mm = m16;
                    while (mm) {
                        const int bit = __ffs(mm) - 1;
                        mm &= mm - 1u;
                        stg[wave_base + r] = (uint16_t)(in_eighth + (uint32_t)bit);
                        ++r;
                    }
                    if (lane == 0) { wave_total[c][wave] = placed + sum; wave_last[c][wave] = (uint32_t)wave * eighth + last_off + 1u; }
                    hml_wave_lds_fence();
                }
            }
        }
    }
    __syncthreads();
    // per chain: the starts of every wavefront and the tile's word
    for (int i = (int)threadIdx.x; i < n * NW; i += NW * 64) {
        const int c = i / NW, k = i - c * NW;
        args.c[c].wave_total[(uint64_t)g * NW + k] = wave_total[c][k];
    }
    if (threadIdx.x >= 64u && threadIdx.x < 64u + (uint32_t)n) {
        const int c = (int)threadIdx.x - 64;
        uint32_t tot = 0u, l1 = 0u;
#pragma unroll
        for (int k = 0; k < NW; ++k) { tot += wave_total[c][k]; l1 = wave_last[c][k] > l1 ? wave_last[c][k] : l1; }
        args.c[c].group_word[g] = hml_group_word(s_gen[c], tot, l1 ? l1 - 1u : 0u);
    }
}

// One workgroup per chain (blockIdx.y): tile_before[g] = block starts in tiles 0 .. g - 1, tile_prev[g] = position of the last start
// before tile g, and the chain's block count - or its halt when the blocks outgrow its buffers (hml_state.h).
HML_KERNEL __launch_bounds__(1024) void hml_m_blocks_offsets(const hml_fs_args args, uint32_t n_tiles, uint32_t n_sub) {
    __shared__ uint32_t w_sum[16];
    __shared__ unsigned long long w_near[16];
    const hml_fs_chain& ch = args.c[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile_positions = n_sub * (uint32_t)HML_FUSED_SUB_POSITIONS;
    const uint32_t per = (n_tiles + 1023u) / 1024u;
    const uint32_t a = (uint32_t)tid * per < n_tiles ? (uint32_t)tid * per : n_tiles;
    const uint32_t b = (a + per < n_tiles) ? a + per : n_tiles;
    // this thread's tiles: their starts, and (1 + tile) << POS_BITS | last start's offset of the last non-empty one
    uint32_t sum = 0u;
    unsigned long long near = 0ull;
    for (uint32_t g = a; g < b; ++g) {
        const unsigned long long d = ch.group_word[g];
        const uint32_t tot = (uint32_t)(d >> HML_FUSED_POS_BITS) & ((2u << HML_FUSED_POS_BITS) - 1u);
        sum += tot;
        if (tot) near = ((unsigned long long)(g + 1u) << HML_FUSED_POS_BITS) | (d & ((1ull << HML_FUSED_POS_BITS) - 1ull));
    }
    // exclusive scan over the threads: sums add, `near` takes the later one (keys grow with the tile)
    uint32_t isum = sum;
    unsigned long long inear = near;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t os = __shfl_up(isum, d);
        const unsigned long long on = __shfl_up(inear, d);
        if (lane >= d) { isum += os; inear = on > inear ? on : inear; }
    }
    if (lane == 63) { w_sum[wave] = isum; w_near[wave] = inear; }
    __syncthreads();
    uint32_t before = isum - sum;
    unsigned long long prev = __shfl_up(inear, 1);
    if (lane == 0) prev = 0ull;
    for (int wv = 0; wv < wave; ++wv) { before += w_sum[wv]; prev = w_near[wv] > prev ? w_near[wv] : prev; }
    for (uint32_t g = a; g < b; ++g) {
        const unsigned long long d = ch.group_word[g];
        const uint32_t tot = (uint32_t)(d >> HML_FUSED_POS_BITS) & ((2u << HML_FUSED_POS_BITS) - 1u);
        ch.tile_before[g] = before;
        ch.tile_prev[g] = prev ? (uint32_t)(((prev >> HML_FUSED_POS_BITS) - 1ull) * tile_positions + (prev & ((1ull << HML_FUSED_POS_BITS) - 1ull))) : 0u;
        before += tot;
        if (tot) prev = ((unsigned long long)(g + 1u) << HML_FUSED_POS_BITS) | (d & ((1ull << HML_FUSED_POS_BITS) - 1ull));
    }
    if (tid == 1023) {   // (the last thread's range ends at the last tile: `before` is the number of starts = blocks)
        const uint32_t Bn = before;
        const uint32_t cap = ch.mdl->cap;
        if (Bn > cap) hml_halt(ch.mdl, Bn, ch.host_words);
        else {
            ch.mdl->B = Bn;
            hml_warmup_for_many_blocks(ch.mdl, Bn);
            if (ch.host_words) __hip_atomic_store(ch.host_words, Bn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int K>
HML_KERNEL __launch_bounds__(HML_FUSED_WAVES * 64, HML_FM_MIN_WAVES) void hml_m_blocks_emit(const float2* __restrict__ ia, uint32_t T, const hml_fs_args args, int n,
                                                                                            uint32_t n_sub) {
    constexpr int NW = HML_FUSED_WAVES;
    constexpr uint32_t NT = NW * 64;
    constexpr int NC = HML_FS_MAX_CHAINS;
    __shared__ uint32_t wave_total[NC][NW];
    __shared__ uint32_t s_before[NC], s_prev_start[NC];
    __shared__ uint32_t s_item0[NC + 1];                       // items of the chains before chain c (one item per start; + the end marker in the last workgroup)
    __shared__ uint64_t sm_exp_tab[32];
    __shared__ hml_fm_params<K> sm_emit[NC];
    __shared__ int s_self[NC];
    __shared__ uint32_t s_cap[NC];   // block capacity of the chain's buffers (hml_state.h)
    __shared__ hml_fs_chain s_ch[NC];
    if (threadIdx.x < 32u) sm_exp_tab[threadIdx.x] = HML_EXP2F_TAB[threadIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x;
    const uint32_t eighth = n_sub * (uint32_t)HML_FUSED_WAVE_BATCH;
    const uint32_t tile_positions = eighth * (uint32_t)NW;
    // chain c's parameters (written by the parameter kernel of the sweep before) by wavefront c, c + 8, ...
    for (int c = wave; c < n; c += NW) {
        const hml_model* m = args.c[c].mdl;
        hml_emit_lds_fill<K>(sm_emit[c].plain, m, lane);
        hml_tr2_params_fill<K>(sm_emit[c].fast, m, lane);
        if (lane == 0) {
            s_self[c] = m->self_trans; s_cap[c] = m->cap; s_ch[c] = args.c[c];
            s_before[c] = args.c[c].tile_before[g]; s_prev_start[c] = args.c[c].tile_prev[g];
        }
        if (lane < NW) wave_total[c][lane] = args.c[c].wave_total[(uint64_t)g * NW + lane];
    }
    __syncthreads();
    const bool last_wg = (g == gridDim.x - 1u);
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
    // the k-th start of the workgroup in chain c, as a global position (k < the chain's total); idx_out: its index in its wavefront's list
    auto start_at2 = [&](int c, uint32_t k, uint32_t& wv_out, uint32_t& idx_out) -> uint32_t {
        uint32_t wv = 0u, first = 0u, run = 0u;
#pragma unroll
        for (int q = 0; q < NW - 1; ++q) {
            run += wave_total[c][q];
            if (k >= run) { wv = (uint32_t)q + 1u; first = run; }
        }
        const uint32_t idx = k - first;
        const uint64_t wb = ((uint64_t)g * NW + wv) * eighth;
        wv_out = wv; idx_out = idx;
        return (uint32_t)(wb + (uint32_t)s_ch[c].stage[wb + idx]);
    };
    const uint32_t total_items = s_item0[n];
    struct item_t { uint32_t c, k, t, prev; float ax, ay, zx, zy; };
    // item j -> chain c, the k-th start t of the workgroup in c, and the start before it (the block [prev, t) ends at t)
    auto locate = [&](uint32_t j, item_t& it) {
        int c = 0;
        for (int q = 1; q < n; ++q) c += (j >= s_item0[q]) ? 1 : 0;
        it.c = (uint32_t)c;
        it.k = j - s_item0[c];
        const uint32_t wg_total = s_item0[c + 1] - s_item0[c] - (last_wg ? 1u : 0u);
        uint32_t wv = 0u, idx = 0u;
        it.t = (it.k < wg_total) ? start_at2(c, it.k, wv, idx) : T;
        if (it.k == 0u) it.prev = s_prev_start[c];
        else if (it.k < wg_total && idx > 0u) {
            const uint64_t wb = ((uint64_t)g * NW + wv) * eighth;
            it.prev = (uint32_t)(wb + (uint32_t)s_ch[c].stage[wb + idx - 1u]);
        } else {
            uint32_t a, b;
            it.prev = start_at2(c, it.k - 1u, a, b);
        }
    };
    auto request = [&](item_t& it) {   // the gathers of an item's block (loads only)
        it.ax = it.ay = it.zx = it.zy = 0.0f;
        if (it.t == 0u) return;                              // no block ends at position 0
        const float2 a = ia[it.prev], z = ia[it.t];
        it.ax = a.x; it.ay = a.y; it.zx = z.x; it.zy = z.y;
    };
    auto finish = [&](const item_t& it) {
        const hml_fs_chain& ch = s_ch[it.c];
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
    // ---------------- items: four per thread and round - locate all (their staging loads travel together), gather all, finish all
    constexpr int NI = 4;
#pragma unroll 1
    for (uint32_t j0 = threadIdx.x; j0 < total_items; j0 += (uint32_t)NI * NT) {
        item_t it[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const uint32_t j = j0 + (uint32_t)u * NT;
            it[u].t = 0u; it[u].k = 0u; it[u].c = 0u; it[u].prev = 0u;
            if (j < total_items) locate(j, it[u]);
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            it[u].ax = it[u].ay = it[u].zx = it[u].zy = 0.0f;
            if (j0 + (uint32_t)u * NT < total_items) request(it[u]);
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) if (j0 + (uint32_t)u * NT < total_items) finish(it[u]);
    }
}

#endif
