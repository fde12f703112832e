// K4+K5+K6a in one launch: block starts from the weight summary, their global order, the block statistics and
// the emission terms - the per-sweep block structure of a dynamic chain without a dependent dispatch in
// between (each of which costs ~4.5 us on this part, more than the work it separates).
#ifndef HML_K_BLOCKS_FUSED_H
#define HML_K_BLOCKS_FUSED_H

#include "hml_k_blocks.h"
#include "hml_k_forward.h"

// ------------------------------------------------------------------------------------------
// One workgroup of 8 wavefronts per 32 spans (2^17 positions), four spans per wavefront as in
// hml_k_compact_scan_summary.
//   phase A  every wavefront lists its opened groups (LDS), opens them and keeps the 16-bit start masks in
//            LDS; the workgroup publishes {launch generation, its number of block starts, the position of its
//            last one} as ONE 64-bit word (relaxed device-scope store).
//   offsets  every workgroup reads the words of ALL workgroups before it (763 at T = 10^8, two loads per thread),
//            spinning on words that do not carry this launch's generation yet: their sum is the index of its
//            first block, the last non-empty one gives the start of the block that ends at its first start.
//            Workgroups only wait for lower-numbered ones, which were dispatched before them, and every
//            workgroup publishes before it waits: no chain, no deadlock.
//   phase B  the workgroup's starts are gathered in LDS in position order and dealt out one per thread, so the
//            lanes are densely occupied: thread k writes starts[first + k] and finishes the block that ENDS
//            there - its statistics come from the integral array (K5), its emission terms follow (K6a).  (A
//            workgroup with more than HML_FUSED_LIST starts - compression below 64 - takes several rounds.)
//            The last workgroup has one more item, the end marker: starts[B] = T, the final block, and B.
// Nothing but the 64-bit words crosses workgroups, so no cache write-back is needed inside the launch.
// Same results as hml_k_compact_scan(+_summary) + hml_k_compact_scatter + hml_k_stats_emission, bit for bit.
// ------------------------------------------------------------------------------------------
#define HML_FUSED_WAVES 8                                                   // wavefronts per workgroup
#define HML_FUSED_POSITIONS (HML_FUSED_WAVES * HML_SUM_SPANS * HML_SPAN)     // positions per workgroup
#define HML_FUSED_POS_BITS 17                                               // log2 of it
#define HML_FUSED_GEN_MASK ((1u << (64 - 2 * HML_FUSED_POS_BITS - 1)) - 1u)
#define HML_FUSED_LIST 2048                                                 // starts a workgroup can gather in LDS
__device__ __forceinline__ unsigned long long hml_group_word(uint32_t gen, uint32_t total, uint32_t last_rel) {
    // gen: 29 bits | total: POS_BITS + 1 bits (0..2^POS_BITS) | position of the last start in the workgroup's range: POS_BITS bits
    return ((unsigned long long)(gen & HML_FUSED_GEN_MASK) << (2 * HML_FUSED_POS_BITS + 1)) |
           ((unsigned long long)total << HML_FUSED_POS_BITS) | (unsigned long long)last_rel;
}

template <int K>
__global__ __launch_bounds__(HML_FUSED_WAVES * 64, (K <= 6 ? 6 : 4)) void hml_k_blocks_fused(const uint8_t* __restrict__ summary, const float* __restrict__ w,
                                                          const float2* __restrict__ ia, uint32_t T,
                                                          hml_model* __restrict__ mdl, int32_t base,
                                                          unsigned long long* __restrict__ group_word,
                                                          uint32_t* __restrict__ launch_gen, uint32_t* __restrict__ starts,
                                                          float2* __restrict__ bstat, float* __restrict__ em,
                                                          float* __restrict__ gsc, float* __restrict__ eprobe, int mixture,
                                                          const hml_layout lay, uint32_t* __restrict__ host_B, unsigned long long* __restrict__ dbg) {
    if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 4 + 0] = wall_clock64();
    static_assert(HML_FUSED_POSITIONS == (1 << HML_FUSED_POS_BITS), "workgroup geometry (bit fields of hml_group_word)");
    constexpr int NW = HML_FUSED_WAVES;
    __shared__ uint16_t listed_all[NW][HML_SUM_SPANS * 256];   // per wavefront: opened groups (span << 8 | group), position order
    __shared__ uint16_t mask_all[NW][HML_SUM_SPANS * 256];     // their start masks
    __shared__ uint32_t wave_total[NW], wave_last[NW];         // block starts per wavefront; 1 + relative position of the last
    __shared__ unsigned long long red_sum[NW], red_near[NW];
    __shared__ uint32_t start_list[HML_FUSED_LIST];            // the workgroup's starts (relative positions), in order

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x;
    const uint32_t n_spans = (uint32_t)(((uint64_t)T + HML_SPAN - 1) / HML_SPAN);
    const uint32_t span0 = (g * (uint32_t)NW + (uint32_t)wave) * HML_SUM_SPANS;
    const uint32_t gen = *launch_gen + 1u;
    uint16_t* listed = listed_all[wave];
    uint16_t* masks = mask_all[wave];
    const float thr = mdl->thr;

    // ---------------- phase A
    uint32_t n_listed = 0u, total = 0u, last1 = 0u;
    if (span0 < n_spans) {   // wave-uniform
        uint32_t gw[HML_SUM_SPANS];
#pragma unroll
        for (int s = 0; s < HML_SUM_SPANS; ++s)
            gw[s] = (span0 + s < n_spans)
                        ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(summary) + (uint64_t)(span0 + s) * 64u + lane)
                        : 0u;
        // NaN threshold: !(w < thr) holds everywhere, every position starts a block; key 0 opens every group
        const uint32_t kthr = (thr != thr) ? 0u : hml_weight_key(thr, base);
        const hml_swar_ge sw_ge = hml_swar_ge_make(kthr);
#pragma unroll
        for (int s = 0; s < HML_SUM_SPANS; ++s) {
            const uint32_t fl = (span0 + s < n_spans) ? hml_swar_ge_apply(sw_ge, gw[s]) : 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // group 0 of span 0 is always opened: position 0 starts a block whatever its weight
                const bool open = ((fl >> (8 * j + 7)) & 1u) || (span0 + s == 0u && j == 0 && lane == 0);
                const unsigned long long m = __ballot(open);
                if (open) listed[n_listed + hml_mbcnt(m)] = (uint16_t)((s << 8) | (64 * j + lane));
                n_listed += (uint32_t)__popcll(m);
            }
        }
        // (LDS operations of one wavefront complete in order: the reads below see the writes above)
        for (uint32_t i0 = 0; i0 < n_listed; i0 += 64u) {   // wave-uniform; one pass unless > 64 groups are open
            const uint32_t i = i0 + (uint32_t)lane;
            uint32_t m16 = 0u, rel1 = 0u;
            if (i < n_listed) {
                const uint32_t sg = listed[i];
                const uint32_t in_wave = (sg >> 8) * HML_SPAN + (sg & 255u) * 16u;   // position relative to the wavefront's first
                m16 = hml_group_mask16(w, (uint64_t)span0 * HML_SPAN + in_wave, T, thr);
                if (span0 == 0u && i == 0u) m16 |= 1u;   // position 0 (group 0 of span 0 is listed first)
                masks[i] = (uint16_t)m16;
                if (m16) rel1 = (uint32_t)wave * (HML_SUM_SPANS * HML_SPAN) + in_wave + (31u - (uint32_t)__clz((int)m16)) + 1u;
            }
            total += (uint32_t)__popc(m16);
            last1 = rel1 > last1 ? rel1 : last1;
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
    for (int k = 0; k < NW; ++k) { wg_total += wave_total[k]; wg_last1 = wave_last[k] > wg_last1 ? wave_last[k] : wg_last1; }
    if (threadIdx.x == 0) {
        const uint32_t l1 = wg_last1;
        __hip_atomic_store(&group_word[g], hml_group_word(gen, wg_total, l1 ? l1 - 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---------------- phase B, the part that needs no global offset: gather the starts, gather the statistics
    const uint32_t group_base = g * (uint32_t)HML_FUSED_POSITIONS;
    uint32_t wave_off = 0u;               // block starts of the workgroup before this wavefront
#pragma unroll
    for (int k = 0; k < NW; ++k)
        if (k < wave) wave_off += wave_total[k];
    const uint32_t wave_rel = (uint32_t)wave * (HML_SUM_SPANS * HML_SPAN);
    const bool last_wg = (g == gridDim.x - 1u);
    // items: one per start (write it, finish the block that ends there); the last workgroup has one more, the
    // end marker T, which finishes the final block
    const uint32_t n_items = wg_total + (last_wg ? 1u : 0u);
    // the starts of ranks [lo, lo + HML_FUSED_LIST) into LDS, in position order
    auto gather = [&](uint32_t lo) {
        if (span0 >= n_spans) return;   // wave-uniform
        uint32_t r0 = wave_off;
        for (uint32_t i0 = 0; i0 < n_listed && r0 < lo + (uint32_t)HML_FUSED_LIST; i0 += 64u) {
            const uint32_t i = i0 + (uint32_t)lane;
            uint32_t m16 = 0u, first_rel = 0u;
            if (i < n_listed) {
                const uint32_t sg = listed[i];
                m16 = masks[i];
                first_rel = wave_rel + (sg >> 8) * HML_SPAN + (sg & 255u) * 16u;
            }
            const uint32_t c = (uint32_t)__popc(m16);
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            uint32_t r = r0 + incl - c;
            while (m16) {
                const int bit = __ffs(m16) - 1;
                m16 &= m16 - 1u;
                if (r - lo < (uint32_t)HML_FUSED_LIST) start_list[r - lo] = first_rel + (uint32_t)bit;   // r < lo wraps to a large value
                ++r;
            }
            r0 += __shfl(incl, 63);
        }
    };
    gather(0u);
    __syncthreads();
    // every thread's first item but the workgroup's very first (whose block begins in an earlier workgroup):
    // its statistics only need positions, so their gathers overlap the wait for the offsets below
    const uint32_t k_first = threadIdx.x;
    bool have_first = false;
    float first_sx = 0.0f, first_sq = 0.0f;
    if (k_first > 0u && k_first < n_items && k_first < (uint32_t)HML_FUSED_LIST) {
        const uint32_t t = (k_first < wg_total) ? group_base + start_list[k_first] : T;
        hml_block_stats_one(ia, group_base + start_list[k_first - 1u], t, first_sx, first_sq);
        have_first = true;
    }

    // ---------------- offsets: sum over all earlier groups, and the last start before this group
    unsigned long long acc = 0ull, near = 0ull;   // near: (1 + workgroup index) << POS_BITS | last_rel of the last non-empty earlier one
    for (uint32_t i = threadIdx.x; i < g; i += (uint32_t)(NW * 64)) {
        unsigned long long d = __hip_atomic_load(&group_word[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((uint32_t)(d >> (2 * HML_FUSED_POS_BITS + 1)) != (gen & HML_FUSED_GEN_MASK)) {
            __builtin_amdgcn_s_sleep(1);
            d = __hip_atomic_load(&group_word[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint32_t tot = (uint32_t)(d >> HML_FUSED_POS_BITS) & ((2u << HML_FUSED_POS_BITS) - 1u);
        acc += tot;
        if (tot) near = ((unsigned long long)(i + 1u) << HML_FUSED_POS_BITS) | (d & ((1ull << HML_FUSED_POS_BITS) - 1ull));   // i grows within a thread
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
    // global position of the last start before this workgroup (workgroup 0 holds position 0, so it exists for g > 0)
    const uint32_t prev_group_start =
        nr ? ((uint32_t)(nr >> HML_FUSED_POS_BITS) - 1u) * (uint32_t)HML_FUSED_POSITIONS + (uint32_t)(nr & ((1ull << HML_FUSED_POS_BITS) - 1ull)) : 0u;

    // ---------------- phase B, the rest: emission terms and the writes
    hml_emit_params<K> p;
    hml_emit_load<K>(p, mdl, mixture);
    uint32_t carry = prev_group_start;    // the start before the first one of the current round
    for (uint32_t lo = 0; lo < n_items; lo += (uint32_t)HML_FUSED_LIST) {   // one round unless > HML_FUSED_LIST starts
        if (lo) {
            gather(lo);
            __syncthreads();
        }
        const uint32_t hi = (lo + (uint32_t)HML_FUSED_LIST < n_items) ? lo + (uint32_t)HML_FUSED_LIST : n_items;
        for (uint32_t k = lo + threadIdx.x; k < hi; k += (uint32_t)(NW * 64)) {
            const uint32_t t = (k < wg_total) ? group_base + start_list[k - lo] : T;
            const uint32_t b = before_group + k;
            starts[b] = t;   // (item wg_total of the last workgroup: starts[B] = T)
            if (t == 0u) continue;   // no block ends at position 0
            float sx = first_sx, sq = first_sq;
            if (!(have_first && k == k_first)) {
                const uint32_t prev_t = (k > lo) ? group_base + start_list[k - 1u - lo] : carry;
                hml_block_stats_one(ia, prev_t, t, sx, sq);
            }
            const uint32_t prev_for_n = (k > lo) ? group_base + start_list[k - 1u - lo] : carry;
            bstat[b - 1u] = make_float2(sx, sq);
            hml_emit_block<K>(p, mdl, b - 1u, sx, sq, (float)(t - prev_for_n), em, gsc, eprobe, mixture, lay);
        }
        if (hi - lo == (uint32_t)HML_FUSED_LIST) carry = group_base + start_list[HML_FUSED_LIST - 1];
        __syncthreads();   // the list is rewritten by the next round
    }
    if (dbg) { __syncthreads(); if (threadIdx.x == 0) dbg[blockIdx.x * 4 + 3] = wall_clock64(); }
    // the block count
    if (last_wg && threadIdx.x == 0) {
        const uint32_t Bn = before_group + wg_total;
        mdl->B = Bn;
        hml_warmup_for_many_blocks(mdl, Bn);
        *launch_gen = gen;   // every workgroup has read it: this one only got here after all of them published
        // host-mapped word: lets the host size later grids without a copy in the stream
        if (host_B) __hip_atomic_store(host_B, Bn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#endif
