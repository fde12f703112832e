// Per-sweep block structure, the separate-launch form: threshold scan (float stream or group summary) +
// compaction, and block statistics.  Dynamic sweeps normally use hml_k_blocks_fused.h, which shares the summary
// scan's building blocks from this file; these kernels serve explicit thresholds (hml_create_blocks, the auto
// prior), weakly compressed sweeps and weight_keys = 0.
#ifndef HML_K_BLOCKS_H
#define HML_K_BLOCKS_H

#include "hml_state.h"

// ------------------------------------------------------------------------------------------
// K4 blocks_compact - Blocks<BreakpointArray>::next (reference src/Blocks/BreakpointArray.hpp:216-235).
//
// The reference walks uint16 "next larger weight" pointers; the set of block starts it visits is
// exactly {0} u {t >= 1 : !(w[t] < thr)}.  On the GPU that is a flat scan of w[0..T) - 4 bytes
// per position, nothing else - followed by an ordered compaction:
//   (a) scan:    one wavefront per span of 4096 positions, 16 x (64 lanes x float4) coalesced loads
//                issued up-front; flagged positions are written, in order, as 16-bit offsets into
//                the span's slot of a staging array; the span's count goes to span_count[].
//   (b) scatter: one workgroup per group of 16 spans derives the group's global offset from the totals of
//                the groups before it (summed once per workgroup) plus the counts of the spans before each
//                span inside the group - a handful of coalesced reads instead of a serial scan launch -
//                and turns the staged offsets into starts[], starts[B] = T.
//   A decoupled look-back single-pass version was measured and rejected: with ~8000 one-span tiles in
//   flight its first generation serialises ~128 look-back windows (318 us vs 64 us for the scan alone).
// Nothing but flagged positions is ever written, so at typical compression (1 start per ~500
// positions) the kernel's traffic is the 4*T bytes of w.
// ------------------------------------------------------------------------------------------

typedef float hml_f4 __attribute__((ext_vector_type(4)));

HML_KERNEL __launch_bounds__(256) void hml_k_compact_scan(const float* __restrict__ w, uint32_t T,
                                                          const hml_model* __restrict__ mdl, float thr_override,
                                                          int use_override, uint16_t* __restrict__ stage,
                                                          uint32_t* __restrict__ span_count) {
    // use_override: 0 = the model's threshold, 1 = thr_override
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t span = blockIdx.x * 4u + (uint32_t)wave;
    const uint64_t base = (uint64_t)span * HML_SPAN;
    if (base >= T) return;
    const float thr = use_override ? thr_override : mdl->thr;
    uint32_t running = 0;
    uint16_t* __restrict__ out = stage + base;
    if (base + HML_SPAN <= T) {
        // full span: 16 independent 16-byte loads per lane
        hml_f4 v[16];
        const hml_f4* __restrict__ p = reinterpret_cast<const hml_f4*>(w + base) + lane;
#pragma unroll
        for (int it = 0; it < 16; ++it) v[it] = __builtin_nontemporal_load(p + it * 64);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            bool f0 = !(v[it].x < thr), f1 = !(v[it].y < thr), f2 = !(v[it].z < thr), f3 = !(v[it].w < thr);
            if (span == 0 && it == 0 && lane == 0) f0 = true;   // position 0 always starts a block
            const bool any = f0 | f1 | f2 | f3;
            if (__ballot(any) == 0ull) continue;   // wave-uniform
            const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1), m2 = __ballot(f2), m3 = __ballot(f3);
            const unsigned long long lt = (1ull << lane) - 1ull;
            uint32_t pos = running + __popcll(m0 & lt) + __popcll(m1 & lt) + __popcll(m2 & lt) + __popcll(m3 & lt);
            const uint32_t off = (uint32_t)it * 256u + (uint32_t)lane * 4u;
            if (f0) out[pos++] = (uint16_t)(off);
            if (f1) out[pos++] = (uint16_t)(off + 1);
            if (f2) out[pos++] = (uint16_t)(off + 2);
            if (f3) out[pos++] = (uint16_t)(off + 3);
            running += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
        }
    } else {
        // tail span: guarded scalar loads
        for (int it = 0; it < 16; ++it) {
            bool f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t t = base + (uint64_t)it * 256u + (uint64_t)lane * 4u + j;
                f[j] = (t < T) ? (t == 0 || !(w[t] < thr)) : false;
            }
            const unsigned long long m0 = __ballot(f[0]), m1 = __ballot(f[1]), m2 = __ballot(f[2]), m3 = __ballot(f[3]);
            if ((m0 | m1 | m2 | m3) == 0ull) continue;
            const unsigned long long lt = (1ull << lane) - 1ull;
            uint32_t pos = running + __popcll(m0 & lt) + __popcll(m1 & lt) + __popcll(m2 & lt) + __popcll(m3 & lt);
            const uint32_t off = (uint32_t)it * 256u + (uint32_t)lane * 4u;
            if (f[0]) out[pos++] = (uint16_t)(off);
            if (f[1]) out[pos++] = (uint16_t)(off + 1);
            if (f[2]) out[pos++] = (uint16_t)(off + 2);
            if (f[3]) out[pos++] = (uint16_t)(off + 3);
            running += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
        }
    }
    if (lane == 0) span_count[span] = running;
}

// ------------------------------------------------------------------------------------------
// K4 over a group summary.  The scan only has to decide !(w[t] < thr), and almost every weight is far below
// the threshold.  key(w) is a MONOTONE 8-bit code
//   key(w) = 255 for NaN, 0 for negative w, else clamp((bits(w) >> 20) - base, 0, 255)   (1/8-binade buckets)
// and the summary holds, per group of 16 consecutive positions, the largest key of the group (built once per
// weight array, T/16 bytes).  Monotonicity gives: summary < key(thr) => every weight of the group is < thr,
// so the group holds no block start and its weights are never read.  A group whose summary reaches key(thr)
// is opened: its 16 float weights (one 64-byte line) are compared exactly.  Same block structure as the
// float scan, bit for bit, from T/16 + 64 * (opened groups) bytes instead of 4 * T.
// ------------------------------------------------------------------------------------------
HML_HD uint32_t hml_weight_key(float w, int32_t base) {
    const uint32_t u = hml_f2u(w);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 255u;   // NaN of either sign: !(w < thr) holds for every thr
    if (u & 0x80000000u) return 0u;
    const int32_t k = (int32_t)(u >> 20) - base;
    return k < 0 ? 0u : (k > 255 ? 255u : (uint32_t)k);
}

// one thread per group: four float4 loads, one byte out
HML_KERNEL __launch_bounds__(256) void hml_k_build_summary(const float* __restrict__ w, uint64_t T, int32_t base,
                                                           uint8_t* __restrict__ summary) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n16 = (T + 15) / 16;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        uint32_t top = 0u;
        if (i * 16 + 16 <= T) {
            const float4* __restrict__ p = reinterpret_cast<const float4*>(w + i * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 v = p[r];
                const uint32_t k0 = hml_weight_key(v.x, base), k1 = hml_weight_key(v.y, base),
                               k2 = hml_weight_key(v.z, base), k3 = hml_weight_key(v.w, base);
                const uint32_t m01 = k0 > k1 ? k0 : k1, m23 = k2 > k3 ? k2 : k3;
                const uint32_t m = m01 > m23 ? m01 : m23;
                top = m > top ? m : top;
            }
        } else {
            for (uint64_t t = i * 16; t < T; ++t) {
                const uint32_t k = hml_weight_key(w[t], base);
                top = k > top ? k : top;
            }
        }
        // within a span, byte j of word l holds group 64*j + l: the scan's lane l then sees groups l, 64+l,
        // 128+l, 192+l, and a ballot over byte j is a bit mask of groups 64*j .. 64*j+63 in position order
        const uint64_t sp = i >> 8;
        const uint32_t g = (uint32_t)(i & 255u);
        summary[sp * 256u + (uint64_t)(g & 63u) * 4u + (g >> 6)] = (uint8_t)top;
    }
}

// SWAR: 0x80 in every byte of x that is >= m (m in 0..256), no cross-byte carries
struct hml_swar_ge {
    uint32_t add;    // per-byte addend
    int mode;        // 0: all bytes, 1: m <= 128, 2: m > 128, 3: none
};
__device__ __forceinline__ hml_swar_ge hml_swar_ge_make(uint32_t m) {
    hml_swar_ge g;
    if (m == 0u) { g.mode = 0; g.add = 0u; }
    else if (m > 255u) { g.mode = 3; g.add = 0u; }
    else if (m <= 128u) { g.mode = 1; g.add = (128u - m) * 0x01010101u; }
    else { g.mode = 2; g.add = (256u - m) * 0x01010101u; }
    return g;
}
__device__ __forceinline__ uint32_t hml_swar_ge_apply(const hml_swar_ge g, uint32_t x) {
    const uint32_t t = (x & 0x7f7f7f7fu) + g.add;
    const uint32_t r = g.mode == 1 ? (t | x) : (t & x);
    return g.mode == 0 ? 0x80808080u : (g.mode == 3 ? 0u : (r & 0x80808080u));
}
// gather the four 0x80 flags of a word into bits 0..3
__device__ __forceinline__ uint32_t hml_swar_compress(uint32_t flags) {
    return (((flags >> 7) * 0x00204081u) >> 21) & 15u;
}

// Summary scan: a wavefront takes HML_SUM_SPANS consecutive 4096-position spans.  Lane l reads one summary
// word per span (groups l, 64+l, 128+l, 192+l); ballots turn the per-group flags into bit masks in position
// order, from which every opened group gets its rank (mbcnt) in ONE list over all the wavefront's spans, kept
// in LDS.  Lane i then opens the i-th listed group: 16 float weights (one 64-byte line), compared exactly.
// All lanes work on different groups at once, so a wavefront carries the two dependent memory round trips
// (summary, weights) once for all its spans; the listed order is the position order, one wave scan places
// the block starts, and the scan value at a span's first listed group separates the spans.
#define HML_SUM_SPANS 4   // x 4 wavefronts per workgroup = one HML_GROUP_SPANS group
__device__ __forceinline__ uint32_t hml_mbcnt(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t hml_group_mask16(const float* __restrict__ w, uint64_t t0, uint32_t T, float thr) {
    uint32_t m16 = 0u;
    if (t0 + 16u <= T) {
        const float4* __restrict__ p = reinterpret_cast<const float4*>(w + t0);
        const float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        m16 = (uint32_t)!(v0.x < thr) | ((uint32_t)!(v0.y < thr) << 1) | ((uint32_t)!(v0.z < thr) << 2) | ((uint32_t)!(v0.w < thr) << 3) |
              ((uint32_t)!(v1.x < thr) << 4) | ((uint32_t)!(v1.y < thr) << 5) | ((uint32_t)!(v1.z < thr) << 6) | ((uint32_t)!(v1.w < thr) << 7) |
              ((uint32_t)!(v2.x < thr) << 8) | ((uint32_t)!(v2.y < thr) << 9) | ((uint32_t)!(v2.z < thr) << 10) | ((uint32_t)!(v2.w < thr) << 11) |
              ((uint32_t)!(v3.x < thr) << 12) | ((uint32_t)!(v3.y < thr) << 13) | ((uint32_t)!(v3.z < thr) << 14) | ((uint32_t)!(v3.w < thr) << 15);
    } else {
        // the group that straddles T (groups wholly beyond T hold nothing)
        for (uint32_t r = 0; r < 16u && t0 + r < T; ++r) m16 |= (uint32_t)!(w[t0 + r] < thr) << r;
    }
    return m16;
}
__device__ __forceinline__ void hml_b_compact_scan_summary(const uint8_t* __restrict__ summary, const float* __restrict__ w,
                                                                  uint32_t T, const hml_model* __restrict__ mdl,
                                                                  float thr_override, int use_override, int32_t base,
                                                                  uint16_t* __restrict__ stage,
                                                                  uint32_t* __restrict__ span_count,
                                                                  uint32_t* __restrict__ group_total) {
    static_assert(4 * HML_SUM_SPANS == HML_GROUP_SPANS, "a scan workgroup covers one span group");
    __shared__ uint16_t listed_all[4][HML_SUM_SPANS * 256];   // per wavefront: opened groups (span << 8 | group), position order
    __shared__ uint32_t wave_total[4];
    const int lane = threadIdx.x & 63;
    const uint32_t span0 = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * HML_SUM_SPANS;
    const uint32_t n_spans = (uint32_t)(((uint64_t)T + HML_SPAN - 1) / HML_SPAN);
    uint32_t wave_sum = 0u;
    if (span0 < n_spans) {   // wave-uniform
        // the summary is padded with zeros to whole spans
        uint32_t gw[HML_SUM_SPANS];
#pragma unroll
        for (int s = 0; s < HML_SUM_SPANS; ++s)
            gw[s] = (span0 + s < n_spans)
                        ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(summary) + (uint64_t)(span0 + s) * 64u + lane)
                        : 0u;
        uint16_t* listed = listed_all[threadIdx.x >> 6];
        const float thr = use_override ? thr_override : mdl->thr;
        // NaN threshold: !(w < thr) holds everywhere, every position starts a block; key 0 opens every group
        const uint32_t kthr = (thr != thr) ? 0u : hml_weight_key(thr, base);
        const hml_swar_ge sw_ge = hml_swar_ge_make(kthr);

        uint32_t first_of[HML_SUM_SPANS + 1];   // index of a span's first listed group
        uint32_t n_listed = 0u;
#pragma unroll
        for (int s = 0; s < HML_SUM_SPANS; ++s) {
            first_of[s] = n_listed;
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
        first_of[HML_SUM_SPANS] = n_listed;
        // (LDS operations of one wavefront complete in order: the reads below see the writes above)
        uint32_t starts_before[HML_SUM_SPANS + 1];   // block starts of the wavefront before a span's first listed group
#pragma unroll
        for (int s = 0; s <= HML_SUM_SPANS; ++s) starts_before[s] = 0xffffffffu;
        uint32_t running = 0u;
        for (uint32_t i0 = 0; i0 < n_listed; i0 += 64u) {   // wave-uniform; one pass unless > 64 groups are open
            const uint32_t i = i0 + (uint32_t)lane;
            uint32_t m16 = 0u, sg = 0u;
            if (i < n_listed) {
                sg = listed[i];
                m16 = hml_group_mask16(w, (uint64_t)(span0 + (sg >> 8)) * HML_SPAN + (uint64_t)(sg & 255u) * 16u, T, thr);
            }
            if (span0 == 0u && i == 0u) m16 |= 1u;   // position 0 (group 0 of span 0 is listed first)
            const uint32_t c = (uint32_t)__popc(m16);
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            const uint32_t excl = running + incl - c;
#pragma unroll
            for (int s = 0; s <= HML_SUM_SPANS; ++s)
                if (first_of[s] >= i0 && first_of[s] < i0 + 64u) starts_before[s] = __shfl(excl, (int)(first_of[s] - i0));
            const uint32_t sp = sg >> 8;
            uint32_t mine = starts_before[0];
#pragma unroll
            for (int s = 1; s < HML_SUM_SPANS; ++s) mine = (sp == (uint32_t)s) ? starts_before[s] : mine;
            uint16_t* __restrict__ out = stage + (uint64_t)(span0 + sp) * HML_SPAN;
            uint32_t pos = excl - mine;
            const uint32_t g16 = (sg & 255u) * 16u;
            while (m16) {
                const int b = __ffs(m16) - 1;
                m16 &= m16 - 1u;
                out[pos++] = (uint16_t)(g16 + (uint32_t)b);
            }
            running += __shfl(incl, 63);
        }
        // spans whose first index is the end of the list (nothing listed from there on)
#pragma unroll
        for (int s = 0; s <= HML_SUM_SPANS; ++s)
            if (starts_before[s] == 0xffffffffu) starts_before[s] = running;
#pragma unroll
        for (int s = 0; s < HML_SUM_SPANS; ++s)
            if (span0 + s < n_spans && lane == s) span_count[span0 + s] = starts_before[s + 1] - starts_before[s];
        wave_sum = running;
    }
    // the workgroup's block count: what the scatter kernel sums over the groups before a span
    if (lane == 0) wave_total[threadIdx.x >> 6] = wave_sum;
    __syncthreads();
    if (threadIdx.x == 0) group_total[blockIdx.x] = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
}
// the kernel: hml_b_compact_scan_summary over one chain (hml_k_many.h runs it over several chains in one launch)
HML_KERNEL __launch_bounds__(256) void hml_k_compact_scan_summary(const uint8_t* __restrict__ summary, const float* __restrict__ w,
                                                                  uint32_t T, const hml_model* __restrict__ mdl,
                                                                  float thr_override, int use_override, int32_t base,
                                                                  uint16_t* __restrict__ stage,
                                                                  uint32_t* __restrict__ span_count,
                                                                  uint32_t* __restrict__ group_total) {
    hml_b_compact_scan_summary(summary, w, T, mdl, thr_override, use_override, base, stage, span_count, group_total);
}


__device__ __forceinline__ uint32_t hml_wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// block count of every group of HML_GROUP_SPANS spans, for the float scan (the summary scan writes its own)
HML_KERNEL __launch_bounds__(256) void hml_k_group_totals(const uint32_t* __restrict__ span_count, uint32_t n_spans,
                                                          uint32_t* __restrict__ group_total) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t first = g * HML_GROUP_SPANS;
    if (first >= n_spans) return;
    uint32_t v = 0u;
#pragma unroll
    for (uint32_t i = 0; i < HML_GROUP_SPANS; ++i) v += (first + i < n_spans) ? span_count[first + i] : 0u;
    group_total[g] = v;
}

// One workgroup per span group, HML_SUM_SPANS spans per wavefront.  The exclusive offset of a span is the sum
// of the totals of the groups before its group (summed once per workgroup: T = 10^8 has 1526 groups, six
// loads per thread) plus the counts of the spans before it inside the group.
__device__ __forceinline__ void hml_b_compact_scatter(const uint16_t* __restrict__ stage,
                                                             const uint32_t* __restrict__ span_count,
                                                             const uint32_t* __restrict__ group_total, uint32_t n_spans,
                                                             uint32_t T, hml_model* __restrict__ mdl,
                                                             uint32_t* __restrict__ starts, uint32_t* __restrict__ host_B) {
    static_assert(4 * HML_SUM_SPANS == HML_GROUP_SPANS && HML_GROUP_SPANS <= 64, "one workgroup per span group");
    __shared__ uint32_t part[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x;
    const uint32_t first = g * HML_GROUP_SPANS;
    const uint32_t cap = mdl->cap;
    uint32_t acc = 0u;
    for (uint32_t i = threadIdx.x; i < g; i += 256u) acc += group_total[i];
    // lane l of every wavefront holds the count of span l of the group
    const uint32_t cnt_l = ((uint32_t)lane < HML_GROUP_SPANS && first + (uint32_t)lane < n_spans) ? span_count[first + (uint32_t)lane] : 0u;
    acc = hml_wave_sum_u32(acc);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    const uint32_t before_group = part[0] + part[1] + part[2] + part[3];
    // exclusive prefix of the group's span counts (16 lanes)
    uint32_t incl = cnt_l;
#pragma unroll
    for (int d = 1; d < HML_GROUP_SPANS; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    const uint32_t excl_l = before_group + incl - cnt_l;
#pragma unroll
    for (int k = 0; k < HML_SUM_SPANS; ++k) {
        const uint32_t in_group = (uint32_t)wave * HML_SUM_SPANS + (uint32_t)k;
        const uint32_t span = first + in_group;
        if (span >= n_spans) break;   // wave-uniform
        const uint32_t cnt = __shfl(cnt_l, (int)in_group);
        const uint32_t off = __shfl(excl_l, (int)in_group);
        const bool is_last = (span == n_spans - 1u);
        const uint32_t base = span * (uint32_t)HML_SPAN;
        const uint16_t* __restrict__ in = stage + (uint64_t)base;
        for (uint32_t i = lane; i < cnt; i += 64) if (off + i < cap) starts[off + i] = base + (uint32_t)in[i];   // (cap: hml_state.h, "block capacity")
        if (is_last && lane == 0) {
            const uint32_t B = off + cnt;
            if (B > cap) hml_halt(mdl, B, host_B);
            else {
                mdl->B = B;
                hml_warmup_for_many_blocks(mdl, B);
                starts[B] = T;
                // host-mapped word: lets the host size later grids without a copy in the stream
                if (host_B) __hip_atomic_store(host_B, B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
// the kernel: hml_b_compact_scatter over one chain (hml_k_many.h runs it over several chains in one launch)
HML_KERNEL __launch_bounds__(256) void hml_k_compact_scatter(const uint16_t* __restrict__ stage,
                                                             const uint32_t* __restrict__ span_count,
                                                             const uint32_t* __restrict__ group_total, uint32_t n_spans,
                                                             uint32_t T, hml_model* __restrict__ mdl,
                                                             uint32_t* __restrict__ starts, uint32_t* __restrict__ host_B) {
    hml_b_compact_scatter(stage, span_count, group_total, n_spans, T, mdl, starts, host_B);
}


// ------------------------------------------------------------------------------------------
// K4 for weakly compressed input (most positions start a block): the scan stages FLAGS, not offsets.  With B ~ T the 16-bit
// offsets above are 2 B per block written and read again; the flags of a span are 4096 bits = 64 words, whatever the
// number of blocks: the 64 ballots of the scan (iteration it, component j -> word 4 it + j, bit l = position
// 256 it + 4 l + j of the span), written as one 512-byte line.  The scatter kernel reads the line back, ranks the flags
// with the same ballot arithmetic and hands the starts of 256 positions at a time to memory through a per-wavefront LDS
// buffer, so that the stores are contiguous runs.  Same starts[] as the offset form, bit for bit (test_gpu_parity.py).
// Reference: Blocks<BreakpointArray>::next, src/Blocks/BreakpointArray.hpp:216-235.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void hml_wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
HML_KERNEL __launch_bounds__(256) void hml_k_compact_scan_bits(const float* __restrict__ w, uint32_t T,
                                                               const hml_model* __restrict__ mdl, float thr_override,
                                                               int use_override, unsigned long long* __restrict__ stage_bits,
                                                               uint32_t* __restrict__ span_count) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t span = blockIdx.x * 4u + (uint32_t)wave;
    const uint64_t base = (uint64_t)span * HML_SPAN;
    if (base >= T) return;
    const float thr = use_override ? thr_override : mdl->thr;
    uint32_t running = 0;
    unsigned long long mine = 0ull;   // word `lane` of the span's 64
    if (base + HML_SPAN <= T) {
        hml_f4 v[16];
        const hml_f4* __restrict__ p = reinterpret_cast<const hml_f4*>(w + base) + lane;
#pragma unroll
        for (int it = 0; it < 16; ++it) v[it] = __builtin_nontemporal_load(p + it * 64);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            bool f0 = !(v[it].x < thr);
            const bool f1 = !(v[it].y < thr), f2 = !(v[it].z < thr), f3 = !(v[it].w < thr);
            if (span == 0 && it == 0 && lane == 0) f0 = true;   // position 0 always starts a block
            const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1), m2 = __ballot(f2), m3 = __ballot(f3);
            mine = (lane == 4 * it) ? m0 : (lane == 4 * it + 1) ? m1 : (lane == 4 * it + 2) ? m2 : (lane == 4 * it + 3) ? m3 : mine;
            running += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
        }
    } else {
        for (int it = 0; it < 16; ++it) {
            bool f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t t = base + (uint64_t)it * 256u + (uint64_t)lane * 4u + j;
                f[j] = (t < T) ? (t == 0 || !(w[t] < thr)) : false;
            }
            const unsigned long long m0 = __ballot(f[0]), m1 = __ballot(f[1]), m2 = __ballot(f[2]), m3 = __ballot(f[3]);
            mine = (lane == 4 * it) ? m0 : (lane == 4 * it + 1) ? m1 : (lane == 4 * it + 2) ? m2 : (lane == 4 * it + 3) ? m3 : mine;
            running += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
        }
    }
    stage_bits[(uint64_t)span * 64u + (uint32_t)lane] = mine;
    if (lane == 0) span_count[span] = running;
}

HML_KERNEL __launch_bounds__(256) void hml_k_compact_scatter_bits(const unsigned long long* __restrict__ stage_bits,
                                                                  const uint32_t* __restrict__ span_count,
                                                                  const uint32_t* __restrict__ group_total, uint32_t n_spans,
                                                                  uint32_t T, hml_model* __restrict__ mdl,
                                                                  uint32_t* __restrict__ starts, uint32_t* __restrict__ host_B) {
    static_assert(4 * HML_SUM_SPANS == HML_GROUP_SPANS && HML_GROUP_SPANS <= 64, "one workgroup per span group");
    __shared__ uint32_t part[4];
    __shared__ uint32_t buf[4][256];   // per wavefront: the starts of 256 positions, in order
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x;
    const uint32_t first = g * HML_GROUP_SPANS;
    const uint32_t cap = mdl->cap;
    uint32_t acc = 0u;
    for (uint32_t i = threadIdx.x; i < g; i += 256u) acc += group_total[i];
    const uint32_t cnt_l = ((uint32_t)lane < HML_GROUP_SPANS && first + (uint32_t)lane < n_spans) ? span_count[first + (uint32_t)lane] : 0u;
    acc = hml_wave_sum_u32(acc);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    const uint32_t before_group = part[0] + part[1] + part[2] + part[3];
    uint32_t incl = cnt_l;
#pragma unroll
    for (int d = 1; d < HML_GROUP_SPANS; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    const uint32_t excl_l = before_group + incl - cnt_l;
    uint32_t* const mybuf = buf[wave];
    const unsigned long long lt = (1ull << lane) - 1ull;
    // the flag words of the wavefront's spans, all in flight
    unsigned long long word[HML_SUM_SPANS];
#pragma unroll
    for (int k = 0; k < HML_SUM_SPANS; ++k) {
        const uint32_t span = first + (uint32_t)wave * HML_SUM_SPANS + (uint32_t)k;
        word[k] = (span < n_spans) ? stage_bits[(uint64_t)span * 64u + (uint32_t)lane] : 0ull;
    }
#pragma unroll
    for (int k = 0; k < HML_SUM_SPANS; ++k) {
        const uint32_t in_group = (uint32_t)wave * HML_SUM_SPANS + (uint32_t)k;
        const uint32_t span = first + in_group;
        if (span >= n_spans) break;   // wave-uniform
        const uint32_t cnt = __shfl(cnt_l, (int)in_group);
        const uint32_t off = __shfl(excl_l, (int)in_group);
        const uint32_t base = span * (uint32_t)HML_SPAN;
        const uint32_t wlo = (uint32_t)word[k], whi = (uint32_t)(word[k] >> 32);
        uint32_t running = 0u;
        for (int it = 0; it < 16; ++it) {   // wave-uniform
            unsigned long long m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                m[j] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)whi, 4 * it + j) << 32) |
                       (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)wlo, 4 * it + j);   // (a wave-uniform lane: v_readlane, no LDS permute)
            const uint32_t n_it = (uint32_t)(__popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]));
            if (n_it == 0u) continue;
            uint32_t pos = (uint32_t)(__popcll(m[0] & lt) + __popcll(m[1] & lt) + __popcll(m[2] & lt) + __popcll(m[3] & lt));
            const uint32_t t0 = base + (uint32_t)it * 256u + (uint32_t)lane * 4u;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((m[j] >> lane) & 1ull) mybuf[pos++] = t0 + (uint32_t)j;
            hml_wave_lds_order();
            for (uint32_t i = (uint32_t)lane; i < n_it; i += 64u) if (off + running + i < cap) starts[off + running + i] = mybuf[i];
            hml_wave_lds_order();
            running += n_it;
        }
        if (span == n_spans - 1u && lane == 0) {
            const uint32_t B = off + cnt;
            if (B > cap) hml_halt(mdl, B, host_B);
            else {
                mdl->B = B;
                hml_warmup_for_many_blocks(mdl, B);
                starts[B] = T;
                if (host_B) __hip_atomic_store(host_B, B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// K5 block_stats - Statistics<IntegralArray,Normal>::addBlockStats / setStats (reference
// src/Statistics/IntegralArray.hpp:104-124,198-212) with KahanAggregator (src/KahanAggregator.hpp:26-45):
//   pos = Kahan(IA[start], IA[c] for every cell boundary c in (start,end));  neg = IA[end] unless
//   end % 65535 == 0;  (sum x, sum x^2) = pos - neg.      Same operations, same order, float.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void hml_block_stats_one(const float2* __restrict__ ia, uint32_t start, uint32_t end,
                                                    float& s, float& q) {
    float ps = 0.0f, pq = 0.0f, es = 0.0f, eq = 0.0f;
    {
        const float2 v = ia[start];
        const float y = v.x - es, t = ps + y; es = (t - ps) - y; ps = t;
        const float y2 = v.y - eq, t2 = pq + y2; eq = (t2 - pq) - y2; pq = t2;
    }
    for (uint64_t c = ((uint64_t)start + HML_CELLSIZE) / HML_CELLSIZE * HML_CELLSIZE; c < end; c += HML_CELLSIZE) {
        const float2 v = ia[c];
        const float y = v.x - es, t = ps + y; es = (t - ps) - y; ps = t;
        const float y2 = v.y - eq, t2 = pq + y2; eq = (t2 - pq) - y2; pq = t2;
    }
    float ns = 0.0f, nq = 0.0f;
    if (end % HML_CELLSIZE != 0) {
        const float2 v = ia[end];
        // KahanAggregator::subtract on a fresh aggregator: y = x - 0; temp = 0 + y
        const float y = v.x - 0.0f; ns = 0.0f + y;
        const float y2 = v.y - 0.0f; nq = 0.0f + y2;
    }
    s = ps - ns;
    q = pq - nq;
}

HML_KERNEL __launch_bounds__(256) void hml_k_block_stats(const float2* __restrict__ ia,
                                                         const uint32_t* __restrict__ starts,
                                                         const hml_model* __restrict__ mdl,
                                                         float2* __restrict__ bstat) {
    const uint32_t B = mdl->B;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += stride) {
        const uint32_t s = starts[b], e = starts[b + 1];
        float sx, sq;
        hml_block_stats_one(ia, s, e, sx, sq);
        bstat[b] = make_float2(sx, sq);
    }
}

#endif
