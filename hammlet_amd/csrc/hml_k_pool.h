// Chain-parallel pooling of the state marginals: kernels that pack a chain's recorded marginals
// into the all-reduce payload and unpack the pooled payload back into a chain context.
//
// A chain keeps its marginals as per-state DIFFERENCE arrays diff[K][T+1] (+1 where a recorded segment of the state
// starts, -1 where it ends) and a bitmap of segment boundaries (hml_k_record, reference src/StateMarginals.hpp:51-137).
// Both are additive over recorded sweeps - and therefore over chains once the states carry common labels - except
// that +1/-1 of different sweeps can cancel at a boundary, so the boundaries travel as their own row:
//   payload (int32): rows 0..K-1  diff[perm[r]][0..T]      relabelled difference arrays
//                    row  K       1 at segment boundaries  (sum > 0 after the all-reduce <=> some chain cut there)
//                    tail         [recorded sweeps, used[0..K-1]]   used[r] = relabelled state r was ever recorded
// One ncclAllReduce(sum, int32) over the payload pools everything.
#ifndef HML_K_POOL_H
#define HML_K_POOL_H

#include "hml_state.h"

HML_KERNEL __launch_bounds__(256) void hml_k_pool_export(const int32_t* __restrict__ diff, const uint32_t* __restrict__ boundary,
                                                         const hml_model* __restrict__ mdl, const int32_t* __restrict__ perm,
                                                         uint32_t T, int K, int32_t* __restrict__ payload) {
    const uint64_t T1 = (uint64_t)T + 1u;
    const uint64_t n = (uint64_t)(K + 1) * T1;
    int32_t* __restrict__ tail = payload + n;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long seen = 0ull;   // wave-uniform: rows this wavefront has flagged as used already
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x; i0 < n; i0 += stride) {   // workgroup-uniform trip count
        const uint64_t i = i0 + threadIdx.x;
        int32_t v = 0;
        int r = -1;
        if (i < n) {
            r = (int)(i / T1);
            const uint64_t t = i - (uint64_t)r * T1;
            if (r < K) v = diff[(uint64_t)perm[r] * T1 + t];
            else v = (t < T && (t == 0u || ((boundary[t >> 5] >> (t & 31u)) & 1u))) ? 1 : 0;
            payload[i] = v;
        }
        // a state was recorded at least once iff its difference array is not identically zero (its prefix sums are
        // the non-negative counts, and they start from zero).  One atomic per wavefront and row at most (round 4: one per
        // non-zero ELEMENT cost 0.7 s on config 5, whose 3 10^7 segments all landed on the same K words)
        bool nz = (r >= 0 && r < K && v != 0);
        unsigned long long m = __ballot(nz);
        while (m != 0ull) {   // wave-uniform; one trip unless the wavefront straddles two rows
            const int r0 = __builtin_amdgcn_readlane(r, __ffsll((long long)m) - 1);
            if (!((seen >> r0) & 1ull)) {
                if ((threadIdx.x & 63u) == 0u) atomicOr(reinterpret_cast<unsigned int*>(tail + 1 + r0), 1u);
                seen |= 1ull << r0;
            }
            nz = nz && r != r0;
            m = __ballot(nz);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) tail[0] = (int32_t)mdl->n_recorded;
}

HML_KERNEL __launch_bounds__(256) void hml_k_pool_add(int32_t* __restrict__ acc, const int32_t* __restrict__ other, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc[i] += other[i];
}

// pooled payload -> the context's difference arrays (states now carry the common labels)
HML_KERNEL __launch_bounds__(256) void hml_k_pool_install_diff(const int32_t* __restrict__ payload, uint32_t T, int K,
                                                               int32_t* __restrict__ diff) {
    const uint64_t n = (uint64_t)K * ((uint64_t)T + 1u);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) diff[i] = payload[i];
}

// ... and the boundary bitmap (one thread per 32-bit word) plus the recorded-sweep bookkeeping of the model
HML_KERNEL __launch_bounds__(256) void hml_k_pool_install_boundary(const int32_t* __restrict__ payload, uint32_t T, int K,
                                                                   uint32_t* __restrict__ boundary, hml_model* __restrict__ mdl) {
    const uint64_t T1 = (uint64_t)T + 1u;
    const int32_t* __restrict__ row = payload + (uint64_t)K * T1;
    const uint64_t words = (T1 + 31u) / 32u;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += stride) {
        uint32_t bits = 0u;
        for (uint32_t j = 0; j < 32u; ++j) {
            const uint64_t t = w * 32u + j;
            if (t < T1 && row[t] != 0) bits |= 1u << j;
        }
        boundary[w] = bits;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int32_t* __restrict__ tail = payload + (uint64_t)(K + 1) * T1;
        mdl->n_recorded = (unsigned long long)(uint32_t)tail[0];
        int mx = -1;
        for (int r = 0; r < K; ++r) if (tail[1 + r] != 0) mx = r;
        mdl->max_state_recorded = mx;
    }
}

// ---- the boundary-list form (round 4).  The difference arrays are zero except at recorded segment boundaries, and a strongly
// compressed chain has few of them (config 3 after 100 recorded sweeps: 23 000 boundaries in 10^8 positions - 0.6 MB of
// non-zero entries inside a 2.4 GB payload).  One chain's list (int32):
//   header   [M, recorded sweeps, used[0..K-1]]
//   entries  M x [position, delta of relabelled state 0, ..., K-1]        (the chain's marginal segments, in position order)
// The ranks exchange their lists (ncclAllGather of equally sized slots) and every rank adds all of them into zeroed
// difference arrays: the same pooled arrays as the sum of the dense payloads.
HML_HD uint64_t hml_pool_list_header(int K) { return 2u + (uint64_t)K; }

HML_KERNEL __launch_bounds__(256) void hml_k_pool_list_pack(const uint32_t* __restrict__ seg, const int32_t* __restrict__ g, uint64_t M,
                                                            const hml_model* __restrict__ mdl, const int32_t* __restrict__ perm, int K,
                                                            int32_t* __restrict__ list) {
    const uint64_t H = hml_pool_list_header(K);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
        int32_t* __restrict__ e = list + H + i * (uint64_t)(K + 1);
        e[0] = (int32_t)seg[i];
        for (int r = 0; r < K; ++r) {
            const int32_t v = g[i * (uint64_t)K + (uint64_t)perm[r]];
            e[1 + r] = v;
            // a state was recorded at least once iff its difference array is not identically zero (hml_k_pool_export)
            if (v != 0) atomicOr(reinterpret_cast<unsigned int*>(list + 2 + r), 1u);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { list[0] = (int32_t)(uint32_t)M; list[1] = (int32_t)mdl->n_recorded; }
}

// all ranks' lists (n_lists slots of `slot` int32 each) -> the context's ZEROED difference arrays and boundary bitmap
HML_KERNEL __launch_bounds__(256) void hml_k_pool_list_install(const int32_t* __restrict__ lists, int n_lists, uint64_t slot, uint32_t T, int K,
                                                               int32_t* __restrict__ diff, uint32_t* __restrict__ boundary,
                                                               hml_model* __restrict__ mdl) {
    const uint64_t H = hml_pool_list_header(K);
    const uint64_t T1 = (uint64_t)T + 1u;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (int r = 0; r < n_lists; ++r) {
        const int32_t* __restrict__ l = lists + (uint64_t)r * slot;
        const uint64_t M = (uint64_t)(uint32_t)l[0];
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
            const int32_t* __restrict__ e = l + H + i * (uint64_t)(K + 1);
            const uint32_t t = (uint32_t)e[0];
            if (t > T) continue;   // (cannot happen; the arrays end at T)
            for (int s = 0; s < K; ++s)
                if (e[1 + s] != 0) atomicAdd(&diff[(uint64_t)s * T1 + t], e[1 + s]);
            atomicOr(&boundary[t >> 5], 1u << (t & 31u));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long rec = 0ull;
        int mx = -1;
        for (int r = 0; r < n_lists; ++r) {
            const int32_t* __restrict__ l = lists + (uint64_t)r * slot;
            rec += (unsigned long long)(uint32_t)l[1];
            for (int s = 0; s < K; ++s) if (l[2 + s] != 0 && s > mx) mx = s;
        }
        mdl->n_recorded = rec;
        mdl->max_state_recorded = mx;
        atomicOr(&boundary[0], 1u);   // position 0 starts a segment (the dense payload's boundary row says so too)
    }
}

#endif