// Finalisation of the state marginals: boundary compaction, per-segment gathers, dense expansion.
#ifndef HML_K_MARGINALS_H
#define HML_K_MARGINALS_H

#include "hml_state.h"

// count boundary bits per span of 4096 positions (128 words); position 0 always counts
HML_KERNEL __launch_bounds__(256) void hml_k_marg_count(const uint32_t* __restrict__ boundary, uint32_t T,
                                                        uint32_t* __restrict__ span_count) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t span = blockIdx.x * 4u + (uint32_t)wave;
    const uint64_t base = (uint64_t)span * HML_SPAN;
    if (base >= T) return;
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint64_t word = base / 32u + (uint64_t)lane + 64u * k;
        const uint64_t t0 = word * 32u;
        uint32_t bits = (t0 < T) ? boundary[word] : 0u;
        if (t0 < T && t0 + 32u > T) bits &= (1u << (T - t0)) - 1u;
        if (word == 0) bits |= 1u;
        cnt += __popc(bits);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) cnt += __shfl_xor(cnt, m);
    if (lane == 0) span_count[span] = cnt;
}

// write the boundary positions of each span, in order, at seg_start[span_offset + ...]
HML_KERNEL __launch_bounds__(256) void hml_k_marg_scatter(const uint32_t* __restrict__ boundary, uint32_t T,
                                                          const uint32_t* __restrict__ span_offset,
                                                          uint32_t* __restrict__ seg_start) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t span = blockIdx.x * 4u + (uint32_t)wave;
    const uint64_t base = (uint64_t)span * HML_SPAN;
    if (base >= T) return;
    uint32_t run = span_offset[span];
    for (int k = 0; k < 2; ++k) {
        const uint64_t word = base / 32u + (uint64_t)lane + 64u * k;
        const uint64_t t0 = word * 32u;
        uint32_t bits = (t0 < T) ? boundary[word] : 0u;
        if (t0 < T && t0 + 32u > T) bits &= (1u << (T - t0)) - 1u;
        if (word == 0) bits |= 1u;
        const uint32_t c = __popc(bits);
        // exclusive prefix of c across the wavefront
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        uint32_t pos = run + incl - c;
        while (bits) {
            const int bit = __ffs(bits) - 1;
            bits &= bits - 1u;
            seg_start[pos++] = (uint32_t)(t0 + (uint64_t)bit);
        }
        run += __shfl(incl, 63);
    }
}

// gather diff[s][seg_start[i]] for all states: out[i*K + s]
HML_KERNEL __launch_bounds__(256) void hml_k_marg_gather(const int32_t* __restrict__ diff, uint32_t T, int K,
                                                         const uint32_t* __restrict__ seg_start, uint32_t M,
                                                         int32_t* __restrict__ out) {
    const uint64_t T1 = (uint64_t)T + 1u;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
        const uint32_t t = seg_start[i];
        for (int s = 0; s < K; ++s) out[(uint64_t)i * K + s] = diff[(uint64_t)s * T1 + t];
    }
}

// ---- dense expansion: counts[s][t] = prefix sum over t of diff[s][t]; row K = boundary indicator ----
// three-phase scan over chunks of 4096 positions per state
HML_KERNEL __launch_bounds__(256) void hml_k_dense_partial(const int32_t* __restrict__ diff, uint32_t T, int K,
                                                           int32_t* __restrict__ chunk_sum, uint32_t n_chunks) {
    __shared__ int32_t red[4];
    const int s = blockIdx.y;
    const uint32_t chunk = blockIdx.x;
    const uint64_t T1 = (uint64_t)T + 1u;
    const uint64_t base = (uint64_t)chunk * HML_SPAN;
    int32_t acc = 0;
    for (int k = 0; k < 16; ++k) {
        const uint64_t t = base + (uint64_t)k * 256u + threadIdx.x;
        if (t < T) acc += diff[(uint64_t)s * T1 + t];
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sum[(uint64_t)s * n_chunks + chunk] = red[0] + red[1] + red[2] + red[3];
}

HML_KERNEL __launch_bounds__(1024) void hml_k_dense_chunkscan(int32_t* __restrict__ chunk_sum, uint32_t n_chunks) {
    __shared__ int32_t part[1024];
    const int s = blockIdx.x;
    int32_t* cs = chunk_sum + (uint64_t)s * n_chunks;
    const int tid = threadIdx.x;
    const uint32_t per = (n_chunks + 1023u) / 1024u;
    const uint32_t a = (uint32_t)tid * per < n_chunks ? (uint32_t)tid * per : n_chunks;
    const uint32_t b = (a + per < n_chunks) ? a + per : n_chunks;
    int32_t sum = 0;
    for (uint32_t i = a; i < b; ++i) sum += cs[i];
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int32_t v = (tid >= d) ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int32_t run = part[tid] - sum;
    for (uint32_t i = a; i < b; ++i) { const int32_t v = cs[i]; cs[i] = run; run += v; }
}

HML_KERNEL __launch_bounds__(256) void hml_k_dense_final(const int32_t* __restrict__ diff, uint32_t T, int K,
                                                         const int32_t* __restrict__ chunk_sum, uint32_t n_chunks,
                                                         const int32_t* __restrict__ perm, int32_t* __restrict__ out) {
    __shared__ int32_t wsum[4];
    const int snew = blockIdx.y;
    const int s = perm ? perm[snew] : snew;
    const uint32_t chunk = blockIdx.x;
    const uint64_t T1 = (uint64_t)T + 1u;
    const uint64_t base = (uint64_t)chunk * HML_SPAN;
    int32_t run = chunk_sum[(uint64_t)s * n_chunks + chunk];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < 16; ++k) {
        const uint64_t t = base + (uint64_t)k * 256u + threadIdx.x;
        const int32_t v = (t < T) ? diff[(uint64_t)s * T1 + t] : 0;
        int32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int32_t before = 0;
        for (int w2 = 0; w2 < wave; ++w2) before += wsum[w2];
        const int32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t < T) out[(uint64_t)snew * T + t] = run + before + incl;
        run += total;
        __syncthreads();
    }
}

HML_KERNEL __launch_bounds__(256) void hml_k_dense_boundary(const uint32_t* __restrict__ boundary, uint32_t T,
                                                            int32_t* __restrict__ out_row) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += stride)
        out_row[t] = (t == 0) ? 1 : (int32_t)((boundary[t >> 5] >> (t & 31u)) & 1u);
}

#endif
