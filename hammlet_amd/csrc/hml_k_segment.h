// Maximum-posterior-margin segmentation of the recorded state marginals on the device - what the reference's
// post-processing tool computes from the marginals file (reference src/tools/maxSegmentation.cpp:53-82): the
// arg-max state of every marginal segment (first maximum, strict `>` starting from count 0, so an all-zero row gives
// state 0), adjacent segments with the same state merged.
//
// Input: the marginal segments in the form the run-length export already gathers - seg_start[M] and the count
// DIFFERENCES g[M][K] at the segment starts (counts of segment i = sum of g[0..i]).  Three-phase scan over chunks
// of 256 segments (one lane per segment), fused with the arg-max; then run starts are flagged, counted and
// scattered in order.  All of it is O(M*K) integer work on data that is already in HBM.
#ifndef HML_K_SEGMENT_H
#define HML_K_SEGMENT_H

#include "hml_state.h"

// phase 1: per chunk of 256 segments, the column sums of g -> chunk_sum[s * n_chunks + chunk]
HML_KERNEL __launch_bounds__(256) void hml_k_seg_partial(const int32_t* __restrict__ g, uint32_t M, int K,
                                                         int32_t* __restrict__ chunk_sum, uint32_t n_chunks) {
    __shared__ int32_t red[4][HML_CAP_K];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = 0; s < K; ++s) {
        int32_t v = i < M ? g[(uint64_t)i * K + s] : 0;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
        if (lane == 0) red[wave][s] = v;
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)K)
        chunk_sum[(uint64_t)threadIdx.x * n_chunks + blockIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// phase 3 (phase 2 is hml_k_dense_chunkscan over the K rows of chunk_sum): running counts of every segment and
// their arg-max -> seg_state[i]
// (KM: the most states the running counts are kept for - HML_MAX_K, or HML_CAP_K for the reference-compatible mode's larger models)
template <int KM>
HML_KERNEL __launch_bounds__(256) void hml_k_seg_argmax(const int32_t* __restrict__ g, uint32_t M, int K,
                                                        const int32_t* __restrict__ chunk_base, uint32_t n_chunks,
                                                        int16_t* __restrict__ seg_state) {
    __shared__ int32_t wsum[4][KM];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t incl[KM];
#pragma unroll
    for (int s = 0; s < KM; ++s) {
        if (s < K) {
            int32_t v = i < M ? g[(uint64_t)i * K + s] : 0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int32_t o = __shfl_up(v, d);
                if (lane >= d) v += o;
            }
            incl[s] = v;
            if (lane == 63) wsum[wave][s] = v;
        }
    }
    __syncthreads();
    int best = 0;
    int32_t best_count = 0;
#pragma unroll
    for (int s = 0; s < KM; ++s) {
        if (s < K) {
            int32_t c = chunk_base[(uint64_t)s * n_chunks + blockIdx.x] + incl[s];
            for (int w2 = 0; w2 < wave; ++w2) c += wsum[w2][s];
            if (c > best_count) { best_count = c; best = s; }
        }
    }
    if (i < M) seg_state[i] = (int16_t)best;
}

// run starts: segment i opens a run if i == 0 or its state differs from its predecessor's; count per chunk
HML_KERNEL __launch_bounds__(256) void hml_k_seg_run_count(const int16_t* __restrict__ seg_state, uint32_t M,
                                                           int32_t* __restrict__ chunk_runs) {
    __shared__ int32_t red[4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    int32_t f = (i < M && (i == 0 || seg_state[i] != seg_state[i - 1])) ? 1 : 0;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) f += __shfl_xor(f, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = f;
    __syncthreads();
    if (threadIdx.x == 0) chunk_runs[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// scatter, in order: run r starts at position seg_start[i] with state seg_state[i]
HML_KERNEL __launch_bounds__(256) void hml_k_seg_run_scatter(const int16_t* __restrict__ seg_state, const uint32_t* __restrict__ seg_start,
                                                             uint32_t M, const int32_t* __restrict__ chunk_base,
                                                             uint32_t* __restrict__ run_start, int16_t* __restrict__ run_state) {
    __shared__ int32_t wsum[4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int16_t st = i < M ? seg_state[i] : (int16_t)0;
    const int32_t f = (i < M && (i == 0 || st != seg_state[i - 1])) ? 1 : 0;
    int32_t incl = f;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t pos = chunk_base[blockIdx.x] + incl - f;
    for (int w2 = 0; w2 < wave; ++w2) pos += wsum[w2];
    if (f) { run_start[pos] = seg_start[i]; run_state[pos] = st; }
}

#endif
