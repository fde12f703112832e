// One-time construction kernels: maxlet coefficients, Haar breakpoint weights, integral array.
#ifndef HML_K_BUILD_H
#define HML_K_BUILD_H

#include "hml_state.h"

// ------------------------------------------------------------------------------------------
// K1 haar_maxlet - MaxletTransform (reference src/wavelet.hpp:97-188), univariate.
//
// The reference streams values through a stack; the result is a fixed pairwise tree:
//   S^0[i] = x_i,  S^l[i] = S^(l-1)[2i] + S^(l-1)[2i+1]            (complete intervals only)
//   c[(2i+1) * 2^(l-1)] = n_l * |S^(l-1)[2i] - S^(l-1)[2i+1]|       if the right interval is complete
//                       = +inf                                       otherwise;  c[0] = +inf
// with n_l = sqrt2half^l as an iterated float product (norm[] is built on the host that way).
// One launch handles 10 levels: a workgroup reduces a tile of 1024 elements of the current
// level in LDS and emits the tile sum for the next launch (stride 2^base apart in c[]).
// ------------------------------------------------------------------------------------------
#define HML_MAXLET_LOG_TILE 10
#define HML_MAXLET_TILE 1024

HML_KERNEL __launch_bounds__(256) void hml_k_maxlet(const float* __restrict__ in, uint64_t n, int base,
                                                    float* __restrict__ coeff, uint64_t T,
                                                    float* __restrict__ out_sums, const float* __restrict__ norm) {
    __shared__ float s[HML_MAXLET_TILE];
    __shared__ float cl[HML_MAXLET_TILE];
    const uint64_t tile = blockIdx.x;
    const uint64_t tbase = tile * HML_MAXLET_TILE;   // in units of input elements
    const int tid = threadIdx.x;
    const uint64_t remain = n - tbase;
    const int nv = remain >= HML_MAXLET_TILE ? HML_MAXLET_TILE : (int)remain;   // complete input elements
    for (int i = tid; i < HML_MAXLET_TILE; i += 256) {
        s[i] = (i < nv) ? in[tbase + i] : 0.0f;
        cl[i] = HML_INF_F;
    }
    __syncthreads();
    const float inf = HML_INF_F;
    for (int j = 1; j <= HML_MAXLET_LOG_TILE; ++j) {
        const int npairs = HML_MAXLET_TILE >> j;
        const int nvj = nv >> (j - 1);   // complete elements at this level
        float lsum0 = 0.0f, lsum1 = 0.0f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = tid + q * 256;
            if (p < npairs) {
                const float l = s[2 * p], r = s[2 * p + 1];
                const int mid = (2 * p + 1) << (j - 1);   // local index of the wavelet's discontinuity
                float sum = 0.0f;
                if (2 * p + 1 < nvj) {
                    const float d = __builtin_fabsf(l - r);
                    const float c = norm[base + j] * d;
                    // max(0, c): std::max returns its first argument unless first < second
                    cl[mid] = (0.0f < c) ? c : 0.0f;
                    sum = l + r;
                } else {
                    cl[mid] = inf;
                }
                if (q == 0) lsum0 = sum; else lsum1 = sum;
            }
        }
        __syncthreads();
        if (tid < npairs) s[tid] = lsum0;
        if (tid + 256 < npairs) s[tid + 256] = lsum1;
        __syncthreads();
    }
    // write coefficients of every discontinuity inside the tile (local index 0 belongs to a higher level)
    for (int i = tid; i < HML_MAXLET_TILE; i += 256) {
        if (i == 0) continue;
        const uint64_t t = (tbase + (uint64_t)i) << base;
        if (t < T) coeff[t] = cl[i];
    }
    if (tid == 0) {
        if (nv == HML_MAXLET_TILE) out_sums[tile] = s[0];
        if (tile == 0 && base == 0) coeff[0] = inf;
    }
}

// positions that no launch reaches: multiples of 2^(10*launches) other than 0 cannot exist below T
// once 2^(10*launches) >= T; the host loops until the level array has one tile.

// ------------------------------------------------------------------------------------------
// K2 breakpoint_weights - HaarBreakpointWeights (reference src/wavelet.hpp:68-93) as a gather.
//
// The reference's in-place passes only ever combine RAW coefficients: position t (lowest set bit
// m) receives c[t-i] and c[t+i] for i = m/2, m/4, ..., 1, and the `R < size` test (strict)
// forces +inf whenever a wavelet's support ends at or beyond T:
//   e = (t + m < T) ? c[t] : inf
//   for i = m/2 .. 1:  e = max(e, c[t-i]);  if (t+i < T) e = max(e, (t+2i < T) ? c[t+i] : inf)
// w[0] = inf.  Then w *= multiplier (src/main.cpp:332-334).
// ------------------------------------------------------------------------------------------
HML_KERNEL __launch_bounds__(256) void hml_k_weights(const float* __restrict__ c, float* __restrict__ w, uint64_t T,
                                                     float multiplier) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const float inf = HML_INF_F;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += stride) {
        float e;
        if (t == 0) {
            e = inf;
        } else {
            const uint64_t m = t & (~t + 1);
            e = (t + m < T) ? c[t] : inf;
            for (uint64_t i = m >> 1; i >= 1; i >>= 1) {
                const float a = c[t - i];
                e = (e < a) ? a : e;   // std::max(e, a)
                if (t + i < T) {
                    const float b = (t + 2 * i < T) ? c[t + i] : inf;
                    e = (e < b) ? b : e;
                }
            }
        }
        w[t] = e * multiplier;
    }
}

HML_KERNEL __launch_bounds__(256) void hml_k_scale(float* __restrict__ w, uint64_t T, float multiplier) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += stride) w[t] = w[t] * multiplier;
}

// ------------------------------------------------------------------------------------------
// K3 integral_array - Statistics<IntegralArray,Normal> constructor (reference
// src/Statistics/IntegralArray.hpp:136-191) with KahanCumulativeSum(reverse) (src/utils.hpp:15-76).
//
// IA has T+1 pairs (sum x, sum x^2), IA[T] = (0,0).  Inside every cell [a, a+65535) the array
// holds the reverse Kahan-compensated cumulative sum of the cell.  The sum is sequential by
// definition, so one wavefront owns a cell: all lanes stage 1024 observations in LDS, lane 0
// runs the compensated recurrence, all lanes write the results back coalesced.
// ------------------------------------------------------------------------------------------
#define HML_IA_SEG 1024

HML_KERNEL __launch_bounds__(256) void hml_k_integral(const float* __restrict__ x, float2* __restrict__ ia, uint64_t T) {
    __shared__ float xs[4][HML_IA_SEG];
    __shared__ float2 os[4][HML_IA_SEG];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t n = T + 1;
    const uint64_t cell = (uint64_t)blockIdx.x * 4 + wave;
    const uint64_t a = cell * HML_CELLSIZE;
    if (a >= n) return;
    const uint64_t e = (a + HML_CELLSIZE < n) ? a + HML_CELLSIZE : n;   // one past the cell
    const uint64_t r = e - 1;                                           // last index of the cell
    // running state lives in lane 0 only
    float ss, sq, cs = 0.0f, cq = 0.0f;
    {
        const float xr = (r < T) ? x[r] : 0.0f;
        ss = (r < T) ? xr : 0.0f;
        sq = (r < T) ? xr * xr : 0.0f;
        if (lane == 0) ia[r] = make_float2(ss, sq);
    }
    // remaining indices r-1 ... a, processed in segments from the top
    uint64_t hi = r;   // one past the top index still to do
    while (hi > a) {
        const uint64_t lo = (hi - a > HML_IA_SEG) ? hi - HML_IA_SEG : a;
        const int len = (int)(hi - lo);
        for (int i = lane; i < len; i += 64) xs[wave][i] = x[lo + i];   // all < T because hi <= r <= T
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) {
            for (int i = len - 1; i >= 0; --i) {
                const float v = xs[wave][i];
                const float v2 = v * v;
                const float y = v - cs;
                const float t = ss + y;
                cs = (t - ss) - y;
                ss = t;
                const float y2 = v2 - cq;
                const float t2 = sq + y2;
                cq = (t2 - sq) - y2;
                sq = t2;
                os[wave][i] = make_float2(ss, sq);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < len; i += 64) ia[lo + i] = os[wave][i];
        hi = lo;
    }
}

#endif
