// Text reader kernels: the values of a whitespace-separated decimal text, in order, as float32 - what the
// reference's `while ( input >> v )` produces one value at a time (reference src/wavelet.hpp:131, the start-up
// cost that dwarfs the sampling: SURVEY.md section 8f rank 1).
//
// HBM-bound byte work, two passes over a chunk of text that was cut at whitespace by the host:
//   hml_k_text_count   16 bytes per lane, 4 KiB per workgroup: a token starts where a non-blank byte follows a
//                      blank one; token starts per tile -> tile_count[]
//   hml_k_text_scan    one workgroup: exclusive prefix of the tile counts (tile_base[]) + the chunk's total
//   hml_k_text_parse   the tile (+ 64 bytes overhang) is staged in LDS, the token starts are compacted into an LDS
//                      list in position order, and lane k converts the k-th token of the tile (dense lanes,
//                      hml_parse_token).  values[tile_base + k] is written coalesced.  Tokens that the device
//                      cannot decide (hml_text.h) are appended to a list {token index, byte offset}; the host
//                      resolves exactly those with the stream extraction of libstdc++.
// Traffic: 2 x n bytes read + 4 bytes per token written.  The text buffer is padded with blanks to a whole
// number of tiles plus one, so no load needs a bounds check and every token ends at a blank.
#ifndef HML_K_TEXT_H
#define HML_K_TEXT_H

#include "hml_text.h"

#define HML_TEXT_TILE 4096u        // bytes per workgroup tile = 256 lanes x 16 bytes
#define HML_TEXT_OVERHANG 64u      // >= HML_TOK_MAX + 1
#define HML_TEXT_MAX_TOK_TILE 2048u

struct hml_text_irr { uint32_t token; uint32_t offset; };   // chunk-local token index and byte offset
struct hml_text_meta { uint32_t tokens; uint32_t irregular; };

// bit j = byte j of the 16-byte vector is whitespace
__device__ __forceinline__ uint32_t hml_ws_mask16(const uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t c = (w[i] >> (8 * b)) & 0xffu;
            m |= (uint32_t)hml_is_space(c) << (4 * i + b);
        }
    }
    return m;
}

// token-start mask of this lane's 16 bytes; `text` is the chunk (tile-aligned), off = byte offset of the lane
__device__ __forceinline__ uint32_t hml_text_starts(const uint8_t* __restrict__ text, uint64_t off, const uint4 v) {
    const uint32_t ws = hml_ws_mask16(v);
    // the byte before this lane's first byte: the previous lane's last one (blank before the chunk's first byte)
    uint32_t prev = __shfl_up((int)(ws >> 15), 1) & 1u;
    if ((threadIdx.x & 63u) == 0u) prev = off == 0 ? 1u : (uint32_t)hml_is_space(text[off - 1]);
    return ~ws & ((ws << 1) | prev) & 0xffffu;
}

HML_KERNEL __launch_bounds__(256) void hml_k_text_count(const uint8_t* __restrict__ text, uint32_t n_tiles,
                                                        uint32_t* __restrict__ tile_count) {
    __shared__ uint32_t wsum[4];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint64_t off = (uint64_t)tile * HML_TEXT_TILE + threadIdx.x * 16u;
    const uint4 v = *reinterpret_cast<const uint4*>(text + off);
    uint32_t c = __popc(hml_text_starts(text, off, v));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor((int)c, d);
    if ((threadIdx.x & 63u) == 0u) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[tile] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one workgroup of 1024: tile_base = exclusive prefix of tile_count; meta->tokens = total
HML_KERNEL __launch_bounds__(1024) void hml_k_text_scan(const uint32_t* __restrict__ tile_count, uint32_t n_tiles,
                                                        uint32_t* __restrict__ tile_base, hml_text_meta* __restrict__ meta) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (n_tiles + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, n_tiles);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += tile_count[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t c = tile_count[i]; tile_base[i] = run; run += c; }
    if (threadIdx.x == 1023u) meta->tokens = part[1023];
}

HML_KERNEL __launch_bounds__(256) void hml_k_text_parse(const uint8_t* __restrict__ text, uint32_t n_tiles,
                                                        const uint32_t* __restrict__ tile_base, float* __restrict__ values,
                                                        hml_text_meta* __restrict__ meta, hml_text_irr* __restrict__ irr,
                                                        uint32_t irr_cap) {
    __shared__ uint4 bytes4[(HML_TEXT_TILE + HML_TEXT_OVERHANG) / 16];
    __shared__ uint16_t toks[HML_TEXT_MAX_TOK_TILE];
    __shared__ uint32_t wsum[4];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint64_t tile_off = (uint64_t)tile * HML_TEXT_TILE;
    const uint64_t off = tile_off + threadIdx.x * 16u;
    const uint4 v = *reinterpret_cast<const uint4*>(text + off);
    bytes4[threadIdx.x] = v;
    if (threadIdx.x < HML_TEXT_OVERHANG / 16u)
        bytes4[256u + threadIdx.x] = *reinterpret_cast<const uint4*>(text + tile_off + HML_TEXT_TILE + threadIdx.x * 16u);
    uint32_t starts = hml_text_starts(text, off, v);
    const uint32_t cnt = __popc(starts);
    // exclusive position of this lane's first token start within the tile
    uint32_t inc = cnt;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
        if (lane >= (uint32_t)d) inc += o;
    }
    if (lane == 63u) wsum[wave] = inc;
    __syncthreads();
    uint32_t pos = inc - cnt;
    for (uint32_t wv = 0; wv < wave; ++wv) pos += wsum[wv];
    const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    while (starts) {
        const uint32_t j = (uint32_t)__ffs((int)starts) - 1u;
        starts &= starts - 1u;
        toks[pos++] = (uint16_t)(threadIdx.x * 16u + j);
    }
    __syncthreads();
    const uint8_t* __restrict__ bytes = reinterpret_cast<const uint8_t*>(bytes4);
    const uint32_t base = tile_base[tile];
    for (uint32_t k = threadIdx.x; k < total; k += 256u) {
        const uint32_t t0 = toks[k];
        float val = 0.0f;
        const int st = hml_parse_token([bytes, t0](int i) { return (uint32_t)bytes[t0 + (uint32_t)i]; }, HML_TOK_MAX + 1, &val);
        values[base + k] = st == HML_TOK_OK ? val : 0.0f;
        if (st != HML_TOK_OK) {
            const uint32_t slot = atomicAdd(&meta->irregular, 1u);
            if (slot < irr_cap) { irr[slot].token = base + k; irr[slot].offset = (uint32_t)tile_off + t0; }
        }
    }
}

#endif
