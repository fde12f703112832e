// Philox4x32-10 counter-based generator (Salmon et al., "Parallel random numbers: as easy as
// 1, 2, 3", SC'11) and the addressing scheme the Gibbs sampler uses on the device.
//
// The reference draws everything from one sequential std::mt19937 (src/Distribution.hpp:15,
// src/main.cpp:107-108).  A sequential engine cannot feed 10^5 parallel categorical draws, so
// every random decision of a sweep gets an *address* instead:
//
//   counter = { draw>>2, index, epoch_lo, (kind<<24) | epoch_hi }      key = { seed_lo, seed_hi+chain }
//
//   kind   what is drawn                               index
//   CAT    backward categorical of FB sweep            trellis row (1..B)      (ForwardBackward.hpp:133-162)
//   MIX    per-block categorical of a mixture sweep    block (0..B-1)          (Mixture.hpp:111)
//   THETA  (gamma, normal) of emission parameter k     k                       (Distribution.hpp:77-87)
//   PI     gamma for initial-distribution entry k      k                       (Distribution.hpp:116-139)
//   TRANS  gamma for transition entry (i,j)            i*K+j                   (Distribution.hpp:162-178)
//   DATA   synthetic trace generator                   position
//
// `epoch` counts parameter-draw events of a chain (constructor draw, prior draws, sweeps).
// `draw` is the running index of 32-bit outputs inside one addressed sub-stream, so rejection
// loops can consume as many outputs as they need without disturbing any other variate.
#ifndef HML_PHILOX_H
#define HML_PHILOX_H

#include "hml_common.h"

enum {
    HML_KIND_CAT = 1,
    HML_KIND_MIX = 2,
    HML_KIND_THETA = 3,
    HML_KIND_PI = 4,
    HML_KIND_TRANS = 5,
    HML_KIND_DATA = 6,
    HML_KIND_HOST = 7    // draws made by the host on a chain's behalf (Trellis::sample of the C++ surface); index = call number
};

typedef struct { uint32_t v[4]; } hml_u32x4;

HML_HD uint32_t hml_mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

HML_HD hml_u32x4 hml_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                   uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = hml_mulhi32(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = hml_mulhi32(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    hml_u32x4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// Key of a chain.
typedef struct { uint32_t k0, k1; } hml_key;

HML_HD hml_key hml_make_key(uint64_t seed, uint32_t chain) {
    hml_key k;
    k.k0 = (uint32_t)seed;
    k.k1 = (uint32_t)(seed >> 32) + chain;
    return k;
}

// The four 32-bit outputs draw4*4 .. draw4*4+3 of sub-stream (kind, epoch, index).
HML_HD hml_u32x4 hml_stream4(hml_key key, uint32_t kind, uint64_t epoch, uint32_t index, uint32_t draw4) {
    return hml_philox4x32_10(draw4, index, (uint32_t)epoch,
                             (kind << 24) | ((uint32_t)(epoch >> 32) & 0x00ffffffu), key.k0, key.k1);
}

// A cursor over one addressed sub-stream: next() returns its 32-bit outputs in order.
typedef struct {
    hml_key key;
    uint32_t kind;
    uint64_t epoch;
    uint32_t index;
    uint32_t n;        // outputs consumed so far
    hml_u32x4 buf;
} hml_stream;

HML_HD hml_stream hml_stream_open(hml_key key, uint32_t kind, uint64_t epoch, uint32_t index) {
    hml_stream s;
    s.key = key; s.kind = kind; s.epoch = epoch; s.index = index; s.n = 0;
    s.buf.v[0] = s.buf.v[1] = s.buf.v[2] = s.buf.v[3] = 0;
    return s;
}

HML_HD uint32_t hml_stream_next(hml_stream* s) {
    uint32_t lane = s->n & 3u;
    if (lane == 0) s->buf = hml_stream4(s->key, s->kind, s->epoch, s->index, s->n >> 2);
    s->n++;
    // avoid a dynamically indexed register array on the device
    return lane == 0 ? s->buf.v[0] : lane == 1 ? s->buf.v[1] : lane == 2 ? s->buf.v[2] : s->buf.v[3];
}

#endif
