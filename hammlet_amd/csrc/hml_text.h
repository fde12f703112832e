// Text input: conversion of one whitespace-delimited decimal token to float, shared by the device kernels
// (hml_k_text.h) and the host.  Replaces the extraction `while ( input >> v )` of the reference's reader
// (reference src/wavelet.hpp:131; libstdc++ num_get::_M_extract_float + strtof) for the tokens whose result
// can be PROVEN here; every other token is reported as irregular and is resolved by the host with the real
// `istream >> float`, so the values are bit-identical to the reference's in every case.
//
// A token is regular if, from whitespace to whitespace, it reads
//     [+-]? ( D+ ( '.' D* )? | '.' D+ ) ( [eE] [+-]? D+ )?          (at most HML_TOK_MAX bytes)
// with at most 19 significant digits w and a decimal exponent q, and one of the following holds:
//   * w = 0                                                  -> +-0
//   * w < 2^53 and |q| <= 22 (after dropping trailing zeros of w): d = w * 10^q or w / 10^-q is ONE correctly
//     rounded double operation on exact operands; (float)d is the correctly rounded float unless d lies
//     exactly on the midpoint of two floats without being exact (then v may lie on either side: irregular;
//     an exact d - w * 5^q below 2^53, or 5^-q dividing w - is a true tie and rounds to even like strtof)
//   * otherwise: the 128-bit product of w (normalised) with the 64-bit truncated 5^q determines the 24-bit
//     mantissa and the rounding bit whenever the bits below are neither all zeros nor all ones (+-2): the
//     truncation moves the upper product word by at most one unit in either direction, so the kept bits and
//     "sticky != 0" are certain; else irregular.  Results outside the normal float range: irregular.
// Integer and IEEE basic operations only: hipcc for gfx950 and gcc give the same bits.
#ifndef HML_TEXT_H
#define HML_TEXT_H

#include "hml_common.h"
#include "hml_text_tables.h"

#define HML_TOK_MAX 48        // longest token the device converts (longer ones are irregular)
#define HML_TOK_OK 0
#define HML_TOK_IRREGULAR 1

#if defined(__HIP_DEVICE_COMPILE__)
__device__ static const uint64_t hml_p5_hi_dev[] = HML_P5_HI_TABLE;
__device__ static const int16_t hml_p5_e_dev[] = HML_P5_E_TABLE;
__device__ static const double hml_p10_dev[] = HML_P10_D_TABLE;
__device__ static const uint64_t hml_p5u_dev[] = HML_P5_U64_TABLE;
#define HML_P5_U hml_p5u_dev
#define HML_P5_HI hml_p5_hi_dev
#define HML_P5_E hml_p5_e_dev
#define HML_P10_D hml_p10_dev
#define HML_CLZ64(x) __clzll((long long)(x))
#define HML_MULHI64(a, b) __umul64hi((a), (b))
#else
static const uint64_t hml_p5_hi_host[] = HML_P5_HI_TABLE;
static const int16_t hml_p5_e_host[] = HML_P5_E_TABLE;
static const double hml_p10_host[] = HML_P10_D_TABLE;
static const uint64_t hml_p5u_host[] = HML_P5_U64_TABLE;
#define HML_P5_U hml_p5u_host
#define HML_P5_HI hml_p5_hi_host
#define HML_P5_E hml_p5_e_host
#define HML_P10_D hml_p10_host
#define HML_CLZ64(x) __builtin_clzll(x)
#define HML_MULHI64(a, b) ((uint64_t)(((unsigned __int128)(a) * (unsigned __int128)(b)) >> 64))
#endif

// whitespace of the "C" locale (what the stream's sentry skips)
HML_HD bool hml_is_space(uint32_t c) { return c == 32u || (c - 9u) <= 4u; }

// w * 10^q -> float bits (sign applied by the caller); false = cannot be decided here
HML_HD bool hml_decimal_to_float(uint64_t w, int q, uint32_t* bits) {
    if (w == 0) { *bits = 0; return true; }
    // trailing zeros of w only matter when they bring the pair into the range of the first branch (or make a tie exact)
    if (!(w < (1ull << 53) && q >= -22 && q <= 22))
        while (w % 10u == 0u) { w /= 10u; ++q; }
    if (w < (1ull << 53) && q >= -22 && q <= 22) {
        const double dw = (double)w;
        const double d = q >= 0 ? dw * HML_P10_D[q] : dw / HML_P10_D[-q];
        if ((hml_d2u(d) & 0x1fffffffull) == 0x10000000ull) {
            // d is the midpoint of two floats: the tie is real only if d is exact, i.e. w * 5^q < 2^53 or 5^-q | w
            const bool exact = q >= 0 ? (HML_MULHI64(w, HML_P5_U[q]) == 0 && w * HML_P5_U[q] < (1ull << 53)) : (w % HML_P5_U[-q] == 0);
            if (!exact) return false;
        }
        *bits = hml_f2u((float)d);
        return true;
    }
    if (q < HML_P5_QMIN || q > HML_P5_QMAX) return false;
    const int lz = HML_CLZ64(w);
    const uint64_t ph = HML_MULHI64(w << lz, HML_P5_HI[q - HML_P5_QMIN]);    // in [2^62, 2^64)
    const int s = 38 + (int)(ph >> 63);
    const uint64_t low = ph & ((1ull << s) - 1ull);
    if (low < 2ull || low > (1ull << s) - 3ull) return false;
    uint64_t m = ((ph >> s) + 1ull) >> 1;                                    // 24 bits, rounded (sticky is non-zero)
    int e2 = s + 65 + (int)HML_P5_E[q - HML_P5_QMIN] + q - lz;               // value = m * 2^e2
    if (m == (1ull << 24)) { m >>= 1; ++e2; }
    const int be = e2 + 23 + 127;
    if (be < 1 || be > 254) return false;
    *bits = ((uint32_t)be << 23) | ((uint32_t)m & 0x7fffffu);
    return true;
}

// One token (byte 0 is not whitespace): reading stops at the first whitespace byte, which must come within the
// first `limit` bytes (the callers pad their buffers with blanks), else the token is irregular.  Returns
// HML_TOK_OK and the value, or HML_TOK_IRREGULAR.  `get(i)` yields byte i.
template <class Get>
HML_HD int hml_parse_token(Get get, int limit, float* out) {
    int i = 0;
    uint32_t c = get(0);
    uint32_t sign = 0;
    if (c == '-' || c == '+') { sign = (c == '-') ? 0x80000000u : 0u; ++i; c = i < limit ? get(i) : 0u; }
    uint64_t w = 0;
    int nd = 0;          // significant digits taken into w
    int q = 0;
    bool any = false, dot = false, ok = true;
    for (;; ) {
        const uint32_t dgt = c - '0';
        if (dgt <= 9u) {
            any = true;
            if (w != 0 || dgt != 0) {
                if (nd == 19) ok = false;          // more than 19 significant digits
                else { w = w * 10u + dgt; ++nd; }
            }
            if (dot) --q;
        } else if (c == '.' && !dot) {
            dot = true;
        } else break;
        ++i;
        if (i >= limit) { c = 0u; break; }
        c = get(i);
    }
    if (!any) return HML_TOK_IRREGULAR;
    if (c == 'e' || c == 'E') {
        ++i;
        if (i >= limit) return HML_TOK_IRREGULAR;
        c = get(i);
        bool eneg = false;
        if (c == '-' || c == '+') {
            eneg = c == '-';
            ++i;
            if (i >= limit) return HML_TOK_IRREGULAR;
            c = get(i);
        }
        int ex = 0, ned = 0;
        for (;; ) {
            const uint32_t dgt = c - '0';
            if (dgt > 9u) break;
            if (ex < 100000) ex = ex * 10 + (int)dgt;
            ++ned;
            ++i;
            if (i >= limit) { c = 0u; break; }
            c = get(i);
        }
        if (ned == 0) return HML_TOK_IRREGULAR;
        q += eneg ? -ex : ex;
    }
    if (!hml_is_space(c) || !ok) return HML_TOK_IRREGULAR;   // trailing bytes (or the token is longer than `limit`)
    uint32_t bits;
    if (!hml_decimal_to_float(w, q, &bits)) return HML_TOK_IRREGULAR;
    *out = hml_u2f(bits | sign);
    return HML_TOK_OK;
}

#endif
