// expf / logf / powf as the reference's host computes them: glibc 2.35 (libm.so.6 of this image, Ubuntu GLIBC
// 2.35-0ubuntu3.12), sysdeps/ieee754/flt-32/e_expf.c, e_logf.c, e_powf.c (Szabolcs Nagy's ARM optimized-routines
// algorithms: table + polynomial in double, one rounding to float), in the variant its IFUNC resolver picks on a CPU
// with FMA and AVX2 (sysdeps/x86_64/fpu/multiarch: e_expf-fma.c, e_logf-fma.c, e_powf-fma.c - the same source built with
// -mfma, where GCC contracts every a * b + c into one fused operation).  The reference binary links libm dynamically
// (it is built by `g++ -O3 --std=c++11` without -static), so these are the bits behind src/StateSequence/ForwardBackward.hpp:83,117 (expf),
// src/EFD.hpp:37 and ForwardBackward.hpp:49 (logf), and libstdc++'s gamma / normal variates (logf, powf).
//
// Used by the reference-compatible mode only (hml_set_option "compat"): hml_math.h stays the arithmetic of the default
// path.  The fused operations are written out (__builtin_fma), so -ffp-contract=off keeps its meaning elsewhere.
// Pinned: tests/test_math_glibc_cpu.py compares each function with the host's libm - expf over all 2^32 inputs, logf over
// all 2^31 non-negative ones, powf on 2^31 pairs of the domain the sampler uses (0 <= x <= 1, y > 0) - and
// tests/test_gpu_compat.py the device against the same.  Tables: glibc's own (__logf_data, __powf_log2_data,
// __exp2f_data); tools/glibc_tables.py locates them in libm.so.6 and checks them against the literals below.
#ifndef HML_MATH_GLIBC_H
#define HML_MATH_GLIBC_H

#include "hml_math.h"

#if defined(__HIPCC__)
#define HML_GLIBC_FN __host__ __device__ __forceinline__
#else
#define HML_GLIBC_FN static inline __attribute__((target("fma")))
#endif
#define HML_FMA(a, b, c) __builtin_fma(a, b, c)

// __logf_data.tab: {invc, logc} for the 16 subintervals of [0x1.66p-1, 0x1.66p0)
#define HML_GLIBC_LOGF_TABLE                                                                        \
    {                                                                                               \
        0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2,   \
        0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2, 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3,    \
        0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3,      \
        0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4,   \
        0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1p+0, 0x0p+0,                                \
        0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5, 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4,      \
        0x1.b2036576afce6p-1, 0x1.526e57720db08p-3, 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3,      \
        0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2, 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2      \
    }
// __powf_log2_data.tab: {invc, log2(c)} (POWF_SCALE = 1: no fast to-integer instruction on x86-64)
#define HML_GLIBC_LOG2_TABLE                                                                        \
    {                                                                                               \
        0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2, 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2,   \
        0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2, 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2,    \
        0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2, 0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3,     \
        0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3, 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4,   \
        0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5, 0x1p+0, 0x0p+0,                                \
        0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4, 0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3,      \
        0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3, 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2,     \
        0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2, 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2      \
    }
#if defined(__HIP_DEVICE_COMPILE__)
__device__ static const double hml_glibc_logf_tab_dev[32] = HML_GLIBC_LOGF_TABLE;
__device__ static const double hml_glibc_log2_tab_dev[32] = HML_GLIBC_LOG2_TABLE;
#define HML_GLIBC_LOGF_TAB hml_glibc_logf_tab_dev
#define HML_GLIBC_LOG2_TAB hml_glibc_log2_tab_dev
#else
static const double hml_glibc_logf_tab_host[32] = HML_GLIBC_LOGF_TABLE;
static const double hml_glibc_log2_tab_host[32] = HML_GLIBC_LOG2_TABLE;
#define HML_GLIBC_LOGF_TAB hml_glibc_logf_tab_host
#define HML_GLIBC_LOG2_TAB hml_glibc_log2_tab_host
#endif

// e_expf.c (__expf), FMA build.  Differs from hml_expf (the same algorithm with separate roundings) on 2 of 2^32 inputs.
HML_GLIBC_FN float hml_glibc_expf(float x) {
    const uint32_t ix = hml_f2u(x);
    const uint32_t abstop = (ix >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {                                  // |x| >= 88 or not finite
        if (ix == 0xff800000u) return 0.0f;                  // -inf
        if (abstop >= 0x7f8u) return x + x;                  // +inf, NaN
        if (x > 0x1.62e42ep6f) return HML_INF_F;             // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;                 // underflow
        if (x < -0x1.9d1d9ep6f) return 0x1p-149f;            // __math_may_uflowf: 0x1.4p-75f * 0x1.4p-75f
    }
    const double xd = (double)x;
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
    const double Shift = 0x1.8p52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    double kd = HML_FMA(InvLn2N, xd, Shift);
    const uint64_t ki = hml_d2u(kd);
    kd = kd - Shift;
    const double r = HML_FMA(InvLn2N, xd, -kd);
    uint64_t t = HML_EXP2F_TAB[ki & 31u];
    t += ki << 47;
    const double s = hml_u2d(t);
    const double z = HML_FMA(C0, r, C1);
    const double r2 = r * r;
    double y = HML_FMA(C2, r, 1.0);
    y = HML_FMA(z, r2, y);
    y = y * s;
    return (float)y;
}

// e_logf.c (__logf), FMA build
HML_GLIBC_FN float hml_glibc_logf(float x) {
    uint32_t ix = hml_f2u(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {     // x < 0x1p-126, infinite or NaN
        if (ix * 2u == 0u) return -HML_INF_F;                // log(+-0) = -inf
        if (ix == 0x7f800000u) return x;                     // log(inf) = inf
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return hml_u2f(0x7fc00000u);   // negative or NaN: a NaN (never asked for; glibc's payload is not reproduced)
        ix = hml_f2u(x * 0x1p23f);                           // sub-normal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = HML_GLIBC_LOGF_TAB[2 * i], logc = HML_GLIBC_LOGF_TAB[2 * i + 1];
    const double z = (double)hml_u2f(iz);
    const double Ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    const double r = HML_FMA(z, invc, -1.0);
    const double y0 = HML_FMA((double)k, Ln2, logc);
    const double r2 = r * r;
    double y = HML_FMA(A1, r, A2);
    y = HML_FMA(A0, r2, y);
    y = HML_FMA(y, r2, y0 + r);
    return (float)y;
}

// e_powf.c (__powf), FMA build, for the arguments the sampler has: x in [0, 1] (a canonical uniform), y > 0 finite
// (1 / alpha of libstdc++'s gamma_distribution, random.tcc:2386).  Anything else: NaN (never asked for).
HML_GLIBC_FN float hml_glibc_powf_unit(float x, float y) {
    uint32_t ix = hml_f2u(x);
    const uint32_t iy = hml_f2u(y);
    if (!(y > 0.0f) || iy >= 0x7f800000u || (ix & 0x80000000u) || ix > 0x3f800000u) return hml_u2f(0x7fc00000u);
    if (ix == 0u) return 0.0f;                               // pow(+0, y > 0) = +0
    if (ix == 0x3f800000u) return 1.0f;                      // pow(1, y) = 1
    if (ix < 0x00800000u) {                                  // sub-normal x: normalise
        ix = hml_f2u(x * 0x1p23f);
        ix &= 0x7fffffffu;
        ix -= 23u << 23;
    }
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = HML_GLIBC_LOG2_TAB[2 * i], logc = HML_GLIBC_LOG2_TAB[2 * i + 1];
    const double z = (double)hml_u2f(iz);
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1,
                 A4 = 0x1.71547652ab82bp0;
    const double r = HML_FMA(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double yy = HML_FMA(A0, r, A1);
    const double p = HML_FMA(A2, r, A3);
    const double r4 = r2 * r2;
    double q = HML_FMA(A4, r, y0);
    q = HML_FMA(p, r2, q);
    yy = HML_FMA(yy, r4, q);
    const double ylogx = (double)y * yy;
    if (((hml_d2u(ylogx) >> 47) & 0xffffu) >= (hml_d2u(126.0) >> 47)) {   // |y log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return HML_INF_F;
        if (ylogx <= -150.0) return 0.0f;
        if (ylogx < -149.0) return 0x1p-149f;                              // __math_may_uflowf
    }
    // exp2_inline, sign_bias = 0
    const double ShiftS = 0x1.8p+52 / 32.0;
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    double kd = ylogx + ShiftS;
    const uint64_t ki = hml_d2u(kd);
    kd = kd - ShiftS;
    const double rr = ylogx - kd;
    uint64_t t = HML_EXP2F_TAB[ki & 31u];
    t += ki << 47;
    const double s = hml_u2d(t);
    const double zz = HML_FMA(C0, rr, C1);
    const double rr2 = rr * rr;
    double w = HML_FMA(C2, rr, 1.0);
    w = HML_FMA(zz, rr2, w);
    w = w * s;
    return (float)w;
}

#endif
