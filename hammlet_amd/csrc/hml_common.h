// Shared host/device definitions for the HaMMLET hot path on MI355X.
// Everything in the hml_*.h headers is written with IEEE-754 basic operations only
// (+ - * / sqrt, integer ops, explicit conversions), so that gcc on the host and hipcc
// for gfx950 produce bit-identical results when both are built with -ffp-contract=off.
#ifndef HML_COMMON_H
#define HML_COMMON_H

#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HML_HD __host__ __device__ __forceinline__
#define HML_HDM __host__ __device__ __forceinline__
// a kernel: internal linkage, so that every object of the library (hml_capi.hip is compiled once for the core and once per
// number of states) holds - and loads - its own copy of the kernels it launches
#define HML_KERNEL static __global__
#else
#define HML_HD static inline
#define HML_HDM inline
#endif

// Integral-array cell size (reference: src/Statistics/IntegralArray.hpp:24).
#define HML_CELLSIZE 65535u

// Maximum number of states the device kernels are compiled for.
#define HML_MAX_K 16            // states of the default path (register-resident kernels instantiated for 2 .. 16; 4-bit packed maps)
#define HML_CAP_K 64            // states the model's arrays hold: the reference-compatible mode takes K at run time, a lane per state

HML_HD uint32_t hml_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
HML_HD float hml_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
HML_HD uint64_t hml_d2u(double d) { return __builtin_bit_cast(uint64_t, d); }
HML_HD double hml_u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

#define HML_INF_F (hml_u2f(0x7f800000u))

#endif
