// div_check: hml_tr2_quotient (hml_k_trellis_rows.h: f / Z through one double reciprocal and a correction step) against
// the float division itself, on the GPU: random pairs over the whole range (0 <= f <= Z and unrestricted), quotients on
// the sub-normal grid, and constructed exact ties there (Z = b 2^e, f = b (2n + 1) 2^(e - 150)).  Prints mismatch counts.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Ihammlet_amd/csrc -o gpurun_out/div_check tools/div_check.hip && gpurun_out/div_check
#include "hml_k_trellis_rows.h"
#include <cstdio>

__device__ __forceinline__ uint32_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}
__device__ __forceinline__ bool check(float f, float Z, unsigned long long* bad, float* ex) {
    if (!(Z > 0.0f) || !(Z < 3.4028234663852886e38f) || !(f >= 0.0f) || !(f < 3.4028234663852886e38f)) return true;
    const double Zd = (double)Z;
    const float a = hml_tr2_quotient(f, Zd, hml_tr2_reciprocal(Zd));
    const float b = f / Z;
    if (hml_f2u(a) != hml_f2u(b)) {
        if (atomicAdd(bad, 1ull) < 8ull) { ex[0] = f; ex[1] = Z; ex[2] = a; ex[3] = b; }
        return false;
    }
    return true;
}
// mode 0: random bit patterns, f <= Z; 1: unrestricted; 2: quotient forced to the sub-normal range; 3: exact ties
__global__ void k(int mode, uint64_t n, uint64_t seed, unsigned long long* bad, float* ex) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (mode == 3) {
            // i -> (b odd < 4096, n2 < 2048, e in a few exponents)
            const uint32_t b = 2u * (uint32_t)(i % 2048u) + 1u;
            const uint32_t n2 = (uint32_t)((i / 2048u) % 2048u);
            const int e = (int)((i / (2048u * 2048u)) % 40u) - 20;
            if ((uint64_t)b * (2u * n2 + 1u) >= (1u << 24)) continue;
            const float Z = ldexpf((float)b, e);
            const float f = ldexpf((float)(b * (2u * n2 + 1u)), e - 150);
            check(f, Z, bad, ex);
            continue;
        }
        const uint32_t r0 = mix(i * 2 + seed * 0x9e3779b97f4a7c15ull), r1 = mix(i * 2 + 1 + seed * 0x9e3779b97f4a7c15ull);
        float Z = hml_u2f(r0 & 0x7fffffffu), f = hml_u2f(r1 & 0x7fffffffu);
        if (mode == 0 && f > Z) { const float t = f; f = Z; Z = t; }
        if (mode == 2) {   // f = Z * 2^-(126 .. 150) * (1 .. 2)
            Z = hml_u2f(0x20000000u + (r0 % 0x3f000000u));                     // 2^-63 .. 2^63
            const int sh = 126 + (int)(r1 % 25u);
            f = ldexpf(Z * (1.0f + (float)(r1 >> 8) * 5.9604645e-08f), -sh);
        }
        check(f, Z, bad, ex);
    }
}

int main() {
    unsigned long long* d_bad; float* d_ex;
    hipMalloc(&d_bad, 8); hipMalloc(&d_ex, 16);
    const char* names[] = {"random, f <= Z", "random, unrestricted", "sub-normal quotients", "exact ties on the sub-normal grid"};
    const uint64_t counts[] = {1ull << 34, 1ull << 33, 1ull << 33, 2048ull * 2048ull * 40ull};
    int rc = 0;
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(d_bad, 0, 8);
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, mode, counts[mode], 12345ull + mode, d_bad, d_ex);
        hipDeviceSynchronize();
        unsigned long long bad; float ex[4];
        hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(ex, d_ex, 16, hipMemcpyDeviceToHost);
        printf("%-36s %llu pairs, %llu mismatches", names[mode], (unsigned long long)counts[mode], bad);
        if (bad) { printf("   e.g. f=%a Z=%a: %a vs %a", ex[0], ex[1], ex[2], ex[3]); rc = 1; }
        printf("\n");
    }
    return rc;
}
