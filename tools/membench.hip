// membench: what a read-only pass over N bytes costs on this GPU with the launch geometries the block scan uses.
// Gives the practical ceiling the K4 roofline fraction in DESIGN.md is compared with (the 8 TB/s peak is not
// reachable for a 100 MB transfer that lasts ~20 us: ramp-up and drain are a visible part of it).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/membench tools/membench.hip && gpurun_out/membench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// one wavefront per 4096-byte span (4 x 16 B per lane), like hml_k_compact_scan_keys
__global__ __launch_bounds__(256) void k_span_per_wave(const uint8_t* __restrict__ p, uint64_t n, uint32_t* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t base = wave * 4096u;
    if (base >= n) return;
    const u4* q = reinterpret_cast<const u4*>(p + base) + lane;
    u4 a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 64), c = __builtin_nontemporal_load(q + 128),
       d = __builtin_nontemporal_load(q + 192);
    const uint32_t v = (a.x ^ b.y ^ c.z ^ d.w) + (a.y ^ b.z ^ c.w ^ d.x);
    if (v == 0x12345678u) sink[wave & 1023u] = v;   // never true for the fill pattern; keeps the loads alive
}

// one wavefront per BYTES-byte span, 16 B per lane per load, all loads issued first
template <int LOADS>
__global__ __launch_bounds__(256) void k_wide(const uint8_t* __restrict__ p, uint64_t n, uint32_t* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t base = wave * (uint64_t)(LOADS * 1024);
    if (base >= n) return;
    const u4* q = reinterpret_cast<const u4*>(p + base) + lane;
    u4 r[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) r[i] = __builtin_nontemporal_load(q + i * 64);
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) v += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
    if (v == 0x12345678u) sink[wave & 1023u] = v;
}

// grid-stride persistent variant
__global__ __launch_bounds__(256) void k_persistent(const uint8_t* __restrict__ p, uint64_t n, uint32_t* __restrict__ sink) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const u4* q = reinterpret_cast<const u4*>(p);
    const uint64_t m = n / 16;
    uint32_t v = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += 4 * stride) {
        u4 a = __builtin_nontemporal_load(q + i);
        u4 b = (i + stride < m) ? __builtin_nontemporal_load(q + i + stride) : u4{0, 0, 0, 0};
        u4 c = (i + 2 * stride < m) ? __builtin_nontemporal_load(q + i + 2 * stride) : u4{0, 0, 0, 0};
        u4 d = (i + 3 * stride < m) ? __builtin_nontemporal_load(q + i + 3 * stride) : u4{0, 0, 0, 0};
        v += (a.x ^ b.y ^ c.z ^ d.w) + (a.y ^ b.z ^ c.w ^ d.x);
    }
    if (v == 0x12345678u) sink[threadIdx.x] = v;
}

__global__ void k_empty() {}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <typename F>
static float time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1000.0f / reps;
}

int main() {
    const uint64_t sizes[] = {100000000ull, 400000000ull, 1600000000ull};
    uint32_t* sink = nullptr;
    CHK(hipMalloc(&sink, 4096));
    // a second buffer that is touched between repetitions would defeat the 256 MB infinity cache; instead the
    // buffers rotate (4 copies) so that a repetition never re-reads what the previous one just brought in
    for (uint64_t n : sizes) {
        const int copies = n <= 400000000ull ? 4 : 2;
        std::vector<uint8_t*> bufs(copies);
        for (auto& b : bufs) { CHK(hipMalloc(&b, n + 65536)); CHK(hipMemset(b, 0x5a, n + 65536)); }
        CHK(hipDeviceSynchronize());
        int rot = 0;
        auto report = [&](const char* name, float us) {
            printf("%-28s N=%10llu B  %8.2f us  %7.1f GB/s\n", name, (unsigned long long)n, us, (double)n / us * 1e-3);
        };
        const uint32_t waves4k = (uint32_t)((n + 4095) / 4096);
        report("span-per-wave 4KB", time_us([&] { hipLaunchKernelGGL(k_span_per_wave, dim3((waves4k + 3) / 4), dim3(256), 0, 0, bufs[rot++ % copies], n, sink); }, 200));
        report("wave 8KB (8 loads)", time_us([&] { const uint32_t w = (uint32_t)((n + 8191) / 8192); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wide<8>), dim3((w + 3) / 4), dim3(256), 0, 0, bufs[rot++ % copies], n, sink); }, 200));
        report("wave 16KB (16 loads)", time_us([&] { const uint32_t w = (uint32_t)((n + 16383) / 16384); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wide<16>), dim3((w + 3) / 4), dim3(256), 0, 0, bufs[rot++ % copies], n, sink); }, 200));
        report("wave 2KB (2 loads)", time_us([&] { const uint32_t w = (uint32_t)((n + 2047) / 2048); hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wide<2>), dim3((w + 3) / 4), dim3(256), 0, 0, bufs[rot++ % copies], n, sink); }, 200));
        for (int g : {1024, 2048, 4096, 8192})  {
            char name[64]; snprintf(name, sizeof name, "persistent grid %d", g);
            report(name, time_us([&] { hipLaunchKernelGGL(k_persistent, dim3(g), dim3(256), 0, 0, bufs[rot++ % copies], n, sink); }, 200));
        }
        for (auto b : bufs) hipFree(b);
    }
    printf("empty kernel back-to-back: %.2f us\n", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); }, 1000));
    hipFree(sink);
    return 0;
}
