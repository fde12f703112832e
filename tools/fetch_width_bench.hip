// fetch_width_bench: what the FETCH_SIZE counter reports for streaming reads of 2, 4, 8 and 16 bytes per lane (consecutive lanes, consecutive
// addresses) - the guide's gfx950 correction (the counter doubled) was established for wide streaming reads; the count pass and the states
// kernel of the weakly compressed sweep read 8 and 2 or 4 bytes per lane.  Every kernel reads the same N bytes once.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/fetch_width_bench tools/fetch_width_bench.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fw -o run -- gpurun_out/fetch_width_bench     (then tools/pmc_summary-style: bytes per launch by kernel name)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ p, uint64_t n_elems, uint32_t* __restrict__ sink) {
    // a workgroup streams chunks of 256 elements, grid-stride: consecutive lanes read consecutive elements; four chunks in flight
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_elems; i += 4u * stride) {
        T v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t j = i + (uint64_t)k * stride;
            v[k] = p[j < n_elems ? j : n_elems - 1u];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned char* b = reinterpret_cast<const unsigned char*>(&v[k]);
#pragma unroll
            for (unsigned q = 0; q < sizeof(T); ++q) acc += b[q];
        }
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;   // never true for the fill pattern; keeps the loads alive
}

struct b16 { uint32_t x, y, z, w; };

template <typename T>
static void run(const char* name, const void* d, uint64_t bytes, uint32_t* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const uint64_t n = bytes / sizeof(T);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_read<T>, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const T*>(d), n, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-28s %zu bytes per lane: %.1f MB in %.1f us = %.2f TB/s\n", name, sizeof(T), bytes / 1e6, best * 1e3, bytes / (best * 1e-3) / 1e12);
}

int main() {
    const uint64_t bytes = 1ull << 30;
    void* d; uint32_t* sink;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    run<uint16_t>("k_read<unsigned short>", d, bytes, sink);
    run<uint32_t>("k_read<unsigned int>", d, bytes, sink);
    run<uint64_t>("k_read<unsigned long>", d, bytes, sink);
    run<b16>("k_read<b16>", d, bytes, sink);
    hipDeviceSynchronize();
    return 0;
}
