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

// Stores of 2 bytes per lane over N bytes: FULL = a wavefront's store covers one whole 128-byte line; otherwise each half-wavefront writes a 64-byte run
// into a line of its own and the other halves of those lines follow GAP iterations later (the states kernel of the weakly compressed sweep: 32 rows of a
// chunk per batch, the neighbouring 32 rows one batch later).  Does the FETCH_SIZE counter see reads for such half-line stores?
template <bool FULL, int GAP>
__global__ __launch_bounds__(64) void k_write(uint16_t* __restrict__ p, uint64_t n_elems) {
    const uint32_t lane = threadIdx.x;
    const uint64_t lines = n_elems / 64u;                     // 128-byte lines
    const uint64_t per = lines / gridDim.x;                   // lines per wavefront (contiguous range)
    const uint64_t l0 = (uint64_t)blockIdx.x * per;
    if (FULL) {
        for (uint64_t l = 0; l < per; ++l) p[(l0 + l) * 64u + lane] = (uint16_t)(l + lane);
    } else {
        // pairs of lines (A, B): first the lower half of A (lanes 0-31) and of B (lanes 32-63), GAP pairs later their upper halves
        const uint64_t pairs = per / 2u;
        for (uint64_t i = 0; i < pairs + GAP; ++i) {
            if (i < pairs) { const uint64_t line = l0 + 2u * i + (lane >> 5); p[line * 64u + (lane & 31u)] = (uint16_t)(i + lane); }
            if (i >= (uint64_t)GAP) { const uint64_t j = i - GAP; const uint64_t line = l0 + 2u * j + (lane >> 5); p[line * 64u + 32u + (lane & 31u)] = (uint16_t)(j + lane); }
        }
    }
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
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_write<true, 0>), dim3(8192), dim3(64), 0, 0, reinterpret_cast<uint16_t*>(d), bytes / 2);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_write<false, 1>), dim3(8192), dim3(64), 0, 0, reinterpret_cast<uint16_t*>(d), bytes / 2);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_write<false, 16>), dim3(8192), dim3(64), 0, 0, reinterpret_cast<uint16_t*>(d), bytes / 2);
    }
    hipDeviceSynchronize();
    return 0;
}
