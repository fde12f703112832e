// dispatch_bench: what a dependent kernel boundary costs on this stack, and what it depends on.
// Sequences of kernels on one stream, time per launch from HIP events over 2000 launches.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/dispatch_bench tools/dispatch_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <functional>

__global__ void k_empty() {}
__global__ void k_empty_args(void* a, void* b, void* c, void* d, void* e, void* f, void* g, void* h, void* i, void* j, void* k, void* l,
                             int m, int n, int o) {}
__global__ __launch_bounds__(256) void k_write(uint32_t* p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}
__global__ __launch_bounds__(256) void k_write_nt(uint32_t* p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(i, p + i);
}
__global__ __launch_bounds__(256) void k_read(const uint32_t* p, uint32_t n, uint32_t* sink) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && p[i] == 0xdeadbeefu) sink[0] = 1;
}
__global__ __launch_bounds__(1024) void k_lds(uint32_t* sink) {
    __shared__ uint32_t buf[10240];
    buf[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (buf[(threadIdx.x * 7) & 1023] == 0xdeadbeefu) sink[0] = 1;
}

static float per_launch_us(const std::function<void()>& seq, int launches_per_seq, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) seq();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) seq();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / (reps * launches_per_seq);
}

int main() {
    uint32_t *buf = nullptr, *sink = nullptr;
    const uint32_t N = 64u << 20;   // 256 MB of u32
    (void)hipMalloc(&buf, (size_t)N * 4); (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 0, (size_t)N * 4);
    auto E = [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); };
    printf("empty x1                          : %6.2f us/launch\n", per_launch_us(E, 1, 2000));
    printf("empty, 15 arguments               : %6.2f us/launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_empty_args, dim3(1), dim3(64), 0, 0, buf, buf, buf, buf, buf, buf, buf, buf, buf, buf, buf, buf, 1, 2, 3); }, 1, 2000));
    printf("empty, grid 1024 x 256            : %6.2f us/launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0); }, 1, 2000));
    printf("one workgroup 1024 thr, 40 KB LDS : %6.2f us/launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_lds, dim3(1), dim3(1024), 0, 0, sink); }, 1, 2000));
    for (uint32_t mb : {1u, 4u, 16u, 64u}) {
        const uint32_t n = mb << 18;   // mb MB of u32
        const float w = per_launch_us([&] { hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, 0, buf, n); }, 1, 1000);
        const float we = per_launch_us([&] { hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, 0, buf, n); E(); }, 2, 1000);
        const float wnt = per_launch_us([&] { hipLaunchKernelGGL(k_write_nt, dim3(n / 256), dim3(256), 0, 0, buf, n); }, 1, 1000);
        const float wr = per_launch_us([&] { hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, 0, buf, n); hipLaunchKernelGGL(k_read, dim3(n / 256), dim3(256), 0, 0, buf, n, sink); }, 2, 1000);
        const float r = per_launch_us([&] { hipLaunchKernelGGL(k_read, dim3(n / 256), dim3(256), 0, 0, buf, n, sink); }, 1, 1000);
        printf("%3u MB: write %6.2f | write+empty pair %6.2f (per launch) | nontemporal write %6.2f | read %6.2f | write+read pair %6.2f (per launch)\n", mb, w, we, wnt, r, wr);
    }
    (void)hipFree(buf); (void)hipFree(sink);
    return 0;
}
