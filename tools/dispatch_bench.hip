// dispatch_bench: what a dependent kernel boundary costs on this stack, and what it depends on.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/dispatch_bench tools/dispatch_bench.hip
// Part 1 (per-launch times from HIP events over thousands of launches): empty kernels, argument count, grid size,
// bytes left dirty by the predecessor.
// Part 2 (in-kernel clocks): a six-stage pipeline with the grid shapes of one Gibbs sweep; every stage stamps
// wall_clock64 (100 MHz) when its first workgroup starts and when its last one ends, so the GAP between a stage's
// end and its successor's start is measured directly, for each variant of what the stages do at their end:
//   nothing | plain stores of N bytes | + one system-scope store to a host-mapped word | 17 kernel arguments.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <vector>

__global__ void k_empty() {}
__global__ void k_empty_args(void* a, void* b, void* c, void* d, void* e, void* f, void* g, void* h, void* i, void* j, void* k, void* l,
                             int m, int n, int o) {}
__global__ __launch_bounds__(256) void k_write(uint32_t* p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
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

template <int NT, int WORDS>
__global__ __launch_bounds__(NT) void k_shape(uint32_t* sink) {
    __shared__ uint32_t buf[WORDS > 0 ? WORDS : 1];
    buf[threadIdx.x % (WORDS > 0 ? WORDS : 1)] = threadIdx.x;
    __syncthreads();
    if (buf[(threadIdx.x * 7) % (WORDS > 0 ? WORDS : 1)] == 0xdeadbeefu) sink[0] = 1;
}

static float per_launch_us(hipStream_t s, const std::function<void()>& seq, int launches_per_seq, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) seq();
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) seq();
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / (reps * launches_per_seq);
}

// ---- part 2
struct stamp { unsigned long long start, end; };
#define MAXWG 1024

// a stage: reads what the previous stage wrote (`src`, n_src words spread over the grid), writes n_dst words
template <int NARGS>
__device__ __forceinline__ void stage_body(stamp* ts, int idx, const uint32_t* src, uint32_t n_src, uint32_t* dst, uint32_t n_dst,
                                           uint32_t* host_word, uint32_t* sink) {
    // every workgroup stamps its own slot (plain stores: contended atomics on one word take ~11 ns each and would
    // dominate a 763-workgroup stage); the host takes the minimum / maximum over the slots
    const unsigned long long t_in = wall_clock64();
    const uint32_t nthr = gridDim.x * blockDim.x, gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t i = gid; i < n_src; i += nthr) acc += src[i];
    for (uint32_t i = gid; i < n_dst; i += nthr) dst[i] = acc + i;
    if (acc == 0xdeadbeefu) sink[0] = acc;
    if (host_word && gid == 0) __hip_atomic_store(host_word, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (threadIdx.x == 0) { ts[(size_t)idx * MAXWG + blockIdx.x].start = t_in; ts[(size_t)idx * MAXWG + blockIdx.x].end = wall_clock64(); }
}
__global__ void k_stage(stamp* ts, int idx, const uint32_t* src, uint32_t n_src, uint32_t* dst, uint32_t n_dst, uint32_t* host_word, uint32_t* sink) {
    stage_body<8>(ts, idx, src, n_src, dst, n_dst, host_word, sink);
}
__global__ void k_stage17(stamp* ts, int idx, const uint32_t* src, uint32_t n_src, uint32_t* dst, uint32_t n_dst, uint32_t* host_word, uint32_t* sink,
                          void* a, void* b, void* c, void* d, void* e, void* f, void* g, uint32_t h, uint64_t l) {
    stage_body<17>(ts, idx, src, n_src, dst, n_dst, host_word, sink);
}

int main() {
    uint32_t *buf = nullptr, *sink = nullptr;
    const uint32_t N = 64u << 20;   // 256 MB of u32
    (void)hipMalloc(&buf, (size_t)N * 4); (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 0, (size_t)N * 4);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    auto E = [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); };
    printf("== part 1: per-launch time, back-to-back on one (non-blocking) stream\n");
    printf("empty x1                          : %6.2f us/launch\n", per_launch_us(st, E, 1, 4000));
    printf("empty, 15 arguments               : %6.2f us/launch\n", per_launch_us(st, [&] { hipLaunchKernelGGL(k_empty_args, dim3(1), dim3(64), 0, st, buf, buf, buf, buf, buf, buf, buf, buf, buf, buf, buf, buf, 1, 2, 3); }, 1, 4000));
    printf("empty, grid 1024 x 256            : %6.2f us/launch\n", per_launch_us(st, [&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st); }, 1, 4000));
    printf("one workgroup 1024 thr, 40 KB LDS : %6.2f us/launch\n", per_launch_us(st, [&] { hipLaunchKernelGGL(k_lds, dim3(1), dim3(1024), 0, st, sink); }, 1, 4000));
    printf("one workgroup, by shape: 1024 thr no LDS %5.2f | 1024 thr 40 KB %5.2f | 256 thr no LDS %5.2f | 256 thr 40 KB %5.2f | 64 thr 40 KB %5.2f us/launch\n",
           per_launch_us(st, [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_shape<1024, 16>), dim3(1), dim3(1024), 0, st, sink); }, 1, 4000),
           per_launch_us(st, [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_shape<1024, 10240>), dim3(1), dim3(1024), 0, st, sink); }, 1, 4000),
           per_launch_us(st, [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_shape<256, 16>), dim3(1), dim3(256), 0, st, sink); }, 1, 4000),
           per_launch_us(st, [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_shape<256, 10240>), dim3(1), dim3(256), 0, st, sink); }, 1, 4000),
           per_launch_us(st, [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_shape<64, 10240>), dim3(1), dim3(64), 0, st, sink); }, 1, 4000));
    for (uint32_t mb : {1u, 4u, 16u, 64u}) {
        const uint32_t n = mb << 18;   // mb MB of u32
        const float w = per_launch_us(st, [&] { hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, st, buf, n); }, 1, 1000);
        const float we = per_launch_us(st, [&] { hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, st, buf, n); E(); }, 2, 1000);
        const float wr = per_launch_us(st, [&] { hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, st, buf, n); hipLaunchKernelGGL(k_read, dim3(n / 256), dim3(256), 0, st, buf, n, sink); }, 2, 1000);
        const float r = per_launch_us(st, [&] { hipLaunchKernelGGL(k_read, dim3(n / 256), dim3(256), 0, st, buf, n, sink); }, 1, 1000);
        printf("%3u MB: write %6.2f | write+empty pair %6.2f (per launch) | read %6.2f | write+read pair %6.2f (per launch)\n", mb, w, we, r, wr);
    }

    printf("== part 2: six dependent stages with the grid shapes of one sweep; gap = start(stage k+1) - end(stage k), in-kernel clocks\n");
    const int NS = 6;
    dim3 grids[NS] = {dim3(763), dim3(172), dim3(688), dim3(1), dim3(1024), dim3(1)};
    dim3 blocks[NS] = {dim3(512), dim3(256), dim3(256), dim3(1024), dim3(256), dim3(1024)};
    const char* names[NS] = {"blocks 763x512", "forward 172x256", "maps 688x256", "chain 1x1024", "counts 1024x256", "params 1x1024"};
    stamp* d_ts = nullptr;
    uint32_t* h_word = nullptr; uint32_t* d_word = nullptr;
    (void)hipHostMalloc(&h_word, 64, hipHostMallocMapped);
    (void)hipHostGetDevicePointer((void**)&d_word, h_word, 0);
    const int REPS = 300;
    (void)hipMalloc(&d_ts, sizeof(stamp) * NS * REPS * MAXWG);
    std::vector<stamp> h_all((size_t)NS * REPS * MAXWG), h_ts(NS * REPS);
    struct variant { const char* name; uint32_t words; bool host; bool many_args; bool tiny; };
    const variant vars[] = {
        {"every stage ONE 64-thread workgroup, nothing written", 0u, false, false, true},
        {"stages write nothing", 0u, false, false},
        {"stages write 64 KB each", 16u << 10, false, false},
        {"stages write 1 MB each", 256u << 10, false, false},
        {"stages write 8 MB each", 2048u << 10, false, false},
        {"1 MB each + system-scope store to a host-mapped word in stage 0", 256u << 10, true, false},
        {"1 MB each, 17 kernel arguments", 256u << 10, false, true},
    };
    for (const variant& v : vars) {
        const dim3 g0[NS] = {dim3(763), dim3(172), dim3(688), dim3(1), dim3(1024), dim3(1)};
        const dim3 b0[NS] = {dim3(512), dim3(256), dim3(256), dim3(1024), dim3(256), dim3(1024)};
        for (int k = 0; k < NS; ++k) { grids[k] = v.tiny ? dim3(1) : g0[k]; blocks[k] = v.tiny ? dim3(64) : b0[k]; }
        (void)hipMemset(d_ts, 0, sizeof(stamp) * NS * REPS * MAXWG);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < REPS; ++r) {
            for (int k = 0; k < NS; ++k) {
                uint32_t* dst = buf + (size_t)(k & 1) * (16u << 20);
                const uint32_t* src = buf + (size_t)((k + 1) & 1) * (16u << 20);
                uint32_t* hw = (v.host && k == 0) ? d_word : nullptr;
                if (v.many_args)
                    hipLaunchKernelGGL(k_stage17, grids[k], blocks[k], 0, st, d_ts, r * NS + k, src, std::max(v.words, 1u), dst, v.words, hw, sink,
                                       (void*)buf, (void*)buf, (void*)buf, (void*)buf, (void*)buf, (void*)buf, (void*)buf, 7u, (uint64_t)9);
                else
                    hipLaunchKernelGGL(k_stage, grids[k], blocks[k], 0, st, d_ts, r * NS + k, src, std::max(v.words, 1u), dst, v.words, hw, sink);
            }
        }
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h_all.data(), d_ts, sizeof(stamp) * NS * REPS * MAXWG, hipMemcpyDeviceToHost);
        for (int i = 0; i < NS * REPS; ++i) {
            stamp m{~0ull, 0ull};
            for (int w = 0; w < MAXWG; ++w) { const stamp& q = h_all[(size_t)i * MAXWG + w]; if (q.end) { m.start = std::min(m.start, q.start); m.end = std::max(m.end, q.end); } }
            h_ts[i] = m;
        }
        printf("-- %s: %.2f us per six-stage round (events)\n", v.name, ms * 1000.0f / REPS);
        for (int k = 0; k < NS; ++k) {
            std::vector<double> gap, dur;
            for (int r = 50; r < REPS; ++r) {
                const stamp& cur = h_ts[r * NS + k];
                const stamp& nxt = (k + 1 < NS) ? h_ts[r * NS + k + 1] : h_ts[(r + 1 < REPS ? r + 1 : r) * NS];
                dur.push_back((double)(cur.end - cur.start) * 0.01);
                if (k + 1 < NS || r + 1 < REPS) gap.push_back((double)((long long)nxt.start - (long long)cur.end) * 0.01);
            }
            std::sort(gap.begin(), gap.end()); std::sort(dur.begin(), dur.end());
            printf("   %-16s in-kernel %6.2f us (median) | gap to the next stage %6.2f us (median), %6.2f (90th pct)\n", names[k], dur[dur.size() / 2],
                   gap[gap.size() / 2], gap[gap.size() * 9 / 10]);
        }
    }
    (void)hipFree(buf); (void)hipFree(sink);
    return 0;
}
