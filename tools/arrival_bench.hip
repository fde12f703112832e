// What folding a one-workgroup kernel into its producer buys: a many-workgroup stage A (every workgroup
// reads a little, writes 16 doubles of partial sums) followed by a one-workgroup stage B of 1024 threads that reads all
// partials, reduces them and writes a few words - the shape of hml_k_counts -> hml_k_params and of hml_k_backward_maps ->
// hml_k_backward_chain.  Timed back to back on one stream, hipEvents around REPS rounds:
//   separate : A<<<G,256>>> ; B<<<1,1024>>>                                   (two launches, one dependent boundary)
//   merged   : A'<<<G,256>>> - the workgroup that arrives last (device-scope ticket) runs B's body with its 256 threads
//   merged-1k: A'<<<G/4,1024>>> - four groups per workgroup, the last arrival runs B's body with 1024 threads
//   each with __threadfence() in every wavefront before the ticket, or with ONE lane's acq_rel ticket after the barrier
// for G = 128, 256, 1024 workgroups.  Each round is preceded by a "consumer" stage C<<<64,256>>> that reads B's output, so
// that the chain A -> B -> C -> A ... is dependent like the sweep's.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/arrival_bench tools/arrival_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct shared_t {
    unsigned int ticket;
    unsigned int pad[15];
    double out[16];
};

__device__ __forceinline__ void body_a(const float* __restrict__ in, double* __restrict__ partial, uint32_t g, uint32_t n_groups, int tid,
                                       double* lds /*[4][16]*/) {
    // 256 threads: one load each, wave sums of 16 "states", one double per group and state
    const float v = in[(uint64_t)g * 256u + (uint32_t)tid];
    double acc[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = ((tid & 15) == s) ? (double)v : 0.0;
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const unsigned long long u = __double_as_longlong(acc[s]);
            const unsigned lo = __shfl_xor((unsigned)u, m), hi = __shfl_xor((unsigned)(u >> 32), m);
            acc[s] += __longlong_as_double(((unsigned long long)hi << 32) | lo);
        }
    if ((tid & 63) == 0)
#pragma unroll
        for (int s = 0; s < 16; ++s) lds[(tid >> 6) * 16 + s] = acc[s];
    __syncthreads();
    if (tid < 16) partial[(uint64_t)tid * n_groups + g] = (lds[tid] + lds[16 + tid]) + (lds[32 + tid] + lds[48 + tid]);
}

__device__ __forceinline__ void body_b(const double* __restrict__ partial, uint32_t n_groups, shared_t* sh, int tid, int nthreads, double* lds /*[16][16]*/) {
    // all partials of 16 states, summed by 16 teams of nthreads / 16 threads
    const int per = nthreads / 16, s = tid / per, l = tid % per;
    double a = 0.0;
    for (uint32_t g = (uint32_t)l; g < n_groups; g += (uint32_t)per) a += partial[(uint64_t)s * n_groups + g];
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        const unsigned long long u = __double_as_longlong(a);
        const unsigned lo = __shfl_xor((unsigned)u, m), hi = __shfl_xor((unsigned)(u >> 32), m);
        a += __longlong_as_double(((unsigned long long)hi << 32) | lo);
    }
    if ((l & 15) == 0) lds[s * 16 + (l >> 4)] = a;
    __syncthreads();
    if (tid < 16) {
        double t = 0.0;
        for (int i = 0; i < per / 16; ++i) t += lds[tid * 16 + i];
        sh->out[tid] = t;
    }
}

__global__ __launch_bounds__(256) void k_a(const float* in, double* partial, const shared_t* sh_ro) {
    __shared__ double lds[64];
    (void)sh_ro;
    body_a(in, partial, blockIdx.x, gridDim.x, threadIdx.x, lds);
}
__global__ __launch_bounds__(1024) void k_b(const double* partial, uint32_t n_groups, shared_t* sh) {
    __shared__ double lds[256];
    body_b(partial, n_groups, sh, threadIdx.x, 1024, lds);
}
__global__ __launch_bounds__(256) void k_c(float* in, const shared_t* sh) {   // the next round's input depends on B's output
    const double o = sh->out[threadIdx.x & 15];
    in[blockIdx.x * 256u + threadIdx.x] = (float)(o * 1e-30) + 1.0f;
}
template <int TEAMS, bool FENCE_ALL>
__global__ __launch_bounds__(256 * TEAMS) void k_merged(const float* in, double* partial, shared_t* sh) {
    __shared__ double lds[TEAMS][64];
    __shared__ double lds_b[256];
    __shared__ unsigned int last;
    const int team = threadIdx.x >> 8, tid = threadIdx.x & 255;
    body_a(in, partial, blockIdx.x * TEAMS + team, gridDim.x * TEAMS, tid, lds[team]);
    if (FENCE_ALL) __threadfence();   // every wavefront writes its L2 back ...
    __syncthreads();
    // ... or one lane releases for the workgroup (the barrier orders the others' stores before it)
    if (threadIdx.x == 0) last = (__hip_atomic_fetch_add(&sh->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) sh->ticket = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    body_b(partial, gridDim.x * TEAMS, sh, threadIdx.x, 256 * TEAMS, lds_b);
}

int main() {
    const int REPS = 3000;
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* in; double* partial; shared_t* sh;
    CHK(hipMalloc(&in, 1024 * 256 * sizeof(float)));
    CHK(hipMalloc(&partial, 16 * 1024 * sizeof(double)));
    CHK(hipMalloc(&sh, sizeof(shared_t)));
    CHK(hipMemset(sh, 0, sizeof(shared_t)));
    std::vector<float> h(1024 * 256, 1.0f);
    CHK(hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("%-10s %12s %12s %12s %14s %14s   (us per round A -> B -> C)\n", "groups", "separate", "merged-256", "merged-1024", "m-256,1 fence", "m-1024,1 fence");
    for (unsigned G : {128u, 256u, 1024u}) {
        float ms[5] = {0, 0, 0, 0, 0};
        double check[5] = {0, 0, 0, 0, 0};
        for (int variant = 0; variant < 5; ++variant) {
            for (int pass = 0; pass < 2; ++pass) {   // warm-up pass, timed pass
                CHK(hipEventRecord(e0, st));
                for (int r = 0; r < REPS; ++r) {
                    if (variant == 0) {
                        hipLaunchKernelGGL(k_a, dim3(G), dim3(256), 0, st, in, partial, sh);
                        hipLaunchKernelGGL(k_b, dim3(1), dim3(1024), 0, st, partial, G, sh);
                    } else if (variant == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_merged<1, true>), dim3(G), dim3(256), 0, st, in, partial, sh);
                    else if (variant == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_merged<4, true>), dim3(G / 4), dim3(1024), 0, st, in, partial, sh);
                    else if (variant == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_merged<1, false>), dim3(G), dim3(256), 0, st, in, partial, sh);
                    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_merged<4, false>), dim3(G / 4), dim3(1024), 0, st, in, partial, sh);
                    hipLaunchKernelGGL(k_c, dim3(64), dim3(256), 0, st, in, sh);
                }
                CHK(hipEventRecord(e1, st));
                CHK(hipEventSynchronize(e1));
                CHK(hipEventElapsedTime(&ms[variant], e0, e1));
            }
            shared_t hs;
            CHK(hipMemcpy(&hs, sh, sizeof hs, hipMemcpyDeviceToHost));
            check[variant] = hs.out[0] + hs.out[15];
        }
        printf("%-10u %12.2f %12.2f %12.2f %14.2f %14.2f   (sums %.0f %.0f %.0f %.0f %.0f)\n", G, 1e3 * ms[0] / REPS, 1e3 * ms[1] / REPS, 1e3 * ms[2] / REPS,
               1e3 * ms[3] / REPS, 1e3 * ms[4] / REPS, check[0], check[1], check[2], check[3], check[4]);
    }
    return 0;
}
