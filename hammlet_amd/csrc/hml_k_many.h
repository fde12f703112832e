// Several chains of one GPU in ONE launch per kernel (hml_iterate_many of include/hml.h).
// A single strongly compressed chain is bound by latency - six dependent launches of a few microseconds each - and uses a
// few percent of the machine; chains on separate streams do overlap on the GPU, but every launch costs the host 3-4 us
// under the runtime's lock, so eight chains with their own host threads reach 1.7 times one chain (profiles/
// round2_chains_per_gpu.txt).  Here the chain is the grid's y dimension: each kernel of the sweep is the same device
// function as the single-chain kernel (hml_b_*), called with the pointers of chain blockIdx.y, so one host thread pays
// for one set of launches whatever the number of chains, and a chain's results are bit for bit those of running alone.
// Reference: the sweep is HMM.hpp:99-121 (sampleHMM) per chain; the reference itself runs one chain (main.cpp:108).
#ifndef HML_K_MANY_H
#define HML_K_MANY_H

#include "hml_k_backward.h"
#include "hml_k_blocks.h"
#include "hml_k_forward.h"
#include "hml_k_params.h"

struct hml_chain_dev {
    // construction (read-only in a sweep)
    const uint8_t* summary;
    const float* w;
    const float2* ia;
    int32_t key_base;
    uint32_t n_spans;
    // block structure
    uint16_t* stage;
    uint32_t *span_count, *coarse1, *starts, *host_B;
    float2* bstat;
    // sweep buffers
    hml_model* mdl;
    float *em, *gsc, *rows, *entry, *exitv;
    uint32_t *fb, *redo, *touched;
    unsigned long long *smap, *cmap;
    uint8_t* bentry;
    int16_t* q;
    double* partial;
    int32_t* diff;
    uint32_t* boundary;
    hml_layout lay;   // the chain's own chunk-transposed layout (its stride follows ITS block capacity, hml_ctx.hpp)
};

// The chains of one launch as a KERNEL ARGUMENT (up to sixteen: 3.7 of the 4 KB a launch may carry; round 4): the pointers come with the launch instead of through one
// more dependent load from an array in device memory - the first of the six or seven memory round trips that these short kernels
// consist of, and beside another group's kernels every one of them takes three times as long.
#define HML_MANY_ARG_CHAINS 16
struct hml_many_args { hml_chain_dev c[HML_MANY_ARG_CHAINS]; };
static_assert(sizeof(hml_many_args) + 64 <= 4096, "kernel arguments of the batched kernels");

HML_KERNEL __launch_bounds__(256) void hml_m_compact_scan_summary(const hml_chain_dev* __restrict__ cs, uint32_t T) {
    const hml_chain_dev& c = cs[blockIdx.y];
    hml_b_compact_scan_summary(c.summary, c.w, T, c.mdl, 0.0f, 0, c.key_base, c.stage, c.span_count, c.coarse1);
}
HML_KERNEL __launch_bounds__(256) void hml_m_compact_scatter(const hml_chain_dev* __restrict__ cs, uint32_t T) {
    const hml_chain_dev& c = cs[blockIdx.y];
    hml_b_compact_scatter(c.stage, c.span_count, c.coarse1, c.n_spans, T, c.mdl, c.starts, c.host_B);
}
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_m_stats_emission(const hml_chain_dev* __restrict__ cs, int with_gsc) {
    const hml_chain_dev& c = cs[blockIdx.y];
    hml_b_stats_emission<K>(c.ia, c.starts, c.mdl, c.bstat, c.em, with_gsc ? c.gsc : nullptr, nullptr, 0, c.lay);
}
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_m_forward(const hml_many_args a, int with_gsc, int L) {
    const hml_chain_dev& c = a.c[blockIdx.y];
    hml_b_forward<K>(c.em, with_gsc ? c.gsc : nullptr, c.mdl, c.rows, nullptr, c.entry, c.exitv, c.fb, L, c.lay);
}
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_m_backward_maps(const hml_many_args a, int with_gsc, int L) {
    const hml_chain_dev& c = a.c[blockIdx.y];
    hml_b_backward_maps2<K>(c.rows, c.mdl, c.smap, c.cmap, c.lay, c.entry, c.exitv, c.redo, L, with_gsc ? nullptr : c.starts, c.mdl);   // (two rows per lane)
}
template <int K>
HML_KERNEL __launch_bounds__(1024) void hml_m_backward_chain(const hml_many_args a, int with_gsc, int L) {
    const hml_chain_dev& c = a.c[blockIdx.y];
    hml_b_backward_chain<K>(c.cmap, c.mdl, c.bentry, c.em, with_gsc ? c.gsc : nullptr, c.rows, nullptr, c.entry, c.exitv, c.fb, c.redo, c.touched,
                            c.smap, L, c.lay, 3, 0, with_gsc ? nullptr : c.starts);
}
template <int K>
HML_KERNEL __launch_bounds__(256) void hml_m_counts(const hml_many_args a) {
    const hml_chain_dev& c = a.c[blockIdx.y];
    hml_b_counts<K, true, false>(c.q, c.starts, c.bstat, c.mdl, c.partial, c.smap, c.bentry);
}
// (only the chains whose bit is set record this sweep: a chain may have its marginals switched off)
HML_KERNEL __launch_bounds__(256) void hml_m_record(const hml_many_args a, unsigned long long chains_recording) {
    if (!((chains_recording >> blockIdx.y) & 1ull)) return;
    const hml_chain_dev& c = a.c[blockIdx.y];
    hml_b_record(c.q, c.starts, c.mdl, c.diff, c.boundary);
}
template <int K>
HML_KERNEL __launch_bounds__(1024) void hml_m_params(const hml_many_args a) {
    const hml_chain_dev& c = a.c[blockIdx.y];
    hml_b_params<K, true, true>(c.mdl, c.partial, 0, (int)blockIdx.x, (int)gridDim.x);
}

#endif
