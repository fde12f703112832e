// libhammlet_hip.so - C ABI (include/hml.h) over the gfx950 kernels.  Host side only orchestrates:
// allocation, launches on the context's stream, and the few inherently sequential one-time steps of
// the reference's driver (noise estimate src/main.cpp:303-311, autoPrior src/AutoPriors.hpp:86-110).
#include "hml_capi_shared.hpp"

static thread_local std::string g_err;
int hml_set_err(int code, const std::string& msg) { g_err = msg; return code; }

// the table of K states, or nullptr (outside [2, 16]; a development build knows one K only: -DHML_ONLY_K=5, tools/dev_build.py)
static const hml_ktab* ktab(int K);
#define HML_KTAB(KV, tab) const hml_ktab* tab = ktab(KV); if (!tab) return set_err(HML_ERR_ARG, "number of states must be in [2,16]")

std::atomic<int> hml_live_ctx[64];
int hml_ctx_bind(hml_ctx* c) { return ctx_bind(c); }
int hml_ctx_fetch_model(hml_ctx* c, hml_model* out) { return fetch_model(c, out); }

// ------------------------------------------------------------------------------------------------
// hml_debug_eval: evaluates one of the shared host/device functions on the GPU (parity probe for
// hml_math.h / hml_dist.h: the tests compare with the same function compiled by gcc).
HML_KERNEL void hml_k_debug_eval(int fn, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                 uint64_t n, uint64_t seed) {
    if (fn == 24) {   // gamma draw with ONE active lane per wavefront (divergence-free control)
        const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
        if ((threadIdx.x & 63) != 0) return;
        for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n; i += nw) {
            hml_dev_src src;
            src.s = hml_stream_open(hml_make_key(seed, 0), HML_KIND_THETA, seed, (uint32_t)i);
            out[i] = hml_gamma_f32<hml_devmath>(src, a[i], b[i]);
        }
        return;
    }
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float x = a[i], y = b ? b[i] : 0.0f;
        float r = 0.0f;
        switch (fn) {
            case 0: r = hml_expf(x); break;
            case 1: r = hml_logf(x); break;
            case 2: r = hml_powf_unit(x, y); break;
            case 3: r = HML_SQRTF(x); break;
            case 4: r = x / y; break;
            case 5: {   // gamma(alpha = x, beta = y) from sub-stream (THETA, epoch = seed, index = i)
                hml_dev_src src;
                src.s = hml_stream_open(hml_make_key(seed, 0), HML_KIND_THETA, seed, (uint32_t)i);
                r = hml_gamma_f32<hml_devmath>(src, x, y);
            } break;
            case 6: {   // normal(mean = x, sd = y)
                hml_dev_src src;
                src.s = hml_stream_open(hml_make_key(seed, 0), HML_KIND_PI, seed, (uint32_t)i);
                hml_normal_f32<hml_devmath> nd;
                r = nd.draw(src, x, y);
            } break;
            case 7: r = (float)hml_log((double)x); break;
            case 8: r = (float)hml_exp_nonpos((double)x); break;
            case 9: r = (float)((double)x / (double)y); break;
            case 40: r = hml_glibc_expf(x); break;            // hml_math_glibc.h: the reference-compatible mode's arithmetic
            case 41: r = hml_glibc_logf(x); break;
            case 42: r = hml_glibc_powf_unit(x, y); break;
            case 43: { const double Zd = (double)y; r = hml_tr2_quotient(x, Zd, hml_tr2_reciprocal(Zd)); } break;   // x / y through the double reciprocal (hml_k_trellis_rows.h)
            case 10: r = (float)(1.0 / (double)x); break;
            case 12: case 13: case 14: case 15: case 16: case 17: case 18: case 19: case 20: case 21: case 22: case 23: {
                hml_dev_src src; src.s = hml_stream_open(hml_make_key(seed, 0), HML_KIND_THETA, seed, (uint32_t)i);
                const float alpha = x, beta = y; (void)beta;
                const float malpha = alpha < 1.0f ? alpha + 1.0f : alpha;
                const float a1 = malpha - 1.0f / 3.0f;
                const float a2 = 1.0f / hml_devmath::sqrtf_(9.0f * a1);
                hml_normal_f32<hml_devmath> nd;
                float n = nd.draw(src, 0.0f, 1.0f);
                float v = 1.0f + a2 * n;
                float v3 = v * v * v;
                float u = hml_canonical_f32(src);
                const bool c1 = (double)u > (double)1.0f - 0.0331 * (double)n * (double)n * (double)n * (double)n;
                const bool c2 = ((double)hml_devmath::logf_(u) > (0.5 * (double)n * (double)n + (double)a1 * ((1.0 - (double)v3) + (double)hml_devmath::logf_(v3))));
                float n2 = nd.draw(src, 0.0f, 1.0f);
                float vb = 1.0f + a2 * n2;
                float vb3 = vb * vb * vb;
                float ub = hml_canonical_f32(src);
                const bool d1 = (double)ub > (double)1.0f - 0.0331 * (double)n2 * (double)n2 * (double)n2 * (double)n2;
                const bool d2 = ((double)hml_devmath::logf_(ub) > (0.5 * (double)n2 * (double)n2 + (double)a1 * ((1.0 - (double)vb3) + (double)hml_devmath::logf_(vb3))));
                r = fn == 12 ? n : fn == 13 ? v3 : fn == 14 ? u : fn == 15 ? (float)(c1 ? 1 : 0) + 2.0f * (c2 ? 1 : 0) : fn == 16 ? n2 : fn == 17 ? a2
                    : fn == 18 ? vb3 : fn == 19 ? ub : fn == 20 ? (float)(d1 ? 1 : 0) + 2.0f * (d2 ? 1 : 0) : fn == 21 ? (float)src.s.n : 0.0f;
                if (fn == 22) r = nd.draw(src, 0.0f, 1.0f);   // third normal (fresh pair)
                if (fn == 23) { hml_normal_f32<hml_devmath> nf; r = nf.draw(src, 0.0f, 1.0f); r = nf.draw(src, 0.0f, 1.0f); }  // saved of the fresh pair
            } break;
            case 30: case 31: {   // categorical draw over K = 16 / 5 weights a[i..i+K), u = b[i]: out = {combined, literal, unsure}
                r = 0.0f;
                const int Kc = fn == 30 ? 16 : 5;
                if (i % Kc == 0 && i + Kc <= n) {
                    bool unsure = false;
                    int fastr, lit;
                    if (fn == 30) { float w[16]; for (int j = 0; j < 16; ++j) w[j] = a[i + j]; fastr = hml_categorical_k_fast<16>(w, (double)y, unsure); lit = hml_categorical_k_literal<16>(w, (double)y); if (unsure) fastr = lit; }
                    else { float w[5]; for (int j = 0; j < 5; ++j) w[j] = a[i + j]; fastr = hml_categorical_k_fast<5>(w, (double)y, unsure); lit = hml_categorical_k_literal<5>(w, (double)y); if (unsure) fastr = lit; }
                    out[i] = (float)fastr; out[i + 1] = (float)lit; out[i + 2] = unsure ? 1.0f : 0.0f;
                }
                continue;
            }
            case 11: { const double d = HML_SQRT((double)x * 1.0000001); r = (float)((d - (double)(float)d) * 1e9); } break;
        }
        out[i] = r;
    }
}

extern "C" {

const char* hml_last_error(void) { return g_err.c_str(); }
uint32_t hml_abi_version(void) { return 3; }   // 2: hml_stats.fused_fallbacks, hml_allreduce_marginals_perm, hml_pool_permutation, option "compat"; 3: hml_attach_observations
const char* hml_device_arch(void) { return "gfx950"; }
int hml_device_count(int* n) {
    if (!n) return set_err(HML_ERR_ARG, "null argument");
    HIPCHK(hipGetDeviceCount(n));
    return 0;
}

static void free_all(hml_ctx* c);

// allocations of a fresh context; on failure the caller releases whatever was created
static int create_inner(hml_ctx* c, void* stream) {
    HIPCHK(hipSetDevice(c->device));
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    HIPCHK(hipMalloc(&c->d_mdl, sizeof(hml_model)));
    HIPCHK(hipMemsetAsync(c->d_mdl, 0, sizeof(hml_model), c->stream));
    HIPCHK(hipHostMalloc(&c->h_B, 4 * sizeof(uint32_t), hipHostMallocMapped));
    c->h_B[0] = 0; c->h_B[1] = 0; c->h_B[2] = 0; c->h_B[3] = 0;
    HIPCHK(hipHostGetDevicePointer((void**)&c->d_hB, c->h_B, 0));
    return 0;
}

int hml_create(hml_ctx** out, int device, uint64_t seed, uint32_t chain_id, void* stream) {
    if (!out) return set_err(HML_ERR_ARG, "null output pointer");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (n <= 0) return set_err(HML_ERR_HIP, "no HIP device available: the MI355X kernels cannot run (there is no CPU fallback)");
    if (device < 0 || device >= n) return set_err(HML_ERR_ARG, "device index out of range");
    hml_ctx* c = new hml_ctx();
    c->device = device; c->seed = seed; c->chain = chain_id;
    if (int r = create_inner(c, stream)) {
        free_all(c);
        if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
        delete c;
        return r;
    }
    if (const char* e = getenv("HML_FWD_CHUNK")) { int l = std::max(1, atoi(e)); int sh = 0; while ((1 << (sh + 1)) <= l && sh < 10) ++sh; c->fwdL = 1 << sh; }
    if (const char* e = getenv("HML_FWD_WARMUP")) c->fwdW = c->fwdW_init = std::max(0, atoi(e));
    if (const char* e = getenv("HML_FWD_BURNIN_SWEEPS")) c->fwd_burnin_sweeps = (uint32_t)std::max(0, atoi(e));
    if (const char* e = getenv("HML_FWD_QUIET")) c->fwd_quiet_need = (uint32_t)std::max(1, atoi(e));
    if (const char* e = getenv("HML_FWD_CHUNK_MANY")) { int l = std::max(1, atoi(e)); int sh = 0; while ((1 << (sh + 1)) <= l && sh < 6) ++sh; c->fwdL_many = 1 << sh; }
    if (const char* e = getenv("HML_FWD_CHUNK_MID")) { int l = std::max(1, atoi(e)); int sh = 0; while ((1 << (sh + 1)) <= l && sh < 10) ++sh; c->fwdL_mid = 1 << sh; }
    if (const char* e = getenv("HML_MID_MIN_BLOCKS")) c->mid_min_blocks = (uint32_t)strtoul(e, nullptr, 10);
    if (const char* e = getenv("HML_FWD_CHUNK_DENSE")) { int l = std::max(1, atoi(e)); int sh = 0; while ((1 << (sh + 1)) <= l && sh < 10) ++sh; c->fwdL_dense = 1 << sh; }
    if (const char* e = getenv("HML_DENSE_MIN_BLOCKS")) c->dense_min_blocks = (uint32_t)strtoul(e, nullptr, 10);
    if (const char* e = getenv("HML_USE_GRAPH")) c->use_graph = atoi(e) != 0;
    if (const char* e = getenv("HML_WEIGHT_KEYS")) c->use_keys = atoi(e) != 0;
    if (const char* e = getenv("HML_FUSED_BLOCKS")) { c->fused_blocks = atoi(e) != 0; c->fused_keep = atoi(e) == 2; }   // 0: a GPU shared with other processes
    if (const char* e = getenv("HML_TRELLIS_FUSED")) c->tre_fused = atoi(e) != 0;
    if (const char* e = getenv("HML_TRELLIS_L")) { const int l = atoi(e); c->tre_L = (l <= 0) ? 0u : (l >= HML_TRE_MAX_L) ? (uint32_t)HML_TRE_MAX_L : (l < 32) ? 32u : (uint32_t)l / 32u * 32u; }   // a multiple of 32
    if (const char* e = getenv("HML_TRELLIS_CKPT")) c->tre_ckpt = atoi(e) != 0;
    if (const char* e = getenv("HML_TRELLIS_REFIT_ROUNDS")) { const int r = atoi(e); c->tre_refit_rounds = r < 0 ? 0u : r > 6 ? 6u : (uint32_t)r; }
    if (const char* e = getenv("HML_STAGE_BITS")) c->stage_bits = atoi(e) != 0;
    if (const char* e = getenv("HML_TRELLIS_ROWS")) c->tre_rows = atoi(e) != 0;   // 0: round 2's first pass (hml_k_trellis_tile) for comparison
    if (const char* e = getenv("HML_LATE_RESCALE")) c->late_rescale = atoi(e) != 0;
    if (const char* e = getenv("HML_PARAMS_SPREAD")) c->params_spread = atoi(e) != 0;
    if (const char* e = getenv("HML_COMPAT_CHUNKS")) c->compat_chunks = atoi(e);   // (1: the sequential form; > 1: that many chunks)
    if (const char* e = getenv("HML_COMPAT_WARMUP")) c->compat_warmup = atoi(e);   // (tests: a warm-up too short to forget the start)
    if (const char* e = getenv("HML_WIDE_LANES")) c->wide_lanes = atoi(e);   // (0: models of more than 16 states with a state a lane)
    if (const char* e = getenv("HML_WIDE_W0")) c->wide_w0 = std::max(8, atoi(e));
    if (const char* e = getenv("HML_WIDE_MAX_CHUNKS")) c->wide_max_chunks = (uint32_t)std::max(64, atoi(e));
    if (const char* e = getenv("HML_WIDE_L")) {   // (tests: chunks of that many blocks, a power of two)
        const int L = atoi(e);
        if (L >= 1) { int sh = 0; while ((1 << sh) < L && sh < 20) ++sh; c->wide_lshift = sh; }
    }
    if (const char* e = getenv("HML_COMPAT")) c->compat = atoi(e) != 0;   // option "compat" for unmodified callers (`hammlet -compat`)
    if (const char* e = getenv("HML_TRELLIS_TUNE")) c->tre_autotune = atoi(e) != 0;
    if (const char* e = getenv("HML_FUSED_SPIN_LIMIT")) c->fused_spin_limit = (uint32_t)strtoul(e, nullptr, 10);
    if (const char* e = getenv("HML_MAX_BLOCKS")) c->cap_opt = strtoull(e, nullptr, 10);   // option "max_blocks" (tests: a tiny capacity exercises the growth everywhere)
    if (device < 64) g_live_ctx[device].fetch_add(1);
    *out = c;
    return 0;
}

// the shared construction: the last context that holds it frees it
static void trace_release(hml_ctx* c) {
    if (c->trace) {
        if (c->trace->refs.fetch_sub(1) == 1) {
            void* ptrs[] = {c->trace->d_w, c->trace->d_summary, c->trace->d_coeff, c->trace->d_ia};
            for (void* p : ptrs) if (p) hipFree(p);
            delete c->trace;
        }
        c->trace = nullptr;
    } else {
        // (a load that failed before the trace object existed)
        void* ptrs[] = {c->d_w, c->d_summary, c->d_coeff, c->d_ia};
        for (void* p : ptrs) if (p) hipFree(p);
    }
    c->d_w = nullptr; c->d_summary = nullptr; c->d_coeff = nullptr; c->d_ia = nullptr;
}
static bool trace_shared(const hml_ctx* c) { return c->trace && c->trace->refs.load() > 1; }

// the buffers hml_set_model allocates (keep_engine: growing the buffers of a running chain - the reference-compatible
// mode's mt19937 state lives on)
static void free_sweep_buffers(hml_ctx* c, bool keep_engine = false) {
    void** ptrs[] = {(void**)&c->d_em, (void**)&c->d_gsc, (void**)&c->d_rows, (void**)&c->d_entry, (void**)&c->d_exitA, (void**)&c->d_redo, (void**)&c->d_touched,
                     (void**)&c->d_fb, (void**)&c->d_smap, (void**)&c->d_cmap, (void**)&c->d_scmap, (void**)&c->d_super, (void**)&c->d_bentry2, (void**)&c->d_bentry,
                     (void**)&c->d_q, (void**)&c->d_partial, (void**)&c->d_redo2, (void**)&c->d_tre_bitmap, (void**)&c->d_tre_ckpt, (void**)&c->d_crows, (void**)&c->d_cchunk, (void**)&c->d_cdraws, (void**)&c->d_clists, (void**)&c->d_wacc, (void**)&c->d_wA};
    for (void** p : ptrs) if (*p) { hipFree(*p); *p = nullptr; }
    if (!keep_engine && c->d_mt) { hipFree(c->d_mt); c->d_mt = nullptr; }
}

static void free_all(hml_ctx* c) {
    trace_release(c);
    free_sweep_buffers(c);
    void* ptrs[] = {c->d_group_word, c->d_wave_total, c->d_stage, c->d_span_count, c->d_starts, c->d_bstat, c->d_eprobe, c->d_aprobe, c->d_coarse1,
                    c->d_diff, c->d_boundary, c->d_mdl, c->d_many};
    for (void* p : ptrs) if (p) hipFree(p);
    if (c->h_B) hipHostFree(c->h_B);
    c->h_B = nullptr;
}


void hml_destroy(hml_ctx* c) {
    if (!c) return;
    if (c->device < 64) g_live_ctx[c->device].fetch_sub(1);
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (auto& kv : c->prof) for (auto& p : kv.second.pending) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (auto e : c->ev_pool) hipEventDestroy(e);
    if (c->graph_exec) hipGraphExecDestroy(c->graph_exec);
    free_all(c);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

// ---------------------------------------------------------------------------------------- load
// 8-bit keys of the current weights, bucket window centred on the universal threshold of the noise estimate
static int build_keys(hml_ctx* c) {
    if (!c->use_keys) return 0;
    const uint64_t T = c->T;
    if (!c->d_summary) {
        // one byte per 16-position group, padded with zeros to whole spans
        const uint64_t n_spans = (T + HML_SPAN - 1) / HML_SPAN;
        HIPCHK(hipMalloc(&c->d_summary, n_spans * (HML_SPAN / 16) + 16));
        HIPCHK(hipMemsetAsync(c->d_summary, 0, n_spans * (HML_SPAN / 16) + 16, c->stream));
    }
    const float thr0 = (float)(std::sqrt(2 * std::log((double)std::max<uint64_t>(T, 2))) * c->sigma * c->key_scale);
    uint32_t u; memcpy(&u, &thr0, 4);
    c->key_base = (std::isfinite(thr0) && thr0 > 0 ? (int32_t)(u >> 20) : (int32_t)(0x3f800000u >> 20)) - 128;
    hipLaunchKernelGGL(hml_k_build_summary, dim3(grid_for((T + 15) / 16, 256, 1, 65536)), dim3(256), 0, c->stream, c->d_w, T,
                       c->key_base, c->d_summary);
    KLAUNCH_CHECK();
    return 0;
}

HML_KERNEL void hml_k_max_inplace(float* __restrict__ a, const float* __restrict__ b, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const float x = a[i], y = b[i]; a[i] = (x < y) ? y : x; }   // std::max(x, y)
}

// the per-chain block-structure buffers (every context has its own; the construction above them may be shared)
static int alloc_block_buffers(hml_ctx* c) {
    const uint64_t T = c->T;
    c->n_spans = (uint32_t)((T + HML_SPAN - 1) / HML_SPAN);
    HIPCHK(hipMalloc(&c->d_stage, (uint64_t)c->n_spans * HML_SPAN * sizeof(uint16_t)));
    HIPCHK(hipMalloc(&c->d_span_count, c->n_spans * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&c->d_starts, (c->cap + 1) * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&c->d_coarse1, ((c->n_spans + HML_GROUP_SPANS - 1) / HML_GROUP_SPANS + 1u) * sizeof(uint32_t)));
    {
        const uint64_t n_tiles = (T + HML_FUSED_SUB_POSITIONS - 1) / HML_FUSED_SUB_POSITIONS;   // (tiles of one batch: the most there can be)
        HIPCHK(hipMalloc(&c->d_group_word, (n_tiles + 1) * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(c->d_group_word, 0, (n_tiles + 1) * sizeof(unsigned long long), c->stream));
        HIPCHK(hipMalloc(&c->d_wave_total, (n_tiles + 1) * (HML_FUSED_WAVES + 2) * sizeof(uint32_t)));   // (+ tile_before, tile_prev)
        if (getenv("HML_FUSED_DEBUG")) { HIPCHK(hipMalloc(&c->d_dbg, 4096 * 8 * 8)); HIPCHK(hipMemset(c->d_dbg, 0, 4096 * 8 * 8)); }
    }
    HIPCHK(hipMalloc(&c->d_bstat, c->cap * (uint64_t)c->D * sizeof(float2)));   // (D > 1: cap = T, the planes lie T apart)
    // the capacity the enumeration kernels respect (they run before there is a model: auto prior, explicit thresholds)
    const uint32_t cap32 = (uint32_t)c->cap;
    HIPCHK(hipMemcpyAsync(&c->d_mdl->cap, &cap32, sizeof cap32, hipMemcpyHostToDevice, c->stream));
    return 0;
}

// the block capacity of a context whose observations are being loaded (`attached`: to another context's construction)
static void choose_capacity(hml_ctx* c, bool attached) {
    const uint64_t T = c->T;
    uint64_t cap = T;
    if (c->D == 1) {
        if (c->cap_opt) cap = c->cap_opt;
        // a further chain on a GPU (hml_attach_observations): room for the strongly compressed regime (up to 2^22 blocks a sweep,
        // or a sixteenth of the positions) instead of the worst case - it grows if a sweep needs more
        else if (attached) cap = std::max<uint64_t>(1ull << 20, T / 16);
    }
    c->cap = std::min<uint64_t>(std::max<uint64_t>(cap, 64), T);
}

// d_x[d], h_x[d]: the observations of dimension d (T values each), on the device and on the host
static int build_from_device_x(hml_ctx* c, const float* const* d_x, const float* const* h_x) {
    const uint64_t T = c->T;
    const int D = c->D;
    {   // what an earlier load that failed half-way may have left behind
        trace_release(c);
        void** stale[] = {(void**)&c->d_stage, (void**)&c->d_span_count, (void**)&c->d_starts, (void**)&c->d_coarse1, (void**)&c->d_group_word, (void**)&c->d_wave_total, (void**)&c->d_bstat};
        for (void** q : stale) if (*q) { hipFree(*q); *q = nullptr; }
    }
    // noise estimate (src/main.cpp:303-311): f64 accumulation, in index order, of the finest-level
    // maxlet coefficients c[t] = max over the dimensions of sqrt2half * |x[t-1] - x[t]| at odd t
    // (wavelet.hpp:139-158, level 1)
    {
        const float sqrt2 = (float)std::sqrt(2.0);
        const float sqrt2half = (float)(sqrt2 / 2.0);
        double acc = 0; uint64_t cnt = 0;
        for (uint64_t i = 1; i < T; i += 2) {
            float mx = 0.0f;
            for (int d = 0; d < D; ++d) {
                const float df = std::abs(h_x[d][i - 1] - h_x[d][i]);
                const float cf = sqrt2half * df;
                mx = (mx < cf) ? cf : mx;
            }
            acc += mx;
            cnt++;
        }
        acc /= cnt;
        acc /= 0.797884560802865355879892119868763736951717262329869315331;
        c->sigma = acc;
    }
    HIPCHK(hipMalloc(&c->d_w, T * sizeof(float)));
    HIPCHK(hipMalloc(&c->d_coeff, T * sizeof(float)));
    HIPCHK(hipMalloc(&c->d_ia, (T + 1) * (uint64_t)D * sizeof(float2)));
    // K1: levels 10 at a time, one dimension after the other; the coefficient is the maximum over the dimensions
    {
        float h_norm[64];
        const float sqrt2 = (float)std::sqrt(2.0);
        const float sqrt2half = (float)(sqrt2 / 2.0);
        h_norm[0] = 1.0f;
        float nrm = sqrt2half;
        for (int l = 1; l < 64; ++l) { h_norm[l] = nrm; nrm *= sqrt2half; }
        DevScratch s_norm, s_a, s_b, s_other;   // freed on every path out of this scope
        HIPCHK(hipMalloc(&s_norm.p, sizeof h_norm));
        float* d_norm = s_norm.as<float>();
        HIPCHK(hipMemcpyAsync(d_norm, h_norm, sizeof h_norm, hipMemcpyHostToDevice, c->stream));
        const uint64_t n1 = T >> HML_MAXLET_LOG_TILE;
        HIPCHK(hipMalloc(&s_a.p, std::max<uint64_t>(n1, 1) * sizeof(float)));
        HIPCHK(hipMalloc(&s_b.p, std::max<uint64_t>(n1 >> HML_MAXLET_LOG_TILE, 1) * sizeof(float)));
        if (D > 1) HIPCHK(hipMalloc(&s_other.p, T * sizeof(float)));
        float *bufA = s_a.as<float>(), *bufB = s_b.as<float>(), *d_other = s_other.as<float>();
        for (int d = 0; d < D; ++d) {
            float* coeff_out = d == 0 ? c->d_coeff : d_other;
            uint64_t n = T;
            int base = 0;
            const float* in = d_x[d];
            float* outb = bufA;
            while (true) {
                const uint64_t tiles = (n + HML_MAXLET_TILE - 1) / HML_MAXLET_TILE;
                hipLaunchKernelGGL(hml_k_maxlet, dim3((unsigned)tiles), dim3(256), 0, c->stream, in, n, base, coeff_out, T, outb, d_norm);
                KLAUNCH_CHECK();
                base += HML_MAXLET_LOG_TILE;
                if (base >= 40 || (1ull << base) >= T) break;   // no discontinuity position left below T
                n = T >> base;                                   // complete elements of the next level (>= 1)
                in = outb;
                outb = (outb == bufA) ? bufB : bufA;
            }
            if (d > 0) {
                hipLaunchKernelGGL(hml_k_max_inplace, dim3(grid_for(T, 256, 1, 65536)), dim3(256), 0, c->stream, c->d_coeff, d_other, T);
                KLAUNCH_CHECK();
            }
        }
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    hipLaunchKernelGGL(hml_k_weights, dim3(grid_for(T, 256, 1, 65536)), dim3(256), 0, c->stream, c->d_coeff, c->d_w, T, 1.0f);
    KLAUNCH_CHECK();
    if (int r = build_keys(c)) return r;
    {
        const uint64_t cells = (T + 1 + HML_CELLSIZE - 1) / HML_CELLSIZE;
        for (int d = 0; d < D; ++d) {
            hipLaunchKernelGGL(hml_k_integral, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, c->stream, d_x[d], c->d_ia + (uint64_t)d * (T + 1), T);
            KLAUNCH_CHECK();
        }
    }
    choose_capacity(c, false);
    if (int r = alloc_block_buffers(c)) return r;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->trace = new hml_trace();
    c->trace->d_w = c->d_w; c->trace->d_summary = c->d_summary; c->trace->d_coeff = c->d_coeff; c->trace->d_ia = c->d_ia;
    c->loaded = true;
    return 0;
}

// "-s C P D" (reference src/main.cpp:114-137, src/Mapping.hpp:53-137): D data dimensions whose values follow each other
// in the observation stream, P emission parameters shared by the K = P^D states.  Before the observations are loaded.
int hml_set_dimensions(hml_ctx* c, int D, int P) {
    if (!c) return set_err(HML_ERR_ARG, "null context");
    if (c->loaded) return set_err(HML_ERR_ARG, "dimensions must be set before the observations are loaded");
    if (D <= 0) return set_err(HML_ERR_MODEL, "Number of data dimensions must be positive!");
    if (P < 0) return set_err(HML_ERR_MODEL, "Number of parameters must be positive!");
    if (D > HML_MAX_D) return set_err(HML_ERR_ARG, "at most 4 data dimensions are supported");
    if (P > 0) {   // P = 0: taken from hml_set_model's number of states (K = P^D)
        long k = 1;
        for (int d = 0; d < D; ++d) { k *= P; if (k > HML_CAP_K) return set_err(HML_ERR_ARG, "number of states must be in [2,64]"); }
        if (k <= 1) return set_err(HML_ERR_MODEL, "Requested parameters would yield an HMM with less than 2 states!");
    }
    c->D = D; c->P = P;
    return 0;
}

// x: T * D values, the D dimensions of a position one after the other (T values when D = 1)
int hml_load_observations(hml_ctx* c, const float* x, uint64_t n_values) {
    if (!c || !x) return set_err(HML_ERR_ARG, "null argument");
    if (n_values == 0) return set_err(HML_ERR_ARG, "Input vector for breakpoint weights is empty!");
    const uint64_t D = (uint64_t)c->D;
    if (n_values % D != 0) return set_err(HML_ERR_MODEL, "Input stream did not contain enough values to fill all dimensions at last position!");
    const uint64_t T = n_values / D;
    if (T >= 0xffffffffull) return set_err(HML_ERR_ARG, "at most 2^32-2 positions are supported");
    if (c->loaded) return set_err(HML_ERR_ARG, "observations already loaded");
    if (int r = ctx_bind(c)) return r;
    c->T = T;
    DevScratch s_x;
    HIPCHK(hipMalloc(&s_x.p, n_values * sizeof(float)));
    float* d_x = s_x.as<float>();
    std::vector<float> planes;             // dimension-major copy when D > 1
    const float* h_dim[HML_MAX_D];
    const float* d_dim[HML_MAX_D];
    if (D == 1) { h_dim[0] = x; }
    else {
        planes.resize(n_values);
        for (uint64_t t = 0; t < T; ++t) for (uint64_t d = 0; d < D; ++d) planes[d * T + t] = x[t * D + d];
        for (uint64_t d = 0; d < D; ++d) h_dim[d] = planes.data() + d * T;
    }
    HIPCHK(hipMemcpyAsync(d_x, D == 1 ? x : planes.data(), n_values * sizeof(float), hipMemcpyHostToDevice, c->stream));
    for (uint64_t d = 0; d < D; ++d) d_dim[d] = d_x + d * T;
    return build_from_device_x(c, d_dim, h_dim);
}

int hml_load_observations_device(hml_ctx* c, const void* x_dev, uint64_t T) {
    if (!c || !x_dev) return set_err(HML_ERR_ARG, "null argument");
    if (c->D != 1) return set_err(HML_ERR_ARG, "device input is univariate");
    if (T == 0) return set_err(HML_ERR_ARG, "Input vector for breakpoint weights is empty!");
    if (T >= 0xffffffffull) return set_err(HML_ERR_ARG, "at most 2^32-2 positions are supported");
    if (c->loaded) return set_err(HML_ERR_ARG, "observations already loaded");
    if (int r = ctx_bind(c)) return r;
    c->T = T;
    std::vector<float> h(T);
    HIPCHK(hipMemcpy(h.data(), x_dev, T * sizeof(float), hipMemcpyDeviceToHost));
    const float* h_dim[1] = {h.data()};
    const float* d_dim[1] = {(const float*)x_dev};
    return build_from_device_x(c, d_dim, h_dim);
}

// A second chain over the SAME observations on the same device: shares the source's read-only construction (weights,
// summary, coefficients, integral arrays - reference counted) instead of building a private copy; block structure and sweep
// buffers stay per chain.  The reference holds one chain per process over its one trace (src/main.cpp:108,338-343).
int hml_attach_observations(hml_ctx* c, hml_ctx* src) {
    if (!c || !src) return set_err(HML_ERR_ARG, "null argument");
    if (c == src) return set_err(HML_ERR_ARG, "a context cannot be attached to itself");
    if (!src->loaded || !src->trace) return set_err(HML_ERR_ARG, "the source context has no observations loaded");
    if (c->loaded) return set_err(HML_ERR_ARG, "observations already loaded");
    if (c->device != src->device) return set_err(HML_ERR_ARG, "contexts that share observations must live on one device");
    if (c->use_keys != src->use_keys) return set_err(HML_ERR_ARG, "option weight_keys differs from the source context's");
    if (int r = ctx_bind(c)) return r;
    HIPCHK(hipStreamSynchronize(src->stream));   // (a weight multiplier still on its way)
    trace_release(c);
    {
        void** stale[] = {(void**)&c->d_stage, (void**)&c->d_span_count, (void**)&c->d_starts, (void**)&c->d_coarse1, (void**)&c->d_group_word, (void**)&c->d_wave_total, (void**)&c->d_bstat};
        for (void** q : stale) if (*q) { hipFree(*q); *q = nullptr; }
    }
    c->T = src->T; c->D = src->D;
    if (c->P == 0 || c->D > 1) c->P = src->P;
    c->sigma = src->sigma; c->key_base = src->key_base; c->key_scale = src->key_scale;
    c->trace = src->trace;
    c->trace->refs.fetch_add(1);
    c->d_w = src->d_w; c->d_summary = src->d_summary; c->d_coeff = src->d_coeff; c->d_ia = src->d_ia;
    choose_capacity(c, true);
    if (int r = alloc_block_buffers(c)) { trace_release(c); return r; }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->loaded = true;
    return 0;
}

int hml_noise_sigma(hml_ctx* c, double* sigma) {
    if (!c || !c->loaded) return set_err(HML_ERR_ARG, "no observations loaded");
    *sigma = c->sigma;
    return 0;
}

int hml_scale_weights(hml_ctx* c, float mult) {
    if (!c || !c->loaded) return set_err(HML_ERR_ARG, "no observations loaded");
    if (trace_shared(c)) return set_err(HML_ERR_ARG, "the weights are shared with attached contexts: scale them before hml_attach_observations");
    if (int r = ctx_bind(c)) return r;
    hipLaunchKernelGGL(hml_k_scale, dim3(grid_for(c->T, 256, 1, 65536)), dim3(256), 0, c->stream, c->d_w, c->T, mult);
    KLAUNCH_CHECK();
    c->key_scale *= std::fabs((double)mult) > 0 ? std::fabs((double)mult) : 1.0;
    if (int r = build_keys(c)) return r;
    c->blocks_valid = false;
    return 0;
}

int hml_set_weights(hml_ctx* c, const float* w, uint64_t T) {
    if (!c || !c->loaded || !w) return set_err(HML_ERR_ARG, "no observations loaded");
    if (T != c->T) return set_err(HML_ERR_MODEL, "Block structure and statistics have different number of data points!");
    if (trace_shared(c)) return set_err(HML_ERR_ARG, "the weights are shared with attached contexts: set them before hml_attach_observations");
    if (int r = ctx_bind(c)) return r;
    HIPCHK(hipMemcpyAsync(c->d_w, w, T * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // the key window stays centred on the noise estimate times the multipliers applied through hml_scale_weights: the
    // window only decides how many groups the summary scan opens, never which positions start a block
    if (int r = build_keys(c)) return r;
    c->blocks_valid = false;
    c->hint_stale = true;
    return 0;
}

}  // extern "C"

extern "C" {
}  // extern "C"
static int grow_capacity(hml_ctx* c, uint64_t needed);
// an enumeration at an explicit threshold (with block statistics), waited for; a context with a reduced block capacity
// grows until the blocks fit (hml_state.h)
static int enumerate_blocks_sync(hml_ctx* c, float threshold) {
    if (int r = settle_if_limited(c)) return r;   // (sweeps still under way come first)
    for (int round = 0; round < 64; ++round) {
        if (int r = launch_compact(c, true, threshold)) return r;
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!chain_halted(c)) return 0;
        if (int r = grow_capacity(c, c->h_B[2])) return r;
    }
    return set_err(HML_ERR_HIP, "internal error: the block capacity did not settle");
}
extern "C" {

int hml_create_blocks(hml_ctx* c, float threshold) {
    if (!c || !c->loaded) return set_err(HML_ERR_ARG, "no observations loaded");
    if (int r = ctx_bind(c)) return r;
    if (int r = enumerate_blocks_sync(c, threshold)) return r;
    refresh_hint(c);
    c->blocks_valid = false;   // an explicit threshold is not the model's threshold
    return 0;
}

int hml_autoprior(hml_ctx* c, float s2, float p, float out4[4]) {
    if (!c || !c->loaded) return set_err(HML_ERR_ARG, "no observations loaded");
    if (int r = ctx_bind(c)) return r;
    // y.createBlocks( sqrt(2*log((double)T)) * noiseStdev )   (AutoPriors.hpp:95-96)
    const float thr0 = (float)(std::sqrt(2 * std::log((double)c->T)) * c->sigma);
    if (int r = enumerate_blocks_sync(c, thr0)) return r;
    refresh_hint(c);
    const uint32_t B = *c->h_B;
    std::vector<uint32_t> st(B + 1);
    const int D = c->D;
    std::vector<float2> bs((size_t)B * D);
    HIPCHK(hipMemcpy(st.data(), c->d_starts, (B + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int d = 0; d < D; ++d)
        HIPCHK(hipMemcpy(bs.data() + (size_t)d * B, c->d_bstat + (uint64_t)d * c->cap, B * sizeof(float2), hipMemcpyDeviceToHost));   // (D > 1: cap = T)
    // block means in block order, every dimension of a block in turn (AutoPriors.hpp:100-104), float accumulation
    // (SufficientStatistics<Normal>::addObs); N = nrBlocks * nrDim (AutoPriors.hpp:105)
    float muSum = 0, muSq = 0;
    for (uint32_t b = 0; b < B; ++b) {
        for (int d = 0; d < D; ++d) {
            const float m = bs[(size_t)d * B + b].x / (float)(st[b + 1] - st[b]);
            muSum += m;
            muSq += m * m;
        }
    }
    const double n = (double)((uint64_t)B * (uint64_t)D);
    const double blocksMean = (double)(float)(muSum / n);
    const double avg = (double)(float)(muSum / n);
    const double blocksVar = (double)(float)(muSq / n - (avg * avg));
    const float dataMean = (float)blocksMean, dataVar = (float)blocksVar;
    if (p < 0 || p > 1) return set_err(HML_ERR_MODEL, "Parameter p for automatic priors is a probability and must be in [0,1]!");
    if (s2 <= 0) return set_err(HML_ERR_MODEL, "Parameter s2  for automatic priors is a variance and must be positive!");
    if (dataVar <= 0) return set_err(HML_ERR_MODEL, "Data variance provided to autoprior must be positive!");
    const float M1 = 0.3361, M2 = -0.0042, M3 = -0.0201;
    const float b = -std::log(p);
    const float alpha = 2.0;
    const float beta = s2 * ((2.0 * std::sqrt(b)) / (M1 * std::sqrt(b) + std::sqrt(2.0) * (M2 * b * std::exp(M3 * std::sqrt(b)) + 1)) + b);
    const float mu0 = dataMean;
    const float nu = beta / dataVar;
    if (beta <= 0) return set_err(HML_ERR_MODEL, "Autoprior yields non-positive beta!");
    if (nu <= 0) return set_err(HML_ERR_MODEL, "Autoprior yields non-positive nu!");
    if (!std::isfinite(beta)) return set_err(HML_ERR_MODEL, "Autoprior yields non-finite beta!");
    if (!std::isfinite(mu0)) return set_err(HML_ERR_MODEL, "Autoprior yields non-finite mu0!");
    if (!std::isfinite(nu)) return set_err(HML_ERR_MODEL, "Autoprior yields non-finite nu!");
    out4[0] = alpha; out4[1] = beta; out4[2] = mu0; out4[3] = nu;
    return 0;
}

}  // extern "C"

// entries of one array of the reference-compatible mode's lists by state (hml_compat_lists): the blocks and up to three entries of
// padding per state, rounded to whole 16-byte groups
static uint64_t compat_list_entries(const hml_ctx* c) { return (c->cap + 4u * (uint64_t)HML_CAP_K + 3u) & ~(uint64_t)3u; }

// the most chunks a sweep of the path for more than 16 states is cut into (hml_k_wl_prepare): the arrays are sized for the chunk
// length this bound gives at the context's capacity
static uint32_t wide_max_chunks_of(const hml_ctx* c) {
    if (c->wide_max_chunks) return std::min<uint32_t>(c->wide_max_chunks, HML_WL_MAX_CHUNKS);
    return c->K <= HML_WL_TWO_WAVES_KC ? (uint32_t)HML_WL_MAX_CHUNKS : (uint32_t)HML_WL_MAX_CHUNKS / 2u;
}

// the per-block sweep buffers, sized by the context's block capacity (hml_ctx.hpp; c->K set)
static int alloc_sweep_buffers(hml_ctx* c) {
    const int K = c->K;
    const uint64_t cap = c->cap;
    if (c->compat) {   // the reference-compatible mode (hml_k_compat.h): plain emission terms [b][s], a plain (B + 1) x K trellis, the states
        HIPCHK(hipMalloc(&c->d_em, cap * K * sizeof(float)));
        HIPCHK(hipMalloc(&c->d_gsc, cap * K * sizeof(float)));
        HIPCHK(hipMalloc(&c->d_crows, (cap + 1) * K * sizeof(float)));
        // chunks of the filter and of the backward draws (hml_compat_chunks), the engine's outputs of a sweep (two per block)
        HIPCHK(hipMalloc(&c->d_cchunk, (uint64_t)HML_COMPAT_MAX_CHUNKS * (2 * K * sizeof(float) + 4 * sizeof(uint32_t)) + 64));
        HIPCHK(hipMalloc(&c->d_cdraws, 2 * cap * sizeof(uint32_t)));
        // the count pass's lists by state (hml_compat_lists): statistics, sizes, per-tile counts, flags
        // (three arrays of cap + 4 K entries, each 16-byte aligned: a state's list starts at a multiple of four entries)
        HIPCHK(hipMalloc(&c->d_clists, 3 * compat_list_entries(c) * sizeof(uint32_t) + (uint64_t)HML_CAP_K * HML_CAP_K * sizeof(unsigned long long) +
                                        ((cap + HML_COMPAT_PART_TILE - 1) / HML_COMPAT_PART_TILE) * K * sizeof(uint32_t) + 2 * (HML_CAP_K + 1) * sizeof(uint32_t) + 64));
        HIPCHK(hipMalloc(&c->d_q, cap * sizeof(int16_t)));
        return 0;
    }
    if (c->wide) {   // more than 16 states (hml_k_wide.h, hml_k_wide_lanes.h): [b][s] arrays or chunk-transposed ones, the default path's count tree
        // (chunk-transposed: L K cstride floats with cstride = the chunks rounded up to 64 - at most K (B + 64 L), L the longest chunk
        // hml_k_wl_prepare can choose for this capacity)
        uint64_t Lmax = 1ull << HML_WL_MIN_LSHIFT;
        if (c->wide_lshift >= 0) Lmax = std::max<uint64_t>(Lmax, 1ull << c->wide_lshift);
        while ((cap + Lmax - 1) / Lmax > (uint64_t)wide_max_chunks_of(c)) Lmax *= 2;   // (the longest chunks hml_k_wl_prepare can choose: B <= cap)
        const uint64_t plane = (cap + 1 + 64 * Lmax) * K;
        HIPCHK(hipMalloc(&c->d_em, plane * sizeof(float)));
        HIPCHK(hipMalloc(&c->d_gsc, plane * sizeof(float)));
        HIPCHK(hipMalloc(&c->d_crows, plane * sizeof(float)));
        // (per-chunk arrays: as many chunks as the shortest chunk length makes of the capacity - at least what the lane-per-state form may ask for)
        const uint64_t Lmin = c->wide_lshift >= 0 ? (1ull << c->wide_lshift) : (1ull << HML_WL_MIN_LSHIFT);
        c->wide_chunks_cap = std::max<uint64_t>(HML_COMPAT_MAX_CHUNKS, std::min<uint64_t>(HML_WL_MAX_CHUNKS, (cap + Lmin - 1) / Lmin + 64));
        const uint64_t chunk_bytes = c->wide_chunks_cap * (2 * K * sizeof(float) + 4 * sizeof(uint32_t));
        HIPCHK(hipMalloc(&c->d_cchunk, chunk_bytes + HML_WL_TOT_WORDS * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync((char*)c->d_cchunk + chunk_bytes, 0, HML_WL_TOT_WORDS * sizeof(unsigned long long), c->stream));   // (counters and bit maps of wrong chunks)
        HIPCHK(hipMalloc(&c->d_wA, ((uint64_t)HML_WL_PITCH * HML_WL_PITCH + (uint64_t)HML_CAP_K * HML_WL_GTAB) * sizeof(float)));   // (... and the table of rescale factors behind it)
        HIPCHK(hipMalloc(&c->d_cdraws, 2 * (cap + 1) * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&c->d_q, cap * sizeof(int16_t)));
        const uint64_t n_partial = (uint64_t)HML_REDUCE_GROUPS * K * 2;
        HIPCHK(hipMalloc(&c->d_partial, n_partial * sizeof(double)));
        HIPCHK(hipMemsetAsync(c->d_partial, 0, n_partial * sizeof(double), c->stream));
        HIPCHK(hipMalloc(&c->d_wacc, sizeof(hml_wide_acc)));
        HIPCHK(hipMemsetAsync(c->d_wacc, 0, sizeof(hml_wide_acc), c->stream));
        return 0;
    }
    const int minL = std::min(std::min(std::min(c->fwdL, c->fwdL_dense), c->fwdL_many), c->fwdL_mid);
    const uint64_t maxChunks = (cap + minL - 1) / minL + 1;   // per-chunk arrays serve either geometry
    uint64_t plane = 0;   // floats in one chunk-transposed [L][K][cstride] array
    auto layout = [&](int L, hml_layout& lay) {
        int sh = 0; while ((1 << sh) < L) ++sh;
        lay.lshift = (uint32_t)sh;
        lay.cstride = (uint32_t)(((cap + L - 1) / L + 1 + 63) / 64 * 64);
        plane = std::max(plane, (uint64_t)L * K * lay.cstride);
    };
    layout(c->fwdL, c->lay);
    layout(c->fwdL_dense, c->lay_dense);
    layout(c->fwdL_mid, c->lay_mid);
    layout(c->fwdL_many, c->lay_many);
    HIPCHK(hipMalloc(&c->d_em, plane * sizeof(float)));
    HIPCHK(hipMalloc(&c->d_gsc, plane * sizeof(float)));
    HIPCHK(hipMalloc(&c->d_rows, plane * sizeof(float)));
    HIPCHK(hipMalloc(&c->d_entry, maxChunks * K * sizeof(float)));
    HIPCHK(hipMalloc(&c->d_exitA, maxChunks * K * sizeof(float)));
    HIPCHK(hipMalloc(&c->d_fb, maxChunks * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&c->d_smap, (cap + 2) * sizeof(unsigned long long)));
    // per-chunk arrays serve the backward chunks of 64 rows and the fused trellis path's forward chunks of 16 or 32
    const uint64_t bchunks = (cap + 15) / 16 + 1;
    HIPCHK(hipMalloc(&c->d_cmap, bchunks * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&c->d_bentry, bchunks));
    HIPCHK(hipMalloc(&c->d_scmap, bchunks * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&c->d_super, (bchunks / 64 + 2) * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&c->d_bentry2, bchunks / 64 + 2));
    HIPCHK(hipMalloc(&c->d_redo, bchunks * sizeof(uint32_t)));
    HIPCHK(hipMemsetAsync(c->d_redo, 0, bchunks * sizeof(uint32_t), c->stream));
    HIPCHK(hipMalloc(&c->d_touched, bchunks * sizeof(uint32_t)));
    HIPCHK(hipMemsetAsync(c->d_touched, 0, bchunks * sizeof(uint32_t), c->stream));
    HIPCHK(hipMalloc(&c->d_redo2, bchunks * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&c->d_tre_bitmap, (bchunks / 32 + 2) * sizeof(uint32_t)));
    // (checkpoints of the fused trellis path: (L / 64 - 1) x ceil(B / L) <= B / 64 + 16 vectors of K + 1 words)
    HIPCHK(hipMalloc(&c->d_tre_ckpt, (cap / HML_TRE_CKPT_ROWS + 64) * (uint64_t)(K + 1) * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&c->d_q, cap * sizeof(int16_t)));
    // (the count pass's group partials [2 K][1024] and, behind them, the first level of their tree [16][2 K]: hml_k_params.h)
    const uint64_t n_partial = ((uint64_t)HML_REDUCE_GROUPS + HML_PARAMS_TREE_WGS) * K * 2;
    HIPCHK(hipMalloc(&c->d_partial, n_partial * sizeof(double)));
    HIPCHK(hipMemsetAsync(c->d_partial, 0, n_partial * sizeof(double), c->stream));
    return 0;
}

// More room for blocks: the per-block buffers are released and allocated again for at least `needed` blocks (half as many
// again, at least twice the old capacity, at most T).  Nothing in them outlives a sweep except a static block structure, which
// the next sweep enumerates again.  The stream is idle.
static int grow_capacity(hml_ctx* c, uint64_t needed) {
    const uint64_t cap = std::min<uint64_t>(c->T, std::max<uint64_t>(needed + needed / 2 + 1024, 2 * c->cap));
    if (c->graph_exec) { hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    void** blocks[] = {(void**)&c->d_starts, (void**)&c->d_bstat};
    for (void** q : blocks) if (*q) { hipFree(*q); *q = nullptr; }
    free_sweep_buffers(c, /*keep_engine*/ true);
    if (c->d_eprobe) { hipFree(c->d_eprobe); c->d_eprobe = nullptr; }
    if (c->d_aprobe) { hipFree(c->d_aprobe); c->d_aprobe = nullptr; }
    c->cap = cap;
    HIPCHK(hipMalloc(&c->d_starts, (cap + 1) * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&c->d_bstat, cap * (uint64_t)c->D * sizeof(float2)));
    if (c->model_set) {
        if (int r = alloc_sweep_buffers(c)) return r;
        if (c->probes) {
            HIPCHK(hipMalloc(&c->d_eprobe, c->cap * c->K * sizeof(float)));
            HIPCHK(hipMalloc(&c->d_aprobe, (c->cap + 1) * c->K * sizeof(float)));
        }
    }
    const uint32_t words[2] = {(uint32_t)cap, 0u};   // hml_model: cap, halted (adjacent)
    static_assert(offsetof(hml_model, halted) == offsetof(hml_model, cap) + sizeof(uint32_t), "cap and halted are written together");
    HIPCHK(hipMemcpyAsync(&c->d_mdl->cap, words, sizeof words, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->h_B[2] = 0u;
    c->blocks_valid = false;
    c->hint_stale = false;      // (h_B[0] holds the count the halted enumeration found)
    c->grown++;
    return 0;
}

static int sweep_dispatch(hml_ctx* c, char method, bool record);

// The stream is idle and no sweep was left behind: a chain whose enumeration found more blocks than its buffers hold (it halts
// on the device, hml_state.h) gets larger buffers, and the sweeps it skipped - everything enqueued since the model's sweep
// counter stopped - run again, in order, with their recording flags.  Same results as with room from the start: a halted sweep
// changes nothing but scratch.
int hml_settle(hml_ctx* c) {
    for (int round = 0; round < 64; ++round) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!chain_halted(c)) { c->sweep_log.clear(); c->log_base = c->requested; return 0; }
        const uint64_t needed = c->h_B[2];
        if (c->cap >= c->T) return set_err(HML_ERR_HIP, "internal error: a chain with full block capacity halted");
        std::vector<uint8_t> pending;
        if (c->model_set) {
            hml_model m;
            if (int r = fetch_model(c, &m)) return r;
            // (the log starts at log_base sweeps; what the counter says beyond that ran, the rest did not)
            const uint64_t ran = m.sweeps >= c->log_base ? m.sweeps - c->log_base : 0;
            if (ran < c->sweep_log.size()) pending.assign(c->sweep_log.begin() + ran, c->sweep_log.end());
            c->log_base = m.sweeps;
        }
        if (int r = grow_capacity(c, needed)) return r;
        // the log now holds exactly the sweeps that have to run (again): the model's counter tells how far they got
        c->sweep_log = pending;
        bool halted_again = false;
        for (size_t i = 0; i < pending.size() && !halted_again; ++i) {
            const char method = (pending[i] & 1) ? HML_METHOD_MIXTURE : HML_METHOD_FB;
            const bool record = (pending[i] & 2) != 0;
            if (int r = sweep_dispatch(c, method, record)) return r;
            if (record && c->cb) {   // a recorded sweep the caller wants to see: it happens now (or, halted again, in the next round)
                HIPCHK(hipStreamSynchronize(c->stream));
                halted_again = chain_halted(c);
                if (!halted_again) {
                    if (int r = check_device_error(c)) return r;
                    // (hml.h: the callback is told the sweep's index in its hml_iterate call, not its place in this list)
                    const unsigned long long ordinal = c->log_base + i;
                    c->cb(c, ordinal >= c->call_base ? (uint64_t)(ordinal - c->call_base) : 0u, c->cb_user);
                }
            }
        }
    }
    return set_err(HML_ERR_HIP, "internal error: the block capacity did not settle");
}

extern "C" {

// ---------------------------------------------------------------------------------------- model
int hml_set_model(hml_ctx* c, int K, const float nig4[4], float a_off, float a_diag, float pi_alpha, int self_trans) {
    if (!c || !c->loaded) return set_err(HML_ERR_ARG, "no observations loaded");
    if (K < 2) return set_err(HML_ERR_MODEL, "Requested parameters would yield an HMM with less than 2 states!");
    if (K > HML_CAP_K) return set_err(HML_ERR_ARG, "number of states must be in [2,64]");
    if (c->model_set) return set_err(HML_ERR_ARG, "model already set");
    if (c->D > 1 && c->P == 0) {   // the number of parameters follows from K = P^D
        for (int pp = 2; pp <= K; ++pp) { long k = 1; for (int d = 0; d < c->D; ++d) k *= pp; if (k == K) { c->P = pp; break; } if (k > K) break; }
        if (c->P == 0) return set_err(HML_ERR_ARG, "number of states must be (number of parameters)^(data dimensions)");
    }
    if (c->P > 0) {
        long k = 1;
        for (int d = 0; d < c->D; ++d) k *= c->P;
        if (k != K) return set_err(HML_ERR_ARG, "number of states must be (number of parameters)^(data dimensions)");
    }
    if (!(nig4[0] > 0)) return set_err(HML_ERR_MODEL, "Alpha (" + std::to_string(nig4[0]) + ") must be positive!");
    if (!(nig4[1] > 0)) return set_err(HML_ERR_MODEL, "Beta (" + std::to_string(nig4[1]) + ") must be positive!");
    if (!(nig4[3] > 0)) return set_err(HML_ERR_MODEL, "Nu (" + std::to_string(nig4[3]) + ")must be positive!");
    const hml_ktab* kt = nullptr;   // (before anything is allocated: a development build knows one K only; the reference-compatible mode takes K at run time)
    // more than 16 states: the kernels that take the number of states at run time (hml_k_wide.h; HML_WIDE=1: from 2 states - tests)
    c->wide = !c->compat && (K > HML_MAX_K || (getenv("HML_WIDE") && atoi(getenv("HML_WIDE")) != 0));
    if (!c->compat && !c->wide) { kt = ktab(K); if (!kt) return set_err(HML_ERR_ARG, "number of states must be in [2,16]"); }
    if (int r = ctx_bind(c)) return r;
    free_sweep_buffers(c);   // (what an earlier call that failed half-way left behind)
    c->K = K;
    // the parameter kernel's tree over 16 workgroups (hml_k_params.h) from 8 states: it reads 2 K x 8 KB of group partials -
    // config 4's sweep (10 states) 0.0938 -> 0.0869 ms, config 3's (5 states) unchanged either way
    if (!getenv("HML_PARAMS_SPREAD")) c->params_spread = K >= 8;
    const uint64_t T = c->T;
    if (int r = alloc_sweep_buffers(c)) return r;

    hml_model m;
    memset(&m, 0, sizeof m);
    m.K = K; m.self_trans = self_trans ? 1 : 0; m.dynamic = 1; m.T = (uint32_t)T;
    m.D = c->D; m.P = c->P > 0 ? c->P : K; m.stat_stride = T;
    m.cap = (uint32_t)c->cap;
    for (int st = 0; st < K; ++st) {   // reversed P-ary digits (Mapping.hpp:92-103)
        int nn = st;
        for (int d = 0; d < HML_MAX_D; ++d) { m.map[st][d] = (uint8_t)(d < c->D ? nn % m.P : 0); nn /= m.P; }
    }
    for (int i = 0; i < 4; ++i) m.nig_prior[i] = nig4[i];
    m.a_off = a_off; m.a_diag = a_diag; m.pi_alpha = pi_alpha;
    m.key = hml_make_key(c->seed, c->chain);
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < 4; ++i) m.nig_post[k][i] = nig4[i];
        m.dirPi[k] = pi_alpha;
        m.pi[k] = 1.0f / K;
        for (int j = 0; j < K; ++j) { m.dirA[k * K + j] = (k == j) ? a_diag : a_off; m.A[k * K + j] = 1.0f / K; }
    }
    m.max_state_recorded = -1;
    m.tre_fused = (c->tre_fused && c->D == 1) ? 1u : 0u;
    m.tre_hi_shift = 17u; m.tre_lo_shift = 20u;
    if (const char* e = getenv("HML_TRE_REFIT_SHIFTS")) { unsigned a = 17, b = 20; if (sscanf(e, "%u,%u", &a, &b) == 2 && a < 32 && b < 32) { m.tre_hi_shift = a; m.tre_lo_shift = b; } }
    m.fwd_W0 = (uint32_t)c->fwdW;
    m.fwd_W = m.fwd_W_burnin = (uint32_t)std::max(c->fwdW, c->fwdW_init);
    if (c->compat || c->wide) { m.fwd_W = 64u; m.fwd_W0 = c->wide ? (uint32_t)c->wide_w0 : 32u; }
    m.wl_bwd_W = 64u; m.wl_bwd_quiet = 0u; m.wl_retry = 0u; m.wl_W_need = 0u; m.wl_need_age = 0u;   // the chunked lane-per-state kernels' own policy (hml_chunk_warmup_adapt)
    m.fwd_burnin_sweeps = c->fwd_burnin_sweeps;
    m.fwd_quiet_need = c->fwd_quiet_need;
    m.n_spans = c->n_spans;
    // keep the block count of an earlier enumeration (autoprior) out of the model: B = 0
    HIPCHK(hipMemcpyAsync(c->d_mdl, &m, sizeof m, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->dynamic = true;
    c->hint_stale = true;
    // Theta's constructor samples once from the prior (src/Theta.hpp:126-127)
    if (c->compat) {
        // reference-compatible mode (hml_k_compat.h): the reference's engine, seeded like `rng_t RNG(seed)` (main.cpp:107-108),
        // and a plain (B + 1) x K trellis.  Chain 0 is the reference's run at this seed; further chains of a run (`-chains N`,
        // chain ids 1, 2, ...) are the reference's runs at seed + chain id - N copies of one chain would pool to N times its
        // marginals.
        hml_mt_state h;
        hml_mt_seed(&h, c->seed + (uint64_t)c->chain);
        HIPCHK(hipMalloc(&c->d_mt, sizeof(hml_mt_state)));
        HIPCHK(hipMemcpyAsync(c->d_mt, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        hipLaunchKernelGGL(hml_k_compat_draw, dim3(1), dim3(64), 0, c->stream, c->d_mdl, (hml_mt_state*)c->d_mt, 2);
        KLAUNCH_CHECK();
        c->model_set = true;
        return 0;
    }
    if (c->wide) hipLaunchKernelGGL(hml_k_wide_params, dim3(1), dim3(1024), 0, c->stream, c->d_mdl, c->d_partial, (hml_wide_acc*)c->d_wacc, 2);
    else kt->params(c, 2);
    KLAUNCH_CHECK();
    c->model_set = true;
    return 0;
}

int hml_sample_prior(hml_ctx* c) {
    if (!c || !c->model_set) return set_err(HML_ERR_ARG, "model not set");
    if (int r = ctx_bind(c)) return r;
    if (int r = settle_if_limited(c)) return r;
    if (c->compat) {
        hipLaunchKernelGGL(hml_k_compat_draw, dim3(1), dim3(64), 0, c->stream, c->d_mdl, (hml_mt_state*)c->d_mt, 1);
    } else if (c->wide) {
        hipLaunchKernelGGL(hml_k_wide_params, dim3(1), dim3(1024), 0, c->stream, c->d_mdl, c->d_partial, (hml_wide_acc*)c->d_wacc, 1);
    } else { HML_KTAB(c->K, kt); kt->params(c, 1); }
    KLAUNCH_CHECK();
    if (c->dynamic) c->blocks_valid = false;
    c->hint_stale = true;
    return 0;
}

HML_KERNEL __launch_bounds__(64) void hml_k_set_self_trans(hml_model* mdl, int on) {
    if (threadIdx.x == 0) mdl->self_trans = on;
}

int hml_set_self_transitions(hml_ctx* c, int on) {
    if (!c || !c->model_set) return set_err(HML_ERR_ARG, "model not set");
    if (int r = ctx_bind(c)) return r;
    if (int r = settle_if_limited(c)) return r;
    hipLaunchKernelGGL(hml_k_set_self_trans, dim3(1), dim3(64), 0, c->stream, c->d_mdl, on ? 1 : 0);
    KLAUNCH_CHECK();
    if (c->graph_exec) { hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    return 0;
}

int hml_set_static_blocks(hml_ctx* c) {
    if (!c || !c->model_set) return set_err(HML_ERR_ARG, "model not set");
    if (int r = ctx_bind(c)) return r;
    if (int r = settle_if_limited(c)) return r;
    hipLaunchKernelGGL(hml_k_set_dynamic, dim3(1), dim3(64), 0, c->stream, c->d_mdl, 0, 1);
    KLAUNCH_CHECK();
    c->dynamic = false;
    if (!cap_limited(c)) {
        if (int r = launch_compact(c, false, 0.0f)) return r;
    } else {
        // a context with a reduced block capacity (attached chains, option max_blocks): the enumeration at the model's
        // threshold may find more blocks than the buffers hold - it then writes nothing and halts the chain.  The structure
        // is only valid once an enumeration has fitted, so wait for it and grow like enumerate_blocks_sync (ADVICE round 4:
        // static sweeps enqueued behind a halted enumeration ran as no-ops until the host noticed)
        bool fitted = false;
        for (int round = 0; round < 64 && !fitted; ++round) {
            if (int r = launch_compact(c, false, 0.0f)) return r;
            HIPCHK(hipStreamSynchronize(c->stream));
            fitted = !chain_halted(c);
            if (!fitted) { if (int r = grow_capacity(c, c->h_B[2])) return r; }
        }
        if (!fitted) return set_err(HML_ERR_HIP, "internal error: the block capacity did not settle");
        refresh_hint(c);
    }
    c->blocks_valid = true;
    return 0;
}

int hml_set_dynamic(hml_ctx* c, int on) {
    if (!c || !c->model_set) return set_err(HML_ERR_ARG, "model not set");
    if (int r = ctx_bind(c)) return r;
    if (int r = settle_if_limited(c)) return r;
    hipLaunchKernelGGL(hml_k_set_dynamic, dim3(1), dim3(64), 0, c->stream, c->d_mdl, on ? 1 : 0, on ? 1 : 0);
    KLAUNCH_CHECK();
    c->dynamic = on != 0;
    if (on) { c->blocks_valid = false; c->hint_stale = true; }
    return 0;
}

int hml_set_recording(hml_ctx* c, int marginals, hml_record_cb cb, void* user) {
    if (!c) return set_err(HML_ERR_ARG, "null context");
    c->rec_marginals = marginals != 0;
    c->cb = cb; c->cb_user = user;
    return 0;
}

int hml_enable_probes(hml_ctx* c, int on) {
    if (!c || !c->model_set) return set_err(HML_ERR_ARG, "model not set");
    if (int r = ctx_bind(c)) return r;
    if (int r = settle_if_limited(c)) return r;
    if (on && !c->d_eprobe) {
        HIPCHK(hipMalloc(&c->d_eprobe, c->cap * c->K * sizeof(float)));
        HIPCHK(hipMalloc(&c->d_aprobe, (c->cap + 1) * c->K * sizeof(float)));
    }
    c->probes = on != 0;
    return 0;
}

// ---------------------------------------------------------------------------------------- sweeps
}  // extern "C"

int hml_ctx_ensure_marginal_buffers(hml_ctx* c) { return ensure_marginal_buffers(c); }

// A sweep of the reference-compatible mode (hml_k_compat.h): block starts and block statistics by the default path's
// kernels, the order-dependent part in the reference's order (the number of states is a run-time value there), the
// marginals by hml_k_record.
static hml_compat_chunks chunk_views(const hml_ctx* c) {   // the arrays of d_cchunk (alloc_sweep_buffers)
    hml_compat_chunks ch;
    char* base = (char*)c->d_cchunk;
    const uint64_t n = c->wide ? c->wide_chunks_cap : (uint64_t)HML_COMPAT_MAX_CHUNKS;
    ch.entry = (float*)base; base += n * c->K * sizeof(float);
    ch.exitv = (float*)base; base += n * c->K * sizeof(float);
    ch.nfb = (uint32_t*)base; base += n * sizeof(uint32_t);
    ch.in_state = (int32_t*)base; base += n * sizeof(int32_t);
    ch.out_state = (int32_t*)base; base += n * sizeof(int32_t);
    ch.bad = (uint32_t*)base; base += n * sizeof(uint32_t);
    ch.tot = (unsigned long long*)base;
    ch.W = 0u;
    return ch;
}

// filter and backward draws in C chunks with a wavefront each (hml_k_compat.h): the chunks' entry rows against the exit rows
// before them by as many threads as there are elements, then one wavefront that runs the rare wrong chunk again; the same for
// the backward draws.  KC / PAD: the number of states at compile time, or (PAD) an upper bound of the model's.
template <int KC, class M, bool PAD>
static void launch_chunked_fb(hml_ctx* c, hipStream_t s, uint32_t C, const hml_compat_chunks& ch, float* aprobe) {
    {
        ProfScope ps(c, "forward");
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_compat_forward<KC, M, PAD>), dim3(C), dim3(64), 0, s, c->d_mdl, c->d_em, c->d_gsc, c->d_crows, aprobe, ch);
        if (C > 1u) hipLaunchKernelGGL(hml_k_compat_forward_verify, dim3(grid_for((uint64_t)C * c->K, 256, 1, 4096)), dim3(256), 0, s, c->d_mdl, ch, C);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_compat_forward_check<KC, PAD>), dim3(1), dim3(256), 0, s, c->d_mdl, c->d_em, c->d_gsc, c->d_crows, ch, C);
    }
    {
        ProfScope ps(c, "backward_maps");
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_compat_backward<KC, PAD>), dim3(C), dim3(64), 0, s, c->d_mdl, c->d_crows, c->d_cdraws, c->d_q, ch);
        if (C > 1u) hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_compat_backward_check<KC, PAD>), dim3(1), dim3(256), 0, s, c->d_mdl, c->d_crows, c->d_cdraws, c->d_q, ch, C);
    }
}

static int sweep_compat(hml_ctx* c, char method, bool record) {
    hipStream_t s = c->stream;
    if (c->dynamic || !c->blocks_valid) {
        if (int r = launch_compact(c, false, 0.0f)) return r;   // starts, block count, block statistics at the model's threshold
        if (!c->dynamic) c->blocks_valid = true;
    }
    refresh_hint(c);
    const uint32_t hint = c->B_hint ? c->B_hint : (uint32_t)std::min<uint64_t>(c->T, 1u << 20);
    const int mix = method == HML_METHOD_MIXTURE ? 1 : 0;
    hml_mt_state* const mt = (hml_mt_state*)c->d_mt;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_compat_emission<hml_glibc_exp>), dim3(grid_for(hint, 256, 64, 16384)), dim3(256), 0, s, c->d_mdl, c->d_starts, c->d_bstat, c->d_em, c->d_gsc, mix,
                       c->probes ? c->d_eprobe : nullptr);
    if (mix) hipLaunchKernelGGL(hml_k_compat_mixture, dim3(1), dim3(64), 0, s, c->d_mdl, mt, c->d_em, c->d_q);
    else {
        float* const aprobe = c->probes ? c->d_aprobe : nullptr;
        // filter and backward draws in chunks with a wavefront each, then one wavefront that checks the chunks in order
        // (hml_k_compat.h); one chunk - the sequential form - for short sweeps and when the rows are probed
        uint32_t C = 1u;
        if (!c->probes && c->compat_chunks > 1) C = std::min<uint32_t>((uint32_t)c->compat_chunks, HML_COMPAT_MAX_CHUNKS);   // (tests: any sweep in that many chunks)
        else if (!c->probes && c->compat_chunks == 0 && hint >= 8192u) C = std::min<uint32_t>((uint32_t)HML_COMPAT_MAX_CHUNKS, hint / 64u);
        hml_compat_chunks ch = chunk_views(c);
        ch.W = c->compat_warmup > 0 ? (uint32_t)c->compat_warmup : c->compat_warmup < 0 ? 0u : HML_CHUNK_W_ADAPTIVE;   // (< 0: none - tests)
        hipLaunchKernelGGL(hml_k_compat_draws, dim3(1), dim3(256), 0, s, c->d_mdl, mt, c->d_cdraws, 2u);
        // (up to 16 states: the number of states as a compile-time value - A in registers, loops unrolled; beyond: loops unrolled
        // over 32 or 64 in groups of four that stop at the model's value)
#define HML_COMPAT_FB(KC) case KC: launch_chunked_fb<KC, hml_glibc_exp, false>(c, s, C, ch, aprobe); break;
        switch (c->K <= HML_MAX_K ? c->K : 0) {
            HML_COMPAT_FB(2) HML_COMPAT_FB(3) HML_COMPAT_FB(4) HML_COMPAT_FB(5) HML_COMPAT_FB(6) HML_COMPAT_FB(7) HML_COMPAT_FB(8) HML_COMPAT_FB(9)
            HML_COMPAT_FB(10) HML_COMPAT_FB(11) HML_COMPAT_FB(12) HML_COMPAT_FB(13) HML_COMPAT_FB(14) HML_COMPAT_FB(15) HML_COMPAT_FB(16)
            default:
                if (c->K <= 32) launch_chunked_fb<32, hml_glibc_exp, true>(c, s, C, ch, aprobe);
                else launch_chunked_fb<64, hml_glibc_exp, true>(c, s, C, ch, aprobe);
        }
#undef HML_COMPAT_FB
    }
    // the count pass: by state (partition, then one wavefront over all lists) for long univariate sweeps, in block order otherwise
    hml_compat_lists pl;
    {
        const uint64_t tiles = (c->cap + HML_COMPAT_PART_TILE - 1) / HML_COMPAT_PART_TILE;
        char* base = (char*)c->d_clists;
        pl.sx = (float*)base; base += compat_list_entries(c) * sizeof(float);
        pl.sq = (float*)base; base += compat_list_entries(c) * sizeof(float);
        pl.n = (uint32_t*)base; base += compat_list_entries(c) * sizeof(uint32_t);
        pl.offdiag = (unsigned long long*)base; base += (uint64_t)HML_CAP_K * HML_CAP_K * sizeof(unsigned long long);
        pl.tile_count = (uint32_t*)base; base += tiles * c->K * sizeof(uint32_t);
        pl.state_off = (uint32_t*)base;
    }
    // (by state: a block's size and a flag share a word of the lists - traces below 2^31 positions)
    const int by_state = (c->D == 1 && c->T < (1ull << 31) && (c->compat_chunks > 1 || (c->compat_chunks == 0 && hint >= 8192u))) ? 1 : 0;
    if (by_state) {
        const unsigned tiles_now = (unsigned)(((uint64_t)hint + hint / 8 + HML_COMPAT_PART_TILE) / HML_COMPAT_PART_TILE);   // (kernels find B themselves; tiles beyond it return)
        const unsigned tg = std::min<uint64_t>(std::max(1u, tiles_now), (c->cap + HML_COMPAT_PART_TILE - 1) / HML_COMPAT_PART_TILE);
        hipLaunchKernelGGL(hml_k_compat_part_count, dim3(tg), dim3(256), 0, s, c->d_mdl, c->d_q, pl);
        hipLaunchKernelGGL(hml_k_compat_part_scan, dim3(1), dim3(64), 0, s, c->d_mdl, pl);
        hipLaunchKernelGGL(hml_k_compat_part_scatter, dim3(tg), dim3(256), 0, s, c->d_mdl, c->d_q, c->d_starts, c->d_bstat, pl);
    }
    hipLaunchKernelGGL(hml_k_compat_update, dim3(1), dim3(by_state ? 256 : 64), 0, s, c->d_mdl, mt, c->d_starts, c->d_bstat, c->d_q, mix, pl, by_state);
    if (record && c->rec_marginals) {
        if (c->pooled) return set_err(HML_ERR_ARG, "the marginals of this context are pooled (common labels, several chains): further sweeps cannot be recorded into them");
        if (int r = ensure_marginal_buffers(c)) return r;
        hipLaunchKernelGGL(hml_k_record, dim3(grid_for(hint, 256, 64, 16384)), dim3(256), 0, s, c->d_q, c->d_starts, c->d_mdl, c->d_diff, c->d_boundary);
    }
    KLAUNCH_CHECK();
    return 0;
}

// A sweep of a model with more than 16 states (hml_k_wide.h): block starts and block statistics by the default path's kernels,
// emission terms / filter / backward draws by the lane-per-state kernels of hml_k_compat.h in this path's arithmetic, the
// uniforms of the draws from their Philox addresses ahead of them, the default path's count tree, hml_k_wide_params.
static int sweep_wide(hml_ctx* c, char method, bool record) {
    hipStream_t s = c->stream;
    if (c->dynamic || !c->blocks_valid) {
        if (int r = launch_compact(c, false, 0.0f)) return r;   // starts, block count, block statistics at the model's threshold
        if (!c->dynamic) c->blocks_valid = true;
    }
    refresh_hint(c);
    const uint32_t hint = c->B_hint ? c->B_hint : (uint32_t)std::min<uint64_t>(c->T, 1u << 20);
    const int mix = method == HML_METHOD_MIXTURE ? 1 : 0;
    // filter and backward draws with a chunk a lane over chunk-transposed arrays (hml_k_wide_lanes.h) - unless the rows are probed,
    // a test asked for a number of chunks of the other form, or the sweep has no filter at all
    const bool lanes = c->wide_lanes != 0 && !c->probes && !mix && c->compat_chunks == 0;
    if (lanes) {
        hml_compat_chunks ch = chunk_views(c);
        ch.W = c->compat_warmup > 0 ? (uint32_t)c->compat_warmup : c->compat_warmup < 0 ? 0u : HML_CHUNK_W_ADAPTIVE;
        const uint64_t room = (uint64_t)hint + hint / 4 + 1024;   // (the kernels find B themselves: their loops stride over any grid)
        hipLaunchKernelGGL(hml_k_wl_prepare, dim3(1), dim3(256), 0, s, c->d_mdl, c->d_wA, c->wide_lshift, wide_max_chunks_of(c));
        hipLaunchKernelGGL(hml_k_wl_gtable, dim3(grid_for((uint64_t)c->K * HML_WL_GTAB, 256, 1, 512)), dim3(256), 0, s, c->d_mdl, c->d_wA + HML_WL_PITCH * HML_WL_PITCH);
        {
            ProfScope ps(c, "stats_emission");
            // (a wavefront: 64 chunks x HML_WL_EMIT_ROWS rows)
            hipLaunchKernelGGL(hml_k_wl_emission, dim3(grid_for(room, 64 * HML_WL_EMIT_ROWS * 4, 1, 32768)), dim3(256), 0, s, c->d_mdl, c->d_starts, c->d_bstat, c->d_em, c->d_gsc, c->d_wA + HML_WL_PITCH * HML_WL_PITCH);
        }
        const uint64_t minL = c->wide_lshift >= 0 ? (1ull << c->wide_lshift) : (1ull << HML_WL_MIN_LSHIFT);
        const int tiles = grid_for(room, (int)std::min<uint64_t>(64 * minL, 1u << 30), 1, HML_WL_MAX_CHUNKS / 64);
#define HML_WL_FB(KC)                                                                                                                                   \
        case KC: {                                                                                                                                      \
            constexpr int KS = (KC <= 32) ? 32 : 64;                                                                                                    \
            {                                                                                                                                           \
                ProfScope ps(c, "forward");                                                                                                             \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_wl_forward<KC>), dim3(tiles), dim3(64), 0, s, c->d_mdl, c->d_wA, c->d_em, c->d_gsc, c->d_crows, ch, 0); \
                hipLaunchKernelGGL(hml_k_wl_forward_verify, dim3(grid_for(room / 16 * c->K, 256, 1, 4096)), dim3(256), 0, s, c->d_mdl, ch, 0);          \
                /* many wrong chunks: the filter once more with a longer warm-up (the three launches return at once otherwise) */                      \
                hipLaunchKernelGGL(hml_k_wl_retry_decide, dim3(1), dim3(256), 0, s, c->d_mdl, ch);                                                      \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_wl_forward<KC>), dim3(tiles), dim3(64), 0, s, c->d_mdl, c->d_wA, c->d_em, c->d_gsc, c->d_crows, ch, 1); \
                hipLaunchKernelGGL(hml_k_wl_forward_verify, dim3(grid_for(room / 16 * c->K, 256, 1, 4096)), dim3(256), 0, s, c->d_mdl, ch, 1);          \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_wl_forward_check<KS>), dim3(1), dim3(256), 0, s, c->d_mdl, c->d_em, c->d_gsc, c->d_crows, ch);  \
            }                                                                                                                                           \
            {                                                                                                                                           \
                ProfScope ps(c, "backward_maps");                                                                                                       \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_wl_backward<KC>), dim3(tiles), dim3(64), 0, s, c->d_mdl, c->d_crows, c->d_q, ch);               \
                hipLaunchKernelGGL(hml_k_wl_backward_verify, dim3(grid_for(room / 16, 256, 1, 256)), dim3(256), 0, s, c->d_mdl, ch);                     \
                hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_wl_backward_check<KS>), dim3(1), dim3(256), 0, s, c->d_mdl, c->d_crows, c->d_q, ch);          \
            }                                                                                                                                           \
        } break;
        switch ((c->K + 3) / 4 * 4) {
            HML_WL_FB(4) HML_WL_FB(8) HML_WL_FB(12) HML_WL_FB(16)   // (HML_WIDE=1: this path for any model - tests)
            HML_WL_FB(20) HML_WL_FB(24) HML_WL_FB(28) HML_WL_FB(32) HML_WL_FB(36) HML_WL_FB(40) HML_WL_FB(44) HML_WL_FB(48) HML_WL_FB(52) HML_WL_FB(56)
            HML_WL_FB(60) HML_WL_FB(64)
            default: return set_err(HML_ERR_HIP, "internal error: a model of this many states on the path for more than 16");
        }
#undef HML_WL_FB
    } else {
    {
        ProfScope ps(c, "stats_emission");
        hipLaunchKernelGGL(hml_k_wide_emission, dim3(grid_for(hint, 256, 1, 4096)), dim3(256), 0, s, c->d_mdl, c->d_starts, c->d_bstat,
                           c->d_em, c->d_gsc, mix, c->probes ? c->d_eprobe : nullptr);
    }
    if (mix) hipLaunchKernelGGL(hml_k_wide_mixture, dim3(grid_for(hint, 256, 1, 16384)), dim3(256), 0, s, c->d_mdl, c->d_em, c->d_q);
    else {
        float* const aprobe = c->probes ? c->d_aprobe : nullptr;
        uint32_t C = 1u;   // (one chunk - the sequential form - for short sweeps and when the rows are probed)
        if (!c->probes && c->compat_chunks > 1) C = std::min<uint32_t>((uint32_t)c->compat_chunks, HML_COMPAT_MAX_CHUNKS);
        else if (!c->probes && c->compat_chunks == 0 && hint >= 2048u) C = std::min<uint32_t>((uint32_t)HML_COMPAT_MAX_CHUNKS, hint / 64u);
        hml_compat_chunks ch = chunk_views(c);
        ch.W = c->compat_warmup > 0 ? (uint32_t)c->compat_warmup : c->compat_warmup < 0 ? 0u : HML_CHUNK_W_ADAPTIVE;
        hipLaunchKernelGGL(hml_k_wide_uniforms, dim3(grid_for((hint + 1u) / 2u, 256, 1, 4096)), dim3(256), 0, s, c->d_mdl, c->d_cdraws);
        if (c->K <= 32) launch_chunked_fb<32, hml_dev_exp, true>(c, s, C, ch, aprobe);
        else launch_chunked_fb<64, hml_dev_exp, true>(c, s, C, ch, aprobe);
    }
    }
    {
        ProfScope ps(c, "counts");
        hipLaunchKernelGGL(hml_k_wide_counts, dim3(HML_REDUCE_GROUPS), dim3(256), 0, s, c->d_q, c->d_starts, c->d_bstat, c->d_mdl, c->d_partial, (hml_wide_acc*)c->d_wacc);
    }
    if (record && c->rec_marginals) {
        if (c->pooled) return set_err(HML_ERR_ARG, "the marginals of this context are pooled (common labels, several chains): further sweeps cannot be recorded into them");
        if (int r = ensure_marginal_buffers(c)) return r;
        hipLaunchKernelGGL(hml_k_record, dim3(grid_for(hint, 256, 64, 16384)), dim3(256), 0, s, c->d_q, c->d_starts, c->d_mdl, c->d_diff, c->d_boundary);
    }
    {
        ProfScope ps(c, "params");
        hipLaunchKernelGGL(hml_k_wide_params, dim3(1), dim3(1024), 0, s, c->d_mdl, c->d_partial, (hml_wide_acc*)c->d_wacc, 0);
    }
    KLAUNCH_CHECK();
    return 0;
}

static int sweep_dispatch(hml_ctx* c, char method, bool record) {
    if (c->compat) return sweep_compat(c, method, record);
    if (c->wide) return sweep_wide(c, method, record);
    HML_KTAB(c->K, kt);
    return kt->sweep(c, method, record);
}

extern "C" {

int hml_iterate(hml_ctx* c, char method, uint64_t iterations, uint64_t thinning) {
    if (!c || !c->model_set) return set_err(HML_ERR_ARG, "model not set");
    if (method != HML_METHOD_FB && method != HML_METHOD_MIXTURE)
        return set_err(HML_ERR_ARG, std::string("Unknown sampling type ") + method + "!");
    if (int r = ctx_bind(c)) return r;
    c->call_base = c->requested;
    if (thinning > 0 && thinning <= iterations && c->rec_marginals && c->pooled)   // (before anything is enqueued: a sweep is not abandoned half-way)
        return set_err(HML_ERR_ARG, "the marginals of this context are pooled (common labels, several chains): further sweeps cannot be recorded into them");
    for (uint64_t i = 0; i < iterations; ++i) {
        const bool record = thinning > 0 && ((i + 1) % thinning == 0);
        // a chain with a reduced block capacity that halted (hml_state.h): larger buffers, the skipped sweeps again - then on
        if (chain_halted(c)) { if (int r = hml_settle(c)) return r; }
        refresh_hint(c);
        const bool tre_path = c->tre_fused && c->D == 1 && method == HML_METHOD_FB && c->B_hint >= c->dense_min_blocks;
        if (c->use_graph && !c->compat && !record && !c->profiling && !c->probes && (c->dynamic || c->blocks_valid) &&
            !(tre_path && tre_wants_measurement(c, c->B_hint))) {
            // replay a captured sweep; capture again when the launch geometry (grid hint / mode / chunk length) changed
            const uint32_t hint = c->B_hint;
            bool unused = false;
            // (a captured sweep holds the fused block kernel or the scan + scatter pair: capture again when a second
            // context appeared on the device, a bounded wait expired or the option changed)
            const bool wants_fused = c->fused_blocks && !(c->h_B[1] && !c->fused_keep) && !shares_device(c);
            const bool stale = !c->graph_exec || c->graph_method != method || c->graph_dynamic != c->dynamic ||
                               hint > c->graph_hint || hint * 2u < c->graph_hint || c->graph_fused != wants_fused ||
                               c->graph_dense != (hint >= c->dense_min_blocks) || c->graph_mid != (hint >= c->mid_min_blocks) ||
                               (tre_path && c->graph_tre_L != tre_pick_L(c, hint, true, &unused));
            if (stale && hint) {
                if (c->graph_exec) { hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
                hipGraph_t g = nullptr;
                HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                // whatever happens below, the stream must leave capture mode (a failure in between would otherwise
                // make every later call on this context fail) and a partial graph must not stay behind
                int rr = sweep_dispatch(c, method, false);
                const hipError_t ec = hipStreamEndCapture(c->stream, &g);
                if (rr || ec != hipSuccess) {
                    if (g) hipGraphDestroy(g);
                    (void)hipGetLastError();
                    return rr ? rr : set_err(HML_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ec));
                }
                const hipError_t ei = hipGraphInstantiate(&c->graph_exec, g, nullptr, nullptr, 0);
                hipGraphDestroy(g);
                if (ei != hipSuccess) { c->graph_exec = nullptr; return set_err(HML_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei)); }
                c->graph_method = method; c->graph_dynamic = c->dynamic; c->graph_hint = c->B_hint; c->graph_fused = wants_fused;
                c->graph_dense = c->B_hint >= c->dense_min_blocks;
                c->graph_mid = c->B_hint >= c->mid_min_blocks;
            }
            if (c->graph_exec) {
                log_sweep(c, method, false);
                HIPCHK(hipGraphLaunch(c->graph_exec, c->stream));
                if (tre_path) c->tre_dense_sweeps++;
                continue;
            }
        }
        log_sweep(c, method, record);
        if (int r = sweep_dispatch(c, method, record)) return r;
        if (record && c->cb) {
            HIPCHK(hipStreamSynchronize(c->stream));
            if (chain_halted(c)) { if (int r = hml_settle(c)) return r; continue; }   // (runs the sweep again and calls back)
            if (int r = check_device_error(c)) return r;
            c->cb(c, i, c->cb_user);
        }
    }
    return 0;
}

}  // extern "C"

extern "C" int hml_iterate_many(hml_ctx* const* cs, int n, char method, uint64_t iterations, uint64_t thinning) {
    if (!cs || n < 1) return set_err(HML_ERR_ARG, "no contexts");
    for (int i = 0; i < n; ++i) if (!cs[i] || !cs[i]->model_set) return set_err(HML_ERR_ARG, "model not set");
    if (method != HML_METHOD_FB && method != HML_METHOD_MIXTURE) return set_err(HML_ERR_ARG, std::string("Unknown sampling type ") + method + "!");
    for (int i = 0; i < n; ++i) cs[i]->call_base = cs[i]->requested;
    uint64_t done = 0;
    while (done < iterations && many_eligible(cs, n, method)) {
        if (int r = ctx_bind(cs[0])) return r;
        // everything the chains have enqueued on their own streams comes first (a chain with a reduced block capacity catches up
        // on what it skipped: hml_settle); a block count that predates the current parameters is refreshed chain by chain
        // (sweep_k's first-sweep rule)
        for (int i = 0; i < n; ++i) {
            hml_ctx* c = cs[i];
            if (int r = settle_if_limited(c)) return r;
            if (c->hint_stale) { launch_compact_pair(c, 0, 0.0f); KLAUNCH_CHECK(); }
            HIPCHK(hipStreamSynchronize(c->stream));
            c->hint_stale = false;
            if (c->graph_exec) { hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
        }
        int r = 0;
        uint64_t reached = done;
        { HML_KTAB(cs[0]->K, kt); r = kt->iterate_many(cs, n, done, iterations, thinning, &reached); }
        if (r) return r;
        // the other chains' streams continue behind the batch
        hipEvent_t ev = ev_get(cs[0]);
        HIPCHK(hipEventRecord(ev, cs[0]->stream));
        for (int i = 1; i < n; ++i) HIPCHK(hipStreamWaitEvent(cs[i]->stream, ev, 0));
        cs[0]->ev_pool.push_back(ev);
        if (reached == iterations) return 0;
        HIPCHK(hipStreamSynchronize(cs[0]->stream));
        // the batch stopped early: a chain halted (its blocks outgrew its buffers - the loop's first step lets it catch up, then
        // the batch goes on) or left the strongly compressed regime (chain by chain from here)
        bool halted = false;
        for (int i = 0; i < n; ++i) halted = halted || chain_halted(cs[i]);
        const bool progress = reached > done;
        done = reached;
        if (!halted && !progress) break;
        if (!halted) { bool sparse = true; for (int i = 0; i < n; ++i) sparse = sparse && many_sparse(cs[i]); if (!sparse) break; }
    }
    for (int i = 0; i < n; ++i) if (int r = settle_if_limited(cs[i])) return r;
    // not batched (different shapes or devices, mixture sweeps, weakly compressed or reference-compatible chains): sweep by
    // sweep in turn, so that recorded sweeps stay aligned across the chains
    for (uint64_t i = done; i < iterations; ++i) {
        const bool record = thinning > 0 && ((i + 1) % thinning == 0);
        for (int k = 0; k < n; ++k) {
            hml_ctx* c = cs[k];
            if (int r = ctx_bind(c)) return r;
            if (record && c->rec_marginals && c->pooled) return set_err(HML_ERR_ARG, "the marginals of a context are pooled (common labels, several chains): further sweeps cannot be recorded into them");
            if (chain_halted(c)) { if (int r = hml_settle(c)) return r; }
            log_sweep(c, method, record);
            if (int r = sweep_dispatch(c, method, record)) return r;
            if (record && c->cb) {
                HIPCHK(hipStreamSynchronize(c->stream));
                if (chain_halted(c)) { if (int r = hml_settle(c)) return r; continue; }   // (runs the sweep again and calls back)
                if (int r = check_device_error(c)) return r;
                c->cb(c, i, c->cb_user);
            }
        }
    }
    return 0;
}

extern "C" {

int hml_set_option(hml_ctx* c, const char* name, int value) {
    if (!c || !name) return set_err(HML_ERR_ARG, "null argument");
    if (std::string(name) == "weight_keys") {
        if (c->loaded) return set_err(HML_ERR_ARG, "weight_keys must be set before the observations are loaded");
        c->use_keys = value != 0;
        c->summary_always = value == 2;
        return 0;
    }
    if (std::string(name) == "max_blocks") {   // block capacity of the per-block buffers (hml_ctx.hpp)
        if (c->loaded) return set_err(HML_ERR_ARG, "max_blocks must be set before the observations are loaded");
        if (value < 0) return set_err(HML_ERR_ARG, "max_blocks: a number of blocks, or 0 for the default");
        c->cap_opt = (uint64_t)value;
        return 0;
    }
    if (std::string(name) == "fused_blocks") {
        c->fused_blocks = value != 0;
        c->fused_keep = value == 2;
        if (c->graph_exec) { hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
        return 0;
    }
    if (std::string(name) == "compat") {   // the reference-compatible mode (hml_k_compat.h): before hml_set_model
        if (c->model_set) return set_err(HML_ERR_ARG, "compat must be set before the model");
        c->compat = value != 0;
        return 0;
    }
    if (std::string(name) == "trellis_L") {   // chunk length of the fused trellis path: 0 = measured, else a multiple of 32 up to HML_TRE_MAX_L
        if (value < 0 || value > HML_TRE_MAX_L || value % 32) return set_err(HML_ERR_ARG, "trellis_L: 0 or a multiple of 32 up to 1024");
        c->tre_L = (uint32_t)value;
        return 0;
    }
    return set_err(HML_ERR_ARG, std::string("unknown option ") + name);
}

int hml_sync(hml_ctx* c) {
    if (!c) return set_err(HML_ERR_ARG, "null context");
    if (int r = ctx_bind(c)) return r;
    if (int r = hml_settle(c)) return r;   // (synchronises; a halted chain grows and catches up)
    if (c->d_dbg && getenv("HML_FUSED_DEBUG") && atoi(getenv("HML_FUSED_DEBUG")) == 2) {
        // the many-chain block kernel's stamps (8 words per workgroup): start | phase A done | first gathers requested | offsets known | end
        std::vector<unsigned long long> h(4096 * 8);
        hipMemcpy(h.data(), c->d_dbg, h.size() * 8, hipMemcpyDeviceToHost);
        uint32_t n = 0; while (n < 4096 && h[n * 8]) ++n;
        unsigned long long t0 = ~0ull; for (uint32_t i = 0; i < n; ++i) t0 = std::min(t0, h[i * 8]);
        double mx[5] = {0, 0, 0, 0, 0}, av[5] = {0, 0, 0, 0, 0};
        if (n) for (uint32_t i = 0; i < n; ++i) for (int k = 0; k < 5; ++k) { const double d = (double)(h[i * 8 + k] - t0) * 0.01; mx[k] = std::max(mx[k], d); av[k] += d / n; }
        if (n) fprintf(stderr, "[fused-many dbg] n=%u start avg %.2f max %.2f | phaseA avg %.2f max %.2f | requested avg %.2f max %.2f | offsets avg %.2f max %.2f | end avg %.2f max %.2f (us)\n",
                n, av[0], mx[0], av[1], mx[1], av[2], mx[2], av[3], mx[3], av[4], mx[4]);
        if (n) for (uint32_t i : {0u, n / 4, n / 2, n - 1}) if (i < n) fprintf(stderr, "   wg %u: %.2f %.2f %.2f %.2f %.2f | wavefront 0: parameters in LDS %.2f, weights arrived %.2f, its lists done %.2f\n", i, (h[i*8]-t0)*0.01, (h[i*8+1]-t0)*0.01, (h[i*8+2]-t0)*0.01, (h[i*8+3]-t0)*0.01, (h[i*8+4]-t0)*0.01,
                                                                                  (h[i*8+5]-t0)*0.01, (h[i*8+6]-t0)*0.01, (h[i*8+7]-t0)*0.01);
    } else if (c->d_dbg) {
        std::vector<unsigned long long> h(4096 * 4);
        hipMemcpy(h.data(), c->d_dbg, h.size() * 8, hipMemcpyDeviceToHost);
        const uint32_t n = (uint32_t)std::min<uint64_t>(4096, (c->T + HML_FUSED_SUB_POSITIONS - 1) / HML_FUSED_SUB_POSITIONS);
        unsigned long long t0 = ~0ull; for (uint32_t i = 0; i < n; ++i) if (h[i * 4]) t0 = std::min(t0, h[i * 4]);
        double mx[4] = {0, 0, 0, 0}, av[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < n; ++i) for (int k = 0; k < 4; ++k) { const double d = (double)(h[i * 4 + k] - t0) * 0.01; mx[k] = std::max(mx[k], d); av[k] += d / n; }
        fprintf(stderr, "[fused dbg] n=%u start avg %.2f max %.2f | phaseA avg %.2f max %.2f | offsets avg %.2f max %.2f | end avg %.2f max %.2f (us)\n", n, av[0], mx[0], av[1], mx[1], av[2], mx[2], av[3], mx[3]);
        { uint32_t late = 0, first_late = n; for (uint32_t i = 0; i < n; ++i) if ((h[i * 4] - t0) * 0.01 > 2.0) { ++late; first_late = std::min(first_late, i); }
          fprintf(stderr, "   late starters: %u, first index %u\n", late, first_late); }
        for (uint32_t i : {0u, n / 4, n / 2, n - 1}) fprintf(stderr, "   wg %u: %.2f %.2f %.2f %.2f\n", i, (h[i*4]-t0)*0.01, (h[i*4+1]-t0)*0.01, (h[i*4+2]-t0)*0.01, (h[i*4+3]-t0)*0.01);
    }
    if (c->model_set && getenv("HML_PARAMS_DEBUG")) {
        hml_model m;
        if (fetch_model(c, &m) == 0) {
            fprintf(stderr, "[params dbg] us since the kernel's start: accumulators read %.2f | tree %.2f | theta drawn %.2f | A gammas %.2f | barrier %.2f | end %.2f\n",
                    (m.dbg_t[1] - m.dbg_t[0]) * 0.01, (m.dbg_t[2] - m.dbg_t[0]) * 0.01, (m.dbg_t[3] - m.dbg_t[0]) * 0.01, (m.dbg_t[4] - m.dbg_t[0]) * 0.01,
                    (m.dbg_t[5] - m.dbg_t[0]) * 0.01, (m.dbg_t[6] - m.dbg_t[0]) * 0.01);
            fprintf(stderr, "[params dbg]   theta's gamma core + normal drawn %.2f | last wavefront's tree %.2f\n", (m.dbg_t[7] - m.dbg_t[0]) * 0.01, (m.dbg_t[8] - m.dbg_t[0]) * 0.01);
        }
    }
    if (c->model_set) return check_device_error(c);
    return 0;
}

// ---------------------------------------------------------------------------------------- probes
#define NEED_MODEL() if (!c || !c->model_set) return set_err(HML_ERR_ARG, "model not set"); if (int r_ = ctx_bind(c)) return r_; if (int r_ = settle_if_limited(c)) return r_
#define NEED_LOADED() if (!c || !c->loaded) return set_err(HML_ERR_ARG, "no observations loaded"); if (int r_ = ctx_bind(c)) return r_; if (int r_ = settle_if_limited(c)) return r_

static int current_B(hml_ctx* c, uint32_t* B) {
    HIPCHK(hipMemcpyAsync(c->h_B, &c->d_mdl->B, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *B = *c->h_B;
    return 0;
}

int hml_get_num_blocks(hml_ctx* c, uint64_t* B) {
    NEED_LOADED();
    uint32_t b; if (int r = current_B(c, &b)) return r;
    *B = b; return 0;
}
int hml_get_blocks(hml_ctx* c, uint32_t* starts) {
    NEED_LOADED();
    uint32_t b; if (int r = current_B(c, &b)) return r;
    HIPCHK(hipMemcpy(starts, c->d_starts, ((uint64_t)b + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}
int hml_get_block_stats(hml_ctx* c, float* sum, float* sum_sq) {
    NEED_LOADED();
    uint32_t b; if (int r = current_B(c, &b)) return r;
    std::vector<float2> v(b);
    for (int d = 0; d < c->D; ++d) {   // one plane per data dimension
        HIPCHK(hipMemcpy(v.data(), c->d_bstat + (uint64_t)d * c->T, (uint64_t)b * sizeof(float2), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < b; ++i) { sum[(uint64_t)d * b + i] = v[i].x; sum_sq[(uint64_t)d * b + i] = v[i].y; }
    }
    return 0;
}
int hml_get_states(hml_ctx* c, int16_t* q) {
    NEED_MODEL();
    uint32_t b; if (int r = current_B(c, &b)) return r;
    HIPCHK(hipMemcpy(q, c->d_q, (uint64_t)b * sizeof(int16_t), hipMemcpyDeviceToHost));
    return 0;
}
int hml_get_theta(hml_ctx* c, float* mean_var) {
    NEED_MODEL();
    hml_model m; if (int r = fetch_model(c, &m)) return r;
    for (int k = 0; k < m.P; ++k) { mean_var[2 * k] = m.mu[k]; mean_var[2 * k + 1] = m.var[k]; }   // one pair per parameter
    return 0;
}
int hml_get_dimensions(hml_ctx* c, int* D, int* P) {
    if (!c) return set_err(HML_ERR_ARG, "null context");
    if (D) *D = c->D;
    if (P) *P = c->P > 0 ? c->P : c->K;
    return 0;
}
int hml_get_transitions(hml_ctx* c, float* A, float* pi) {
    NEED_MODEL();
    hml_model m; if (int r = fetch_model(c, &m)) return r;
    if (A) memcpy(A, m.A, (size_t)c->K * c->K * sizeof(float));
    if (pi) memcpy(pi, m.pi, (size_t)c->K * sizeof(float));
    return 0;
}
int hml_set_parameters(hml_ctx* c, const float* mean_var, const float* A, const float* pi) {
    NEED_MODEL();
    hml_model m; if (int r = fetch_model(c, &m)) return r;
    const int K = c->K;
    for (int k = 0; k < m.P; ++k) {
        const float mean = mean_var[2 * k], var = mean_var[2 * k + 1];
        if (!std::isfinite(mean)) return set_err(HML_ERR_MODEL, "Mean (" + std::to_string(mean) + ") must be set to a finite value!");
        if (!std::isfinite(var)) return set_err(HML_ERR_MODEL, "Variance(" + std::to_string(var) + ") must be set to a finite value!");
        if (var <= 0) return set_err(HML_ERR_MODEL, "Variance (" + std::to_string(var) + ") must be positive!");
        m.mu[k] = mean; m.var[k] = var; m.sd[k] = sqrtf(var);
    }
    memcpy(m.A, A, (size_t)K * K * sizeof(float));
    memcpy(m.pi, pi, (size_t)K * sizeof(float));
    HIPCHK(hipMemcpyAsync(c->d_mdl, &m, sizeof m, hipMemcpyHostToDevice, c->stream));
    if (c->compat) hipLaunchKernelGGL(hml_k_compat_derive, dim3(1), dim3(64), 0, c->stream, c->d_mdl);   // (glibc's logf)
    else if (c->wide) hipLaunchKernelGGL(hml_k_wide_derive, dim3(1), dim3(64), 0, c->stream, c->d_mdl);
    else { HML_KTAB(K, kt); kt->derive(c); }
    KLAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->dynamic) c->blocks_valid = false;
    return 0;
}
int hml_get_threshold(hml_ctx* c, float* thr) {
    NEED_MODEL();
    hml_model m; if (int r = fetch_model(c, &m)) return r;
    *thr = m.thr; return 0;
}
int hml_get_block_loglik(hml_ctx* c, float* E) {
    NEED_MODEL();
    if (!c->probes) return set_err(HML_ERR_ARG, "probes are not enabled");
    uint32_t b; if (int r = current_B(c, &b)) return r;
    HIPCHK(hipMemcpy(E, c->d_eprobe, (uint64_t)b * c->K * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
int hml_get_forward_rows(hml_ctx* c, float* rows) {
    NEED_MODEL();
    if (!c->probes) return set_err(HML_ERR_ARG, "probes are not enabled");
    uint32_t b; if (int r = current_B(c, &b)) return r;
    HIPCHK(hipMemcpy(rows, c->d_aprobe, ((uint64_t)b + 1) * c->K * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
int hml_get_counts(hml_ctx* c, uint64_t* trans, uint64_t* occ, float* sum, float* sum_sq, uint64_t* nterms) {
    NEED_MODEL();
    hml_model m; if (int r = fetch_model(c, &m)) return r;
    const int K = c->K;
    for (int i = 0; i < K * K; ++i) trans[i] = m.last_trans[i];
    for (int k = 0; k < K; ++k) { occ[k] = m.last_occ[k]; nterms[k] = m.last_occ[k]; sum[k] = m.last_sum[k]; sum_sq[k] = m.last_sumsq[k]; }
    return 0;
}
int hml_get_weights(hml_ctx* c, float* w) {
    NEED_LOADED();
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(w, c->d_w, c->T * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
int hml_get_coefficients(hml_ctx* c, float* cf) {
    NEED_LOADED();
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(cf, c->d_coeff, c->T * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
int hml_get_integral_array(hml_ctx* c, float* sum, float* sum_sq) {
    NEED_LOADED();
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<float2> v(c->T + 1);
    HIPCHK(hipMemcpy(v.data(), c->d_ia, (c->T + 1) * sizeof(float2), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i <= c->T; ++i) { sum[i] = v[i].x; sum_sq[i] = v[i].y; }
    return 0;
}

// ---------------------------------------------------------------------------------------- results
int hml_recorded_sweeps(hml_ctx* c, uint64_t* n) {
    NEED_MODEL();
    hml_model m; if (int r = fetch_model(c, &m)) return r;
    *n = m.n_recorded; return 0;
}

// marginal segments on the device: starts d_seg[M] and count differences d_g[M*K] at the starts (caller frees)
static int gather_marginal_segments(hml_ctx* c, uint64_t* M_out, uint32_t** d_seg_out, int32_t** d_g_out) {
    const uint32_t T = (uint32_t)c->T;
    const int K = c->K;
    DevBuf d_cnt, d_off, d_seg, d_g;   // (released on every early return; the two results are handed over at the end)
    HIPCHK(hipMalloc(&d_cnt.p, c->n_spans * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&d_off.p, c->n_spans * sizeof(uint32_t)));
    hipLaunchKernelGGL(hml_k_marg_count, dim3((c->n_spans + 3) / 4), dim3(256), 0, c->stream, c->d_boundary, T, d_cnt.as<uint32_t>());
    std::vector<uint32_t> h_cnt(c->n_spans), h_off(c->n_spans);
    HIPCHK(hipMemcpyAsync(h_cnt.data(), d_cnt.p, c->n_spans * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint64_t M = 0;
    for (uint32_t i = 0; i < c->n_spans; ++i) { h_off[i] = (uint32_t)M; M += h_cnt[i]; }
    HIPCHK(hipMemcpyAsync(d_off.p, h_off.data(), c->n_spans * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMalloc(&d_seg.p, (M + 1) * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&d_g.p, std::max<uint64_t>(M, 1) * K * sizeof(int32_t)));
    hipLaunchKernelGGL(hml_k_marg_scatter, dim3((c->n_spans + 3) / 4), dim3(256), 0, c->stream, c->d_boundary, T, d_off.as<uint32_t>(), d_seg.as<uint32_t>());
    hipLaunchKernelGGL(hml_k_marg_gather, dim3(grid_for(M, 256, 1, 16384)), dim3(256), 0, c->stream, c->d_diff, T, K, d_seg.as<uint32_t>(),
                       (uint32_t)M, d_g.as<int32_t>());
    KLAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(c->stream));
    *M_out = M; *d_seg_out = d_seg.release<uint32_t>(); *d_g_out = d_g.release<int32_t>();
    return 0;
}

}  // extern "C"
int hml_ctx_gather_marginal_segments(hml_ctx* c, uint64_t* M, uint32_t** d_seg, int32_t** d_g) { return gather_marginal_segments(c, M, d_seg, d_g); }
extern "C" {

int hml_marginals_rle(hml_ctx* c, uint64_t* n_segments, int* n_columns, uint64_t* seg_len, int32_t* counts) {
    NEED_MODEL();
    hml_model m; if (int r = fetch_model(c, &m)) return r;
    if (m.err_code) { char buf[256]; return set_err(HML_ERR_MODEL, deverr_text(m.err_code, m.err_value, buf, sizeof buf)); }
    const uint32_t T = (uint32_t)c->T;
    const int K = c->K;
    const int ncol = m.max_state_recorded + 1;
    if (!c->d_diff || m.n_recorded == 0) {   // nothing recorded: one segment, no count columns
        *n_segments = 1; *n_columns = 0;
        if (seg_len) seg_len[0] = T;
        return 0;
    }
    uint64_t M = 0;
    DevBuf b_seg, b_g;
    { uint32_t* sg = nullptr; int32_t* gg = nullptr; const int r = gather_marginal_segments(c, &M, &sg, &gg); b_seg.p = sg; b_g.p = gg; if (r) return r; }
    *n_segments = M; *n_columns = ncol;
    if (!seg_len) return 0;
    std::vector<uint32_t> h_seg(M);
    std::vector<int32_t> h_g(M * K);
    HIPCHK(hipMemcpyAsync(h_seg.data(), b_seg.p, M * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(h_g.data(), b_g.p, M * K * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // running sums over segments: counts of a segment = sum of the differences at all boundaries up to it
    std::vector<int32_t> cur(K, 0);
    for (uint64_t i = 0; i < M; ++i) {
        for (int s = 0; s < K; ++s) cur[s] += h_g[i * K + s];
        seg_len[i] = (uint64_t)((i + 1 < M ? h_seg[i + 1] : T) - h_seg[i]);
        if (counts) for (int s = 0; s < ncol; ++s) counts[i * ncol + s] = cur[s];
    }
    return 0;
}

int hml_max_segmentation(hml_ctx* c, uint64_t* n_runs, uint64_t* run_len, int32_t* run_state) {
    NEED_MODEL();
    if (!n_runs) return set_err(HML_ERR_ARG, "null argument");
    hml_model m; if (int r = fetch_model(c, &m)) return r;
    if (m.err_code) { char buf[256]; return set_err(HML_ERR_MODEL, deverr_text(m.err_code, m.err_value, buf, sizeof buf)); }
    const uint32_t T = (uint32_t)c->T;
    const int K = c->K;
    if (!c->d_diff || m.n_recorded == 0) {   // nothing recorded: every count is zero, the arg-max is state 0
        *n_runs = 1;
        if (run_len) run_len[0] = T;
        if (run_state) run_state[0] = 0;
        return 0;
    }
    uint64_t M = 0;
    DevBuf b_seg, b_g, b_cs, b_rc, b_st;
    { uint32_t* sg = nullptr; int32_t* gg = nullptr; const int r = gather_marginal_segments(c, &M, &sg, &gg); b_seg.p = sg; b_g.p = gg; if (r) return r; }
    uint32_t* const d_seg = b_seg.as<uint32_t>();
    int32_t* const d_g = b_g.as<int32_t>();
    const uint32_t n_chunks = (uint32_t)((M + 255) / 256);
    HIPCHK(hipMalloc(&b_cs.p, (uint64_t)K * n_chunks * sizeof(int32_t)));
    HIPCHK(hipMalloc(&b_rc.p, ((uint64_t)n_chunks + 1) * sizeof(int32_t)));
    HIPCHK(hipMalloc(&b_st.p, M * sizeof(int16_t)));
    int32_t *const d_cs = b_cs.as<int32_t>(), *const d_rc = b_rc.as<int32_t>();
    int16_t* const d_st = b_st.as<int16_t>();
    HIPCHK(hipMemsetAsync(d_rc + n_chunks, 0, sizeof(int32_t), c->stream));
    hipLaunchKernelGGL(hml_k_seg_partial, dim3(n_chunks), dim3(256), 0, c->stream, d_g, (uint32_t)M, K, d_cs, n_chunks);
    hipLaunchKernelGGL(hml_k_dense_chunkscan, dim3(K), dim3(1024), 0, c->stream, d_cs, n_chunks);
    if (K <= HML_MAX_K) hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_seg_argmax<HML_MAX_K>), dim3(n_chunks), dim3(256), 0, c->stream, d_g, (uint32_t)M, K, d_cs, n_chunks, d_st);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(hml_k_seg_argmax<HML_CAP_K>), dim3(n_chunks), dim3(256), 0, c->stream, d_g, (uint32_t)M, K, d_cs, n_chunks, d_st);
    hipLaunchKernelGGL(hml_k_seg_run_count, dim3(n_chunks), dim3(256), 0, c->stream, d_st, (uint32_t)M, d_rc);
    hipLaunchKernelGGL(hml_k_dense_chunkscan, dim3(1), dim3(1024), 0, c->stream, d_rc, n_chunks + 1u);   // d_rc[n_chunks] = total
    KLAUNCH_CHECK();
    int32_t R = 0;
    HIPCHK(hipMemcpyAsync(&R, d_rc + n_chunks, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *n_runs = (uint64_t)R;
    int rc = 0;
    if (run_len) {
        DevBuf b_rs, b_rq;
        HIPCHK(hipMalloc(&b_rs.p, (uint64_t)R * sizeof(uint32_t)));
        HIPCHK(hipMalloc(&b_rq.p, (uint64_t)R * sizeof(int16_t)));
        uint32_t* const d_rs = b_rs.as<uint32_t>();
        int16_t* const d_rq = b_rq.as<int16_t>();
        hipLaunchKernelGGL(hml_k_seg_run_scatter, dim3(n_chunks), dim3(256), 0, c->stream, d_st, d_seg, (uint32_t)M, d_rc, d_rs, d_rq);
        KLAUNCH_CHECK();
        std::vector<uint32_t> h_rs(R);
        std::vector<int16_t> h_rq(R);
        HIPCHK(hipMemcpyAsync(h_rs.data(), d_rs, (uint64_t)R * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(h_rq.data(), d_rq, (uint64_t)R * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int32_t r = 0; r < R; ++r) {
            run_len[r] = (uint64_t)((r + 1 < R ? h_rs[r + 1] : T) - h_rs[r]);
            if (run_state) run_state[r] = h_rq[r];
        }
    }
    return rc;
}

int hml_marginals_dense_device(hml_ctx* c, void* out_dev, const int32_t* perm) {
    NEED_MODEL();
    const uint32_t T = (uint32_t)c->T;
    const int K = c->K;
    int32_t* out = (int32_t*)out_dev;
    if (!c->d_diff) {
        HIPCHK(hipMemsetAsync(out, 0, (uint64_t)(K + 1) * T * sizeof(int32_t), c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return 0;
    }
    const uint32_t n_chunks = c->n_spans;
    int32_t *d_cs = nullptr, *d_perm = nullptr;
    HIPCHK(hipMalloc(&d_cs, (uint64_t)K * n_chunks * sizeof(int32_t)));
    if (perm) {
        uint64_t seen = 0u;
        for (int k = 0; k < K; ++k) {
            if (perm[k] < 0 || perm[k] >= K || ((seen >> perm[k]) & 1u)) { hipFree(d_cs); return set_err(HML_ERR_ARG, "perm is not a permutation of the K states"); }
            seen |= (uint64_t)1 << perm[k];
        }
        HIPCHK(hipMalloc(&d_perm, K * sizeof(int32_t)));
        HIPCHK(hipMemcpyAsync(d_perm, perm, K * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(hml_k_dense_partial, dim3(n_chunks, K), dim3(256), 0, c->stream, c->d_diff, T, K, d_cs, n_chunks);
    hipLaunchKernelGGL(hml_k_dense_chunkscan, dim3(K), dim3(1024), 0, c->stream, d_cs, n_chunks);
    hipLaunchKernelGGL(hml_k_dense_final, dim3(n_chunks, K), dim3(256), 0, c->stream, c->d_diff, T, K, d_cs, n_chunks, d_perm, out);
    hipLaunchKernelGGL(hml_k_dense_boundary, dim3(grid_for(T, 256, 1, 65536)), dim3(256), 0, c->stream, c->d_boundary, T,
                       out + (uint64_t)K * T);
    KLAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(c->stream));
    hipFree(d_cs);
    if (d_perm) hipFree(d_perm);
    return 0;
}

int hml_relabel_permutation(hml_ctx* c, int32_t* perm) {
    NEED_MODEL();
    if (!perm) return set_err(HML_ERR_ARG, "null argument");
    hml_model m; if (int r = fetch_model(c, &m)) return r;
    const int K = c->K, D = m.D;
    std::vector<int32_t> p(K);
    for (int k = 0; k < K; ++k) p[k] = k;
    // ascending tuple (mean of the parameter mapped to dimension 0, dimension 1, ...); stable: ties keep their order
    std::stable_sort(p.begin(), p.end(), [&](int32_t a, int32_t b) {
        for (int d = 0; d < D; ++d) {
            const float ma = m.mu[m.map[a][d]], mb = m.mu[m.map[b][d]];
            if (ma < mb) return true;
            if (mb < ma) return false;
        }
        return false;
    });
    for (int k = 0; k < K; ++k) perm[k] = p[k];
    return 0;
}

int hml_categorical_draw(hml_ctx* c, const float* weights, int K, uint32_t* index) {
    if (!c || !weights || !index || K < 1) return set_err(HML_ERR_ARG, "invalid argument");
    // std::discrete_distribution with one weight consumes no random number and returns 0 (libstdc++)
    if (K == 1) { *index = 0; return 0; }
    const uint64_t n = c->host_draws++;
    const hml_u32x4 o = hml_stream4(hml_make_key(c->seed, c->chain), HML_KIND_HOST, n >> 32, (uint32_t)n, 0);
    *index = (uint32_t)hml_categorical(weights, K, hml_canonical_f64(o.v[0], o.v[1]));
    return 0;
}

int hml_get_stats(hml_ctx* c, hml_stats* out) {
    NEED_MODEL();
    hml_model m; if (int r = fetch_model(c, &m)) return r;
    out->sweeps = m.sweeps; out->block_updates = m.block_updates; out->uniform_fallbacks = m.uniform_fallbacks;
    out->forward_refits = m.forward_refits; out->forward_serial = m.forward_serial;
    out->forward_warmup = m.fwd_W;
    out->fused_fallbacks = m.fused_fallbacks;
    out->buffer_growths = c->grown; out->block_capacity = c->cap;
    return 0;
}

int hml_profile_enable(hml_ctx* c, int on) {
    if (!c) return set_err(HML_ERR_ARG, "null context");
    if (on && c->ev_pool.size() < 64) {
        // events for the first brackets are created here, not inside the region the caller is about to time
        // (hipEventCreate costs tens of microseconds; a 20-sweep timed region saw four of them in its first sweep)
        if (int r = ctx_bind(c)) return r;
        while (c->ev_pool.size() < 64) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->ev_pool.push_back(e); }
    }
    c->profiling = on;
    return 0;
}

int hml_profile_get(hml_ctx* c, const char* name, double* total_ms, uint64_t* launches) {
    if (!c) return set_err(HML_ERR_ARG, "null context");
    if (int r = ctx_bind(c)) return r;
    HIPCHK(hipStreamSynchronize(c->stream));
    auto& acc = c->prof[name];
    for (auto& p : acc.pending) {
        float ms = 0;
        hipEventElapsedTime(&ms, p.first, p.second);
        acc.ms += ms; acc.n++;
        c->ev_pool.push_back(p.first); c->ev_pool.push_back(p.second);
    }
    acc.pending.clear();
    if (total_ms) *total_ms = acc.ms;
    if (launches) *launches = acc.n;
    return 0;
}

int hml_debug_eval(int device, int fn, const float* a, const float* b, float* out, uint64_t n, uint64_t seed) {
    HIPCHK(hipSetDevice(device));
    float *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
    HIPCHK(hipMalloc(&d_a, n * 4)); HIPCHK(hipMalloc(&d_o, n * 4));
    HIPCHK(hipMemcpy(d_a, a, n * 4, hipMemcpyHostToDevice));
    if (b) { HIPCHK(hipMalloc(&d_b, n * 4)); HIPCHK(hipMemcpy(d_b, b, n * 4, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(hml_k_debug_eval, dim3(1024), dim3(256), 0, 0, fn, d_a, d_b, d_o, n, seed);
    KLAUNCH_CHECK();
    HIPCHK(hipMemcpy(out, d_o, n * 4, hipMemcpyDeviceToHost));
    hipFree(d_a); hipFree(d_o); if (d_b) hipFree(d_b);
    return 0;
}

int hml_synth_gauss(float* x, int16_t* states, uint64_t T, int K, const float* mu, float sigma, double mean_dwell,
                    uint64_t seed, int nthreads) {
    if (!x || !mu || K < 1) return set_err(HML_ERR_ARG, "invalid argument");
    hml_synth_gauss_trace(x, states, T, K, mu, sigma, mean_dwell, seed, nthreads);
    return 0;
}


int hml_synth_depth(float* x, int16_t* states, uint64_t T, double depth, double ln_sigma, uint64_t seed, int nthreads) {
    if (!x) return set_err(HML_ERR_ARG, "invalid argument");
    hml_synth_depth_trace(x, states, T, depth, ln_sigma, seed, nthreads);
    return 0;
}

}  // extern "C"


#if defined(HML_ONLY_K)
#define HML_DECLARE_KTAB_(K) extern hml_ktab hml_ktab_##K; static const hml_ktab* ktab(int k) { return k == K ? &hml_ktab_##K : nullptr; }
#define HML_DECLARE_KTAB(K) HML_DECLARE_KTAB_(K)
HML_DECLARE_KTAB(HML_ONLY_K)
#else
#define HML_EXT_KTAB(K) extern hml_ktab hml_ktab_##K;
HML_EXT_KTAB(2) HML_EXT_KTAB(3) HML_EXT_KTAB(4) HML_EXT_KTAB(5) HML_EXT_KTAB(6) HML_EXT_KTAB(7) HML_EXT_KTAB(8) HML_EXT_KTAB(9)
HML_EXT_KTAB(10) HML_EXT_KTAB(11) HML_EXT_KTAB(12) HML_EXT_KTAB(13) HML_EXT_KTAB(14) HML_EXT_KTAB(15) HML_EXT_KTAB(16)
static const hml_ktab* ktab(int k) {
    static const hml_ktab* const tabs[17] = {nullptr, nullptr, &hml_ktab_2, &hml_ktab_3, &hml_ktab_4, &hml_ktab_5, &hml_ktab_6, &hml_ktab_7, &hml_ktab_8,
                                             &hml_ktab_9, &hml_ktab_10, &hml_ktab_11, &hml_ktab_12, &hml_ktab_13, &hml_ktab_14, &hml_ktab_15, &hml_ktab_16};
    return (k >= 2 && k <= 16) ? tabs[k] : nullptr;
}
#endif
