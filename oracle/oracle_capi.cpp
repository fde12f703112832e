// TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT (see hml_oracle.hpp).
// extern "C" surface of the CPU restatement for the Python tests (ctypes), plus probes for
// the math/RNG/distribution building blocks.
#include <chrono>
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>
#include <random>
#include <sstream>
#include <vector>

#include "../hammlet_amd/csrc/hml_synth_host.hpp"
#include "../hammlet_amd/csrc/hml_text.h"
#include "hml_oracle.hpp"
#include "../hammlet_amd/csrc/hml_math_glibc.h"

using namespace hml_oracle;

static thread_local std::string g_err;

#define ORC_TRY try {
#define ORC_END                                      \
    }                                                \
    catch (std::exception & e) {                     \
        g_err = e.what();                            \
        return 1;                                    \
    }                                                \
    return 0;

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

void* orc_create(int K, float e_var, float e_p, float t_off, float t_diag, float pi_alpha, int self_trans,
                 float weight_mult, uint64_t seed, uint32_t chain, int rng, int math, int reduce) {
    try {
        Config c;
        c.K = K; c.e_var = e_var; c.e_p = e_p; c.t_off = t_off; c.t_diag = t_diag; c.pi_alpha = pi_alpha;
        c.self_trans = self_trans != 0; c.weight_mult = weight_mult; c.seed = seed; c.chain = chain;
        c.rng = rng; c.math = math; c.reduce = reduce;
        return new Oracle(c);
    } catch (std::exception& e) { g_err = e.what(); return nullptr; }
}
void orc_destroy(void* h) { delete (Oracle*)h; }
// "-s C P D": P emission parameters shared by K = P^D states over D interleaved data dimensions (before orc_load)
int orc_set_dims(void* h, int D, int P) { ORC_TRY ((Oracle*)h)->set_dims(D, P); ORC_END }

int orc_load(void* h, const float* x, uint64_t T, int build_pointers) {
    ORC_TRY ((Oracle*)h)->load(x, T, build_pointers != 0); ORC_END
}
int orc_autoprior(void* h, float out4[4]) {
    ORC_TRY Oracle* o = (Oracle*)h; o->autoprior(); memcpy(out4, o->nig_prior, 16); ORC_END
}
int orc_set_prior(void* h, const float p[4]) { ORC_TRY ((Oracle*)h)->set_nig_prior(p); ORC_END }
int orc_init_model(void* h) { ORC_TRY ((Oracle*)h)->init_model(); ORC_END }
int orc_token(void* h, char tok) {
    ORC_TRY Oracle* o = (Oracle*)h;
    if (tok == 'P') o->token_P(); else if (tok == 'S') o->token_S(); else if (tok == 'D') o->token_D(); else o->token_begin();
    ORC_END
}
int orc_set_record(void* h, int marg, int seq, int blocks, int params, int compr) {
    Oracle* o = (Oracle*)h;
    o->rec_marginals = marg; o->rec_sequences = seq; o->rec_blocks = blocks; o->rec_params = params; o->rec_compression = compr;
    return 0;
}
int orc_set_record_segments(void* h, int on) { ((Oracle*)h)->rec_segments = on != 0; return 0; }
int orc_set_probes(void* h, int on) { ((Oracle*)h)->keep_probes = on != 0; return 0; }
int orc_iterate(void* h, char method, uint64_t iters, uint64_t thin) {
    ORC_TRY Oracle* o = (Oracle*)h;
    o->token_begin();
    for (uint64_t i = 0; i < iters; ++i) o->sweep(method, thin > 0 && ((i + 1) % thin == 0));
    ORC_END
}
int orc_enumerate_blocks(void* h, float thr) { ORC_TRY ((Oracle*)h)->enumerate_blocks(thr); ORC_END }

uint64_t orc_T(void* h) { return ((Oracle*)h)->T; }
double orc_sigma_hat(void* h) { return ((Oracle*)h)->sigma_hat; }
float orc_threshold(void* h) { return ((Oracle*)h)->thr; }
uint64_t orc_nblocks(void* h) { Oracle* o = (Oracle*)h; return o->starts.empty() ? 0 : o->starts.size() - 1; }
uint64_t orc_total_blocks(void* h) { return ((Oracle*)h)->total_blocks; }
uint64_t orc_warn_uniform(void* h) { return ((Oracle*)h)->warn_uniform; }
uint64_t orc_n_recorded(void* h) { return ((Oracle*)h)->n_recorded; }

void orc_get_coeffs(void* h, float* out) { Oracle* o = (Oracle*)h; memcpy(out, o->coeffs.data(), o->T * 4); }
void orc_get_weights(void* h, float* out) { Oracle* o = (Oracle*)h; memcpy(out, o->w.data(), o->T * 4); }
void orc_get_integral(void* h, float* s, float* q) {
    Oracle* o = (Oracle*)h; memcpy(s, o->ia_s.data(), (o->T + 1) * 4); memcpy(q, o->ia_q.data(), (o->T + 1) * 4);
}
void orc_get_blocks(void* h, uint32_t* starts) { Oracle* o = (Oracle*)h; memcpy(starts, o->starts.data(), o->starts.size() * 4); }
void orc_get_block_stats(void* h, float* s, float* q) {
    Oracle* o = (Oracle*)h; size_t B = o->starts.size() - 1;
    memcpy(s, o->bs_s.data(), B * 4); memcpy(q, o->bs_q.data(), B * 4);
}
void orc_get_states(void* h, int16_t* q) { Oracle* o = (Oracle*)h; memcpy(q, o->q.data(), o->q.size() * 2); }
void orc_get_theta(void* h, float* mean_var) {
    Oracle* o = (Oracle*)h;
    for (int k = 0; k < o->nP(); ++k) { mean_var[2 * k] = o->mu[k]; mean_var[2 * k + 1] = o->var[k]; }
}
void orc_get_A(void* h, float* A) { Oracle* o = (Oracle*)h; memcpy(A, o->A.data(), o->A.size() * 4); }
void orc_get_pi(void* h, float* pi) { Oracle* o = (Oracle*)h; memcpy(pi, o->pi.data(), o->pi.size() * 4); }
void orc_set_params(void* h, const float* mean_var, const float* A, const float* pi) {
    Oracle* o = (Oracle*)h; const int K = o->cfg.K;
    for (int k = 0; k < o->nP(); ++k) o->set_theta(k, mean_var[2 * k], mean_var[2 * k + 1]);
    memcpy(o->A.data(), A, (size_t)K * K * 4); memcpy(o->pi.data(), pi, K * 4);
    o->sample_prior_pending = false;
}
// emission terms of the CURRENT block list (orc_enumerate_blocks) under the current theta, without the
// self-transition term: innerProduct - N * logNormalizer (EFD.hpp:23-38, ForwardBackward.hpp:74-76)
void orc_eval_emission(void* h, float* E) {
    Oracle* o = (Oracle*)h; const int K = o->cfg.K; const size_t B = o->starts.size() - 1;
    for (size_t b = 0; b < B; ++b) {
        const float N = (float)(o->starts[b + 1] - o->starts[b]);
        for (int s = 0; s < K; ++s) E[b * K + s] = (0.0f + o->emission_ip(o->bs_s[b], o->bs_q[b], s)) - N * o->log_normalizer(s);
    }
}
void orc_get_loglik(void* h, float* E) { Oracle* o = (Oracle*)h; memcpy(E, o->lastE.data(), o->lastE.size() * 4); }
void orc_get_forward_rows(void* h, float* rows) { Oracle* o = (Oracle*)h; memcpy(rows, o->fwd_rows.data(), o->fwd_rows.size() * 4); }
void orc_get_counts(void* h, uint64_t* trans, uint64_t* occ, float* sum_s, float* sum_q, uint64_t* nterms) {
    Oracle* o = (Oracle*)h; const int K = o->cfg.K;
    memcpy(trans, o->last_trans.data(), (size_t)K * K * 8); memcpy(occ, o->last_occ.data(), K * 8);
    memcpy(sum_s, o->last_sum_s.data(), K * 4); memcpy(sum_q, o->last_sum_q.data(), K * 4);
    memcpy(nterms, o->last_nterms.data(), K * 8);
}
void orc_get_posterior(void* h, float* nig4K, float* dirA, float* dirPi) {
    Oracle* o = (Oracle*)h; const int K = o->cfg.K;
    for (int k = 0; k < K; ++k) { nig4K[4 * k] = o->post_alpha[k]; nig4K[4 * k + 1] = o->post_beta[k]; nig4K[4 * k + 2] = o->post_mu0[k]; nig4K[4 * k + 3] = o->post_nu[k]; }
    memcpy(dirA, o->dirA.data(), (size_t)K * K * 4); memcpy(dirPi, o->dirPi.data(), K * 4);
}
int orc_marginals_dense(void* h, int32_t* out) {
    ORC_TRY Oracle* o = (Oracle*)h; std::vector<int32_t> v; o->marginals_dense(v); memcpy(out, v.data(), v.size() * 4); ORC_END
}
// text outputs: which = 0 marginals, 1 sequences, 2 blocks, 3 parameters, 4 compression, 5 segments
uint64_t orc_text(void* h, int which, char* buf, uint64_t cap) {
    Oracle* o = (Oracle*)h;
    std::string s = which == 0 ? o->marginals_text() : which == 1 ? o->out_sequences : which == 2 ? o->out_blocks
                    : which == 3 ? o->out_params : which == 5 ? o->out_segments : o->out_compression;
    if (buf && cap >= s.size()) memcpy(buf, s.data(), s.size());
    return s.size();
}

// --------------------------------------------------------------------- building-block probes
void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    hml_u32x4 o = hml_philox4x32_10(c0, c1, c2, c3, k0, k1);
    memcpy(out, o.v, 16);
}
void orc_expf_dev(const float* x, float* y, uint64_t n) { for (uint64_t i = 0; i < n; ++i) y[i] = hml_expf(x[i]); }
void orc_expf_libm(const float* x, float* y, uint64_t n) { for (uint64_t i = 0; i < n; ++i) y[i] = std::exp(x[i]); }
void orc_logf_dev(const float* x, float* y, uint64_t n) { for (uint64_t i = 0; i < n; ++i) y[i] = hml_logf(x[i]); }
void orc_logf_libm(const float* x, float* y, uint64_t n) { for (uint64_t i = 0; i < n; ++i) y[i] = std::log(x[i]); }
void orc_powf_dev(const float* u, const float* p, float* y, uint64_t n) { for (uint64_t i = 0; i < n; ++i) y[i] = hml_powf_unit(u[i], p[i]); }
void orc_powf_libm(const float* u, const float* p, float* y, uint64_t n) { for (uint64_t i = 0; i < n; ++i) y[i] = std::pow(u[i], p[i]); }
// number of float bit patterns in [lo, hi] (as uint32 ranges of the bit pattern) on which hml_expf != expf
uint64_t orc_expf_mismatches(uint32_t lo, uint32_t hi, uint32_t* first_bad) {
    uint64_t bad = 0;
    for (uint64_t u = lo; u <= hi; ++u) {
        float x = hml_u2f((uint32_t)u);
        float a = hml_expf(x), b = std::exp(x);
        if (hml_f2u(a) != hml_f2u(b) && !(a != a && b != b)) { if (!bad && first_bad) *first_bad = (uint32_t)u; bad++; }
    }
    return bad;
}

// restated distributions vs libstdc++ on the same mt19937 stream; returns number of mismatches
uint64_t orc_check_gamma(uint32_t seed, uint64_t n, float alpha, float beta) {
    std::mt19937 a(seed), b(seed);
    EngineSrc<std::mt19937> src(b);
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        std::gamma_distribution<float> d(alpha, beta);
        float x = d(a);
        float y = hml_gamma_f32<libm_math>(src, alpha, beta);
        if (hml_f2u(x) != hml_f2u(y)) bad++;
    }
    if (a() != b()) bad++;  // engines must have consumed the same number of words
    return bad;
}
uint64_t orc_check_normal(uint32_t seed, uint64_t n, float mean, float sd) {
    std::mt19937 a(seed), b(seed);
    EngineSrc<std::mt19937> src(b);
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        std::normal_distribution<float> d(mean, sd);
        float x = d(a);
        hml_normal_f32<libm_math> nd;
        float y = nd.draw(src, mean, sd);
        if (hml_f2u(x) != hml_f2u(y)) bad++;
    }
    if (a() != b()) bad++;
    return bad;
}
uint64_t orc_check_categorical(uint32_t seed, uint64_t n, int K, int zero_every) {
    std::mt19937 a(seed), b(seed), g(seed ^ 0x1234567u);
    uint64_t bad = 0;
    std::vector<float> w(K);
    for (uint64_t i = 0; i < n; ++i) {
        for (int k = 0; k < K; ++k) {
            float u = (float)(g() >> 8) / 16777216.0f;
            int mode = (int)(g() % 4);
            w[k] = mode == 0 ? u : mode == 1 ? u * 1e-30f : mode == 2 ? 0.0f : u * u * u;
        }
        if (zero_every && (i % zero_every) == 0) for (int k = 0; k < K; ++k) w[k] = 0.0f;
        std::discrete_distribution<size_t> d(w.begin(), w.end());
        size_t x = d(a);
        uint32_t r0 = (uint32_t)b(), r1 = (uint32_t)b();
        int y = hml_categorical(w.data(), K, hml_canonical_f64(r0, r1));
        if ((int)x != y) bad++;
    }
    return bad;
}

// host evaluation of the functions hml_debug_eval runs on the GPU
void orc_debug_eval(int fn, const float* a, const float* b, float* out, uint64_t n, uint64_t seed) {
    for (uint64_t i = 0; i < n; ++i) {
        const float x = a[i], y = b ? b[i] : 0.0f;
        float r = 0.0f;
        switch (fn) {
            case 0: r = hml_expf(x); break;
            case 1: r = hml_logf(x); break;
            case 2: r = hml_powf_unit(x, y); break;
            case 3: r = HML_SQRTF(x); break;
            case 4: r = x / y; break;
            case 5: { StreamSrc src(hml_stream_open(hml_make_key(seed, 0), HML_KIND_THETA, seed, (uint32_t)i)); r = hml_gamma_f32<hml_devmath>(src, x, y); } break;
            case 6: { StreamSrc src(hml_stream_open(hml_make_key(seed, 0), HML_KIND_PI, seed, (uint32_t)i)); hml_normal_f32<hml_devmath> nd; r = nd.draw(src, x, y); } break;
            case 7: r = (float)hml_log((double)x); break;
            case 8: r = (float)hml_exp_nonpos((double)x); break;
            case 9: r = (float)((double)x / (double)y); break;
            case 10: r = (float)(1.0 / (double)x); break;
            case 12: case 13: case 14: case 15: case 16: case 17: case 18: case 19: case 20: case 21: case 22: case 23: {
                StreamSrc src(hml_stream_open(hml_make_key(seed, 0), HML_KIND_THETA, seed, (uint32_t)i));
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
            case 11: { const double d = HML_SQRT((double)x * 1.0000001); r = (float)((d - (double)(float)d) * 1e9); } break;
        }
        out[i] = r;
    }
}

// synthetic trace (product generator, exposed here so oracle-only tests need no GPU library)
void orc_synth_gauss(float* x, int16_t* states, uint64_t T, int K, const float* mu, float sigma, double dwell,
                     uint64_t seed, int nthreads) {
    hml_synth_gauss_trace(x, states, T, K, mu, sigma, dwell, seed, nthreads);
}

void orc_synth_depth(float* x, int16_t* states, uint64_t T, double depth, double ln_sigma, uint64_t seed, int nthreads) {
    hml_synth_depth_trace(x, states, T, depth, ln_sigma, seed, nthreads);
}

// timed run for bench.py's cpu_baseline: returns seconds spent in `iters` sweeps
// The reference's reader, restated: `real_t v; while ( input >> v ) { ... push_back( v ) ... }`
// (reference src/wavelet.hpp:127-134, univariate) over an in-memory text.  Returns the number of values;
// *stopped = 1 if the loop ended on a failed extraction before the end of the text.
uint64_t orc_parse_text(const char* text, uint64_t n, float* out, uint64_t cap, int* stopped) {
    std::istringstream input(std::string(text, text + n));
    std::vector<float> vals;
    float v = 0;
    bool early = false;
    for (;;) {
        // the extraction's sentry skips blanks itself; doing it first tells "text used up" from "extraction failed"
        input >> std::ws;
        if (input.eof()) break;
        if (!(input >> v)) { early = true; break; }
        vals.push_back(v);
    }
    if (stopped) *stopped = early ? 1 : 0;
    const uint64_t k = vals.size() < cap ? vals.size() : cap;
    if (k) memcpy(out, vals.data(), k * sizeof(float));
    return vals.size();
}

// probe of the product's token converter (hml_text.h, compiled by gcc here) against strtof: status per token
// (0 = converted, 1 = left to the stream extraction); tokens are NUL-terminated, `stride` bytes apart
void orc_parse_tokens(const char* toks, uint64_t n, uint32_t stride, float* out, uint8_t* status, float* strtof_out) {
    std::vector<char> buf(stride + HML_TOK_MAX + 2);
    for (uint64_t i = 0; i < n; ++i) {
        const char* t = toks + i * stride;
        const size_t len = strlen(t);
        std::fill(buf.begin(), buf.end(), ' ');
        memcpy(buf.data(), t, len);
        const char* p = buf.data();
        float v = 0;
        status[i] = (uint8_t)hml_parse_token([p](int j) { return (uint32_t)(unsigned char)p[j]; }, HML_TOK_MAX + 1, &v);
        out[i] = v;
        strtof_out[i] = strtof(t, nullptr);
    }
}

// n floats as decimal text, one per line, "%.9g" (round-trips a float): what bench.py feeds the unmodified reference binary
// with on the full trace (numpy.savetxt needs minutes for 10^8 values).  Formatting on `nthreads` threads, one sequential write.
int orc_write_text(const float* x, uint64_t n, const char* path, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    FILE* f = fopen(path, "wb");
    if (!f) return 1;
    const uint64_t piece = 1u << 22;
    std::vector<std::vector<char>> bufs(nthreads);
    int rc = 0;
    for (uint64_t base = 0; base < n && !rc; base += piece * (uint64_t)nthreads) {
        std::vector<std::thread> ths;
        std::vector<size_t> used(nthreads, 0);
        for (int t = 0; t < nthreads; ++t) {
            const uint64_t a = base + piece * (uint64_t)t, b = std::min<uint64_t>(n, a + piece);
            if (a >= n) break;
            ths.emplace_back([&, t, a, b] {
                std::vector<char>& buf = bufs[t];
                buf.resize((size_t)(b - a) * 17);
                size_t o = 0;
                for (uint64_t i = a; i < b; ++i) o += (size_t)snprintf(buf.data() + o, 17, "%.9g\n", (double)x[i]);
                used[t] = o;
            });
        }
        for (auto& th : ths) th.join();
        for (size_t t = 0; t < ths.size(); ++t)
            if (fwrite(bufs[t].data(), 1, used[t], f) != used[t]) { rc = 2; break; }
    }
    if (fclose(f) != 0 && !rc) rc = 3;
    return rc;
}

// hml_math_glibc.h (the reference-compatible mode's expf / logf / powf) against this host's libm: mismatches among the
// float bit patterns [lo, hi) (fn 0: expf, 1: logf) or among n pseudo-random pairs x in [0, 1], y > 0 (fn 2: powf; `lo` seeds).
// -1: this CPU has no fused multiply-add (the functions are built for it; glibc itself would then run its other variant).
static __attribute__((target("fma"))) int64_t glibc_mismatches_fma(int fn, uint64_t lo, uint64_t hi, uint32_t* first_bad) {
    int64_t bad = 0;
    if (fn == 0 || fn == 1) {
        for (uint64_t i = lo; i < hi; ++i) {
            const float x = hml_u2f((uint32_t)i);
            const float a = fn == 0 ? hml_glibc_expf(x) : hml_glibc_logf(x);
            const float b = fn == 0 ? std::exp(x) : std::log(x);
            if (hml_f2u(a) != hml_f2u(b) && !(a != a && b != b)) { if (!bad && first_bad) *first_bad = (uint32_t)i; ++bad; }
        }
    } else {
        uint64_t st = lo * 0x9e3779b97f4a7c15ull + 12345u;
        auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
        for (uint64_t i = 0; i < hi; ++i) {
            const uint64_t r = next();
            // x: a canonical uniform (24 bits) or an arbitrary pattern in [0, 1]; y: 1 / alpha for alpha in (0, 1), or arbitrary positive
            float x = (i & 1) ? (float)(r >> 40) * 5.9604645e-08f : hml_u2f((uint32_t)(r % 0x3f800001u));
            float y = (i & 2) ? 1.0f / (0.001f + 0.999f * (float)((r >> 8) & 0xffffffu) * 5.9604645e-08f) : hml_u2f(0x00800000u + (uint32_t)((r >> 20) % 0x7f000000u));
            const float a = hml_glibc_powf_unit(x, y), b = std::pow(x, y);
            if (hml_f2u(a) != hml_f2u(b) && !(a != a && b != b)) { if (!bad && first_bad) { first_bad[0] = hml_f2u(x); first_bad[1] = hml_f2u(y); } ++bad; }
        }
    }
    return bad;
}
int64_t orc_glibc_mismatches(int fn, uint64_t lo, uint64_t hi, uint32_t* first_bad) {
    if (!__builtin_cpu_supports("fma")) return -1;
    return glibc_mismatches_fma(fn, lo, hi, first_bad);
}

double orc_time_sweeps(void* h, char method, uint64_t iters) {
    Oracle* o = (Oracle*)h;
    o->token_begin();
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t i = 0; i < iters; ++i) o->sweep(method, false);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
