// ============================================================================================
// TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT.
//
// CPU restatement ("oracle") of HaMMLET's Forward-Backward Gibbs path, written from the
// behaviour of the reference sources (cited per function as file:line under /root/reference).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product
// (hammlet_amd/, include/) never includes, links or executes anything from this directory.
//
// Pinning: in reference mode (sequential mt19937 + libstdc++ <random>, glibc math, sequential
// Kahan/float accumulations) this code reproduces the output files of the unmodified reference
// binary (oracle/_ref/hammlet, built by oracle/Makefile from /root/reference/src/main.cpp) byte
// for byte; tests/golden/ holds those files and tests/test_oracle_golden.py re-checks them.
//
// Device mode changes exactly three things, each switchable on its own:
//   rng    : every random decision is addressed by a Philox counter (hml_philox.h) instead of
//            being taken from one sequential engine;
//   math   : expf/logf/powf come from hml_math.h (IEEE basic operations only);
//   reduce : per-state sufficient statistics are summed in double over a fixed tree and the
//            transition/occupancy counts are exact integers (the reference accumulates them
//            through float, src/StateSequence/ForwardBackward.hpp:183-187).
// In device mode the HIP kernels must match this code bit for bit.
// ============================================================================================
#ifndef HML_ORACLE_HPP
#define HML_ORACLE_HPP

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../hammlet_amd/csrc/hml_common.h"
#include "../hammlet_amd/csrc/hml_dist.h"
#include "../hammlet_amd/csrc/hml_math.h"
#include "../hammlet_amd/csrc/hml_philox.h"

#include "philox_seq_engine.hpp"

namespace hml_oracle {

enum RngMode { RNG_MT19937 = 0, RNG_PHILOX_SEQ = 1, RNG_CTR = 2, RNG_MT19937_RESTATED = 3 };
enum MathMode { MATH_LIBM = 0, MATH_DEV = 1 };
enum ReduceMode { REDUCE_REF = 0, REDUCE_DEV = 1 };

// fixed geometry of the device-order reduction (must equal the constants in the HIP kernels)
static const int kReduceChunk = 256;    // blocks per chunk
static const int kReduceGroups = 1024;  // chunk c is accumulated by group c % kReduceGroups

struct libm_math {
    static float logf_(float x) { return std::log(x); }
    static float powf_(float u, float p) { return std::pow(u, p); }
    static float sqrtf_(float x) { return std::sqrt(x); }
};

template <class E>
struct EngineSrc {
    E& e;
    explicit EngineSrc(E& e_) : e(e_) {}
    uint32_t next() { return (uint32_t)e(); }
};

struct StreamSrc {
    hml_stream s;
    explicit StreamSrc(hml_stream s_) : s(s_) {}
    uint32_t next() { return hml_stream_next(&s); }
};

struct Config {
    int K = 3;                               // number of STATES (= P^D with the combinations mapping)
    int D = 1;                               // data dimensions            "-s C P D"   (main.cpp:116-129)
    int P = 0;                               // emission parameters (0: P = K, univariate "-s K")
    float e_var = 0.2f, e_p = 0.9f;          // -e normal VAR P           (main.cpp:46,206)
    float t_off = 0.5f, t_diag = 0.5f;       // -t OFFDIAG DIAG           (main.cpp:144-149)
    float pi_alpha = 0.5f;                   // -I                        (main.cpp:161)
    bool self_trans = true;                  // !-S                       (main.cpp:155)
    float weight_mult = 1.0f;                // -m                        (main.cpp:187)
    uint64_t seed = 0;
    uint32_t chain = 0;
    int rng = RNG_MT19937, math = MATH_LIBM, reduce = REDUCE_REF;
};

struct Kahan2 {  // KahanAggregator<SufficientStatistics<Normal>> (KahanAggregator.hpp:26-45)
    float ps = 0, pq = 0, es = 0, eq = 0;      // positive sums and their errors
    float ns = 0, nq = 0, nes = 0, neq = 0;    // negative sums and their errors
    void add(float s, float q) {
        float y = s - es, t = ps + y; es = (t - ps) - y; ps = t;
        float y2 = q - eq, t2 = pq + y2; eq = (t2 - pq) - y2; pq = t2;
    }
    void sub(float s, float q) {
        float y = s - nes, t = ns + y; nes = (t - ns) - y; ns = t;
        float y2 = q - neq, t2 = nq + y2; neq = (t2 - nq) - y2; nq = t2;
    }
    float sum() const { return ps - ns; }
    float sumSq() const { return pq - nq; }
};

class Oracle {
public:
    Config cfg;
    size_t T = 0;
    std::vector<float> coeffs;     // maxlet coefficients (kept for probes)
    std::vector<float> w;          // breakpoint weights (after multiplier)
    std::vector<float> ia_s, ia_q; // integral array (T+1)
    std::vector<uint16_t> ptr;     // BreakpointArray pointers
    double sigma_hat = 0;
    float thr = 0;
    bool dynamic = true;
    bool sample_prior_pending = true;

    // model
    float nig_prior[4] = {0, 0, 0, 0};
    std::vector<float> post_alpha, post_beta, post_mu0, post_nu;  // NIG posterior per state
    std::vector<float> mu, var, sd;
    std::vector<float> A, pi;              // K*K row-major, K
    std::vector<float> dirA, dirPi;        // Dirichlet posteriors (reset to prior after draws)

    std::mt19937 mt;
    PhiloxSeqEngine pseq;
    uint64_t epoch = 0;

    // last sweep
    std::vector<uint32_t> starts;  // B+1 entries, starts[B] = T
    std::vector<float> bs_s, bs_q; // per-block sums (dimension 0)
    std::vector<float> bsd_s, bsd_q; // per-block sums of every dimension, [b * D + d]
    std::vector<int16_t> q;
    std::vector<float> trellis;    // (B+1)*K, rows as left by the backward pass would be; we keep forward rows (backward-ready)
    std::vector<float> lastE;      // B*K emission log-likelihood terms E_s (parity probe)
    std::vector<float> fwd_rows;   // (B+1)*K normalised forward rows alpha_t before the self-transition rescale
    bool keep_probes = false;
    uint64_t warn_uniform = 0;
    uint64_t total_blocks = 0;     // sum of B over sweeps

    // recording
    bool rec_marginals = true, rec_sequences = false, rec_blocks = false, rec_params = false, rec_compression = false,
         rec_segments = false;
    std::string out_sequences, out_blocks, out_params, out_compression, out_segments;
    // `segments` file (Records.hpp:208-209): marginal segments as sorted starts + the set of states with a non-zero count
    std::vector<uint64_t> mseg_start{0}, mseg_states{0};
    std::vector<int32_t> diff;        // K * (T+1) difference array
    std::vector<uint8_t> boundary;    // T bits as bytes
    int max_state_recorded = -1;
    uint64_t n_recorded = 0;

    explicit Oracle(const Config& c) : cfg(c), mt((std::mt19937::result_type)c.seed), pseq(c.seed) {
        if (c.K < 2) throw std::runtime_error("Requested parameters would yield an HMM with less than 2 states!");
        set_dims(c.D, c.P);
    }

    // Mapping with MappingType combinations (Mapping.hpp:53-137): state x -> for data dimension d the parameter
    // (x / P^d) % P, i.e. reversed P-ary digits; nrStates = P^D (Mapping.hpp:25-48)
    int nD() const { return cfg.D; }
    int nP() const { return cfg.P > 0 ? cfg.P : cfg.K; }
    int map_sd(int state, int d) const { int n = state; for (int i = 0; i < d; ++i) n /= nP(); return n % nP(); }
    void set_dims(int D, int P) {
        if (D <= 0) throw std::runtime_error("Number of data dimensions must be positive!");
        cfg.D = D; cfg.P = P;
        if (D > 1 || P > 0) {
            long k = 1;
            for (int d = 0; d < D; ++d) k *= nP();
            if (k != cfg.K) throw std::runtime_error("number of states must be (number of parameters)^(data dimensions)");
        }
    }

    // ---------------------------------------------------------------- math helpers
    float m_expf(float x) const { return cfg.math == MATH_DEV ? hml_expf(x) : std::exp(x); }
    float m_logf(float x) const { return cfg.math == MATH_DEV ? hml_logf(x) : std::log(x); }

    // ---------------------------------------------------------------- load (A.1-A.5)
    // MaxletTransform (wavelet.hpp:97-188) in closed form, sigma-hat (main.cpp:303-311),
    // HaarBreakpointWeights (wavelet.hpp:68-93), weight multiplier (main.cpp:332-334),
    // integral array (IntegralArray.hpp:136-191, utils.hpp:15-76), pointers (BreakpointArray.hpp:130-184)
    void load(const float* x, size_t n, bool build_pointers = true) {
        if (n == 0) throw std::runtime_error("Input vector for breakpoint weights is empty!");
        const size_t D = (size_t)nD();
        // values of the D dimensions of a position follow each other in the stream (wavelet.hpp:131-137)
        if (n % D != 0) throw std::runtime_error("Input stream did not contain enough values to fill all dimensions at last position!");
        T = n / D;
        maxlet(x);
        // noise estimate: f64 accumulation over odd indices in order
        double acc = 0; size_t cnt = 0;
        for (size_t i = 1; i < T; i += 2) { acc += coeffs[i]; cnt++; }
        acc /= cnt;
        acc /= 0.797884560802865355879892119868763736951717262329869315331;
        sigma_hat = acc;
        weights();
        for (size_t i = 0; i < T; ++i) w[i] *= cfg.weight_mult;
        integral_array(x);
        if (build_pointers) pointers();
    }

    void maxlet(const float* x) {
        // streaming stack form, literally as wavelet.hpp:131-176: the stack holds D values per node, the coefficient
        // of a node is the largest detail coefficient over the dimensions (wavelet.hpp:146-158)
        coeffs.assign(T, 0.0f);
        const size_t D = (size_t)nD();
        const float inf = std::numeric_limits<float>::infinity();
        const float sqrt2 = (float)std::sqrt(2.0);
        const float sqrt2half = (float)(sqrt2 / 2.0);
        std::vector<float> S;
        S.reserve(64 * D);
        for (size_t i = 0; i < T; ++i) {
            for (size_t d = 0; d < D; ++d) S.push_back(x[i * D + d]);
            coeffs[i] = inf;
            size_t j = i, m = 1;
            float normalizer = sqrt2half;
            while ((j & m) > 0) {
                float maxCoeff = 0;
                size_t L = S.size() - 2 * D, R = L + D;
                for (size_t d = 0; d < D; ++d) {
                    maxCoeff = std::max(maxCoeff, normalizer * std::abs(S[L] - S[R]));
                    S[L] += S[R];
                    L++; R++;
                }
                coeffs[j] = maxCoeff;
                for (size_t d = 0; d < D; ++d) S.pop_back();
                j -= m;
                m *= 2;
                normalizer *= sqrt2half;
            }
        }
        coeffs[0] = inf;
    }

    static size_t ceil_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

    void weights() {
        // in-place passes of wavelet.hpp:78-92
        w = coeffs;
        const float inf = std::numeric_limits<float>::infinity();
        const size_t size = T;
        for (size_t interval = ceil_pow2(size) / 2; interval >= 1; interval /= 2) {
            const size_t shift = 2 * interval;
            for (size_t index = interval; index < size; index += shift) {
                size_t L = index - interval, R = index + interval;
                if (R < size) {
                    w[R] = std::max(w[R], w[index]);
                } else {
                    w[L] = inf;
                    w[index] = inf;
                }
                w[L] = std::max(w[L], w[index]);
            }
        }
    }

    void integral_array(const float* x) {
        // one reverse Kahan cumulative sum per cell AND dimension (IntegralArray.hpp:172-186: stride nrDim over the
        // interleaved array, i.e. independent per dimension); stored dimension-major: ia_s[d * (T + 1) + t]
        const size_t D = (size_t)nD(), n = T + 1;
        ia_s.resize(n * D); ia_q.resize(n * D);
        for (size_t d = 0; d < D; ++d) {
            float* is = &ia_s[d * n];
            float* iq = &ia_q[d * n];
            for (size_t i = 0; i < T; ++i) { const float v = x[i * D + d]; is[i] = v; iq[i] = v * v; }
            is[T] = 0; iq[T] = 0;
            for (size_t a = 0; a < n; a += HML_CELLSIZE) {
                size_t right = std::min(a + (size_t)HML_CELLSIZE, n) - 1;
                if (a < right) {
                    float ss = is[right], sq = iq[right], cs = 0, cq = 0;
                    for (size_t i = right - 1;; --i) {
                        float y = is[i] - cs, t = ss + y; cs = (t - ss) - y; ss = t; is[i] = ss;
                        float y2 = iq[i] - cq, t2 = sq + y2; cq = (t2 - sq) - y2; sq = t2; iq[i] = sq;
                        if (i == a) break;
                    }
                }
            }
        }
    }

    void pointers() {
        // monotone stack/deque of BreakpointArray.hpp:146-182, with arrays instead of std::deque
        const size_t n = T;
        const uint16_t maxJump = (uint16_t)std::min(n, (size_t)65535);
        ptr.assign(n, maxJump);
        std::vector<uint32_t> st(n);  // stack storage; front index moves forward
        size_t front = 0, back = 0;   // [front, back)
        st[back++] = 0;
        for (size_t right = 1; right < n; ++right) {
            if (back > front) {
                size_t far = st[front];
                if (right - far == maxJump) { ptr[far] = maxJump; front++; }
            }
            while (back > front) {
                size_t left = st[back - 1];
                if (w[left] <= w[right]) { ptr[left] = (uint16_t)(right - left); back--; }
                else break;
            }
            st[back++] = (uint32_t)right;
        }
        while (back > front) { size_t left = st[back - 1]; ptr[left] = (uint16_t)(n - left); back--; }
    }

    // ---------------------------------------------------------------- blocks & stats (A.6, A.7)
    // Blocks<BreakpointArray>::next (BreakpointArray.hpp:216-235)
    inline size_t next_end(size_t start) const {
        size_t end = start + 1;
        if (!ptr.empty()) {
            while (end < T) {
                if (w[end] < thr) end += ptr[end]; else break;
            }
        } else {
            while (end < T && w[end] < thr) ++end;
        }
        return end;
    }

    // addBlockStats (IntegralArray.hpp:104-124)
    inline void block_stats(size_t start, size_t end, float& s, float& sq, size_t d = 0) const {
        const float* is = &ia_s[d * (T + 1)];
        const float* iq = &ia_q[d * (T + 1)];
        Kahan2 k;
        k.add(is[start], iq[start]);
        for (size_t c = ((start + HML_CELLSIZE) / HML_CELLSIZE) * HML_CELLSIZE; c < end; c += HML_CELLSIZE)
            k.add(is[c], iq[c]);
        if (end % HML_CELLSIZE != 0) k.sub(is[end], iq[end]);
        s = k.sum(); sq = k.sumSq();
    }
    // all dimensions of one block: out[d] / outq[d]
    inline void block_stats_all(size_t start, size_t end, float* out, float* outq) const {
        for (int d = 0; d < nD(); ++d) block_stats(start, end, out[d], outq[d], (size_t)d);
    }

    void enumerate_blocks(float threshold) {
        thr = threshold;
        starts.clear();
        size_t s = 0;
        while (s < T) { starts.push_back((uint32_t)s); s = next_end(s); }
        starts.push_back((uint32_t)T);
        size_t B = starts.size() - 1;
        fill_block_stats(B);
    }
    // bs_s/bs_q[b]: dimension 0 (probes of the univariate path); bsd_s/bsd_q[b * D + d]: every dimension
    void fill_block_stats(size_t B) {
        const size_t D = (size_t)nD();
        bs_s.resize(B); bs_q.resize(B); bsd_s.resize(B * D); bsd_q.resize(B * D);
        for (size_t b = 0; b < B; ++b) {
            block_stats_all(starts[b], starts[b + 1], &bsd_s[b * D], &bsd_q[b * D]);
            bs_s[b] = bsd_s[b * D]; bs_q[b] = bsd_q[b * D];
        }
    }

    // ---------------------------------------------------------------- auto prior (A.8)
    // autoPrior (AutoPriors.hpp:86-110) + NormalInverseGammaAutoPrior (AutoPriors.hpp:18-80)
    void autoprior() {
        thr = (float)(std::sqrt(2 * std::log((double)T)) * sigma_hat);
        float muSum = 0, muSq = 0;
        size_t B = 0;
        size_t s = 0;
        const int D = nD();
        while (s < T) {
            size_t e = next_end(s);
            // one observation per block AND dimension (AutoPriors.hpp:100-104), N = nrBlocks * nrDim (:105)
            for (int d = 0; d < D; ++d) {
                float bsum, bsq;
                block_stats(s, e, bsum, bsq, (size_t)d);
                float m = bsum / (float)(e - s);
                muSum += m;
                muSq += m * m;
            }
            ++B;
            s = e;
        }
        double n = (double)(B * (size_t)D);
        double blocksMean = (double)(float)(muSum / n);
        double avg = (double)(float)(muSum / n);
        double blocksVar = (double)(float)(muSq / n - (avg * avg));
        nig_autoprior(cfg.e_var, cfg.e_p, (float)blocksMean, (float)blocksVar, nig_prior);
        reset_theta_post();
    }

    static void nig_autoprior(float s2, float p, float dataMean, float dataVar, float out[4]) {
        if (p < 0 || p > 1) throw std::runtime_error("Parameter p for automatic priors is a probability and must be in [0,1]!");
        if (s2 <= 0) throw std::runtime_error("Parameter s2  for automatic priors is a variance and must be positive!");
        if (dataVar <= 0) throw std::runtime_error("Data variance provided to autoprior must be positive!");
        const float M1 = 0.3361, M2 = -0.0042, M3 = -0.0201;
        const float b = -std::log(p);
        const float alpha = 2.0;
        const float beta = s2 * ((2.0 * std::sqrt(b)) / (M1 * std::sqrt(b) + std::sqrt(2.0) * (M2 * b * std::exp(M3 * std::sqrt(b)) + 1)) + b);
        const float mu0 = dataMean;
        const float nu = beta / dataVar;
        if (beta <= 0) throw std::runtime_error("Autoprior yields non-positive beta!");
        if (nu <= 0) throw std::runtime_error("Autoprior yields non-positive nu!");
        if (!std::isfinite(beta)) throw std::runtime_error("Autoprior yields non-finite beta!");
        if (!std::isfinite(mu0)) throw std::runtime_error("Autoprior yields non-finite mu0!");
        if (!std::isfinite(nu)) throw std::runtime_error("Autoprior yields non-finite nu!");
        out[0] = alpha; out[1] = beta; out[2] = mu0; out[3] = nu;
    }

    void set_nig_prior(const float p[4]) { for (int i = 0; i < 4; ++i) nig_prior[i] = p[i]; reset_theta_post(); }

    void reset_theta_post() {
        const int P = nP();   // one prior per emission parameter (main.cpp:201-209)
        post_alpha.assign(P, nig_prior[0]); post_beta.assign(P, nig_prior[1]);
        post_mu0.assign(P, nig_prior[2]); post_nu.assign(P, nig_prior[3]);
    }
    void reset_dir_post() {
        const int K = cfg.K;
        dirA.assign((size_t)K * K, cfg.t_off);
        for (int i = 0; i < K; ++i) dirA[(size_t)i * K + i] = cfg.t_diag;
        dirPi.assign(K, cfg.pi_alpha);
    }

    // ---------------------------------------------------------------- model init (A.9)
    // Objects as built in main.cpp:152-166,354-362; Theta's constructor draws once (Theta.hpp:126-127).
    void init_model() {
        const int K = cfg.K;
        mu.assign(nP(), NAN); var.assign(nP(), NAN); sd.assign(nP(), NAN);
        A.assign((size_t)K * K, NAN); pi.assign(K, NAN);
        reset_dir_post();
        reset_theta_post();
        draw_theta();   // constructor draw
        epoch++;
    }

    // ---------------------------------------------------------------- parameter draws
    template <class Src, class M>
    void draw_theta_with(Src& src, int k) {
        // Distribution<NormalInverseGamma>::resample (Distribution.hpp:77-87)
        float g = hml_gamma_f32<M>(src, post_alpha[k], (float)(1.0 / post_beta[k]));
        float v = (float)(1.0 / g);
        hml_normal_f32<M> nd;
        float m = nd.draw(src, post_mu0[k], M::sqrtf_(v / post_nu[k]));
        set_theta(k, m, v);
    }
    void set_theta(int k, float m, float v) {
        if (!std::isfinite(m)) throw std::runtime_error("Mean (" + std::to_string(m) + ") must be set to a finite value!");
        if (!std::isfinite(v)) throw std::runtime_error("Variance(" + std::to_string(v) + ") must be set to a finite value!");
        if (v <= 0) throw std::runtime_error("Variance (" + std::to_string(v) + ") must be positive!");
        mu[k] = m; var[k] = v;
        sd[k] = cfg.math == MATH_DEV ? HML_SQRTF(v) : std::sqrt(v);
    }
    template <class E>
    void draw_theta_std(E& eng) {
        for (int k = 0; k < nP(); ++k) {
            std::gamma_distribution<float> gamma(post_alpha[k], 1.0 / post_beta[k]);
            float v = 1.0 / gamma(eng);
            std::normal_distribution<float> normal(post_mu0[k], std::sqrt(v / post_nu[k]));
            float m = normal(eng);
            set_theta(k, m, v);
        }
    }
    void draw_theta() {
        const int K = nP();   // Theta::sample draws every PARAMETER in order (Theta.hpp:203-211)
        switch (cfg.rng) {
            case RNG_MT19937: draw_theta_std(mt); break;
            case RNG_PHILOX_SEQ: draw_theta_std(pseq); break;
            case RNG_MT19937_RESTATED: { EngineSrc<std::mt19937> s(mt); for (int k = 0; k < K; ++k) draw_theta_with<EngineSrc<std::mt19937>, libm_math>(s, k); break; }
            default:
                for (int k = 0; k < K; ++k) {
                    StreamSrc s(hml_stream_open(hml_make_key(cfg.seed, cfg.chain), HML_KIND_THETA, epoch, (uint32_t)k));
                    if (cfg.math == MATH_DEV) draw_theta_with<StreamSrc, hml_devmath>(s, k);
                    else draw_theta_with<StreamSrc, libm_math>(s, k);
                }
        }
        reset_theta_post();
    }

    template <class E>
    void dirichlet_std(E& eng, const float* alphas, float* probs, int n) {
        // dirichlet_sample (Distribution.hpp:116-139)
        float sum = 0;
        for (int d = 0; d < n; ++d) {
            std::gamma_distribution<float> dist(alphas[d], 1.0);
            float r = dist(eng);
            probs[d] = r;
            sum += r;
        }
        for (int d = 0; d < n; ++d) probs[d] /= sum;
    }
    template <class Src, class M>
    void dirichlet_src(Src& src, const float* alphas, float* probs, int n) {
        float sum = 0;
        for (int d = 0; d < n; ++d) { float r = hml_gamma_f32<M>(src, alphas[d], 1.0f); probs[d] = r; sum += r; }
        for (int d = 0; d < n; ++d) probs[d] /= sum;
    }
    template <class M>
    void dirichlet_ctr(uint32_t kind, uint32_t base, const float* alphas, float* probs, int n) {
        float sum = 0;
        for (int d = 0; d < n; ++d) {
            StreamSrc s(hml_stream_open(hml_make_key(cfg.seed, cfg.chain), kind, epoch, base + (uint32_t)d));
            float r = hml_gamma_f32<M>(s, alphas[d], 1.0f);
            probs[d] = r; sum += r;
        }
        for (int d = 0; d < n; ++d) probs[d] /= sum;
    }
    void draw_pi() {
        const int K = cfg.K;
        switch (cfg.rng) {
            case RNG_MT19937: dirichlet_std(mt, dirPi.data(), pi.data(), K); break;
            case RNG_PHILOX_SEQ: dirichlet_std(pseq, dirPi.data(), pi.data(), K); break;
            case RNG_MT19937_RESTATED: { EngineSrc<std::mt19937> s(mt); dirichlet_src<EngineSrc<std::mt19937>, libm_math>(s, dirPi.data(), pi.data(), K); break; }
            default:
                if (cfg.math == MATH_DEV) dirichlet_ctr<hml_devmath>(HML_KIND_PI, 0, dirPi.data(), pi.data(), K);
                else dirichlet_ctr<libm_math>(HML_KIND_PI, 0, dirPi.data(), pi.data(), K);
        }
        dirPi.assign(K, cfg.pi_alpha);
    }
    void draw_A() {
        const int K = cfg.K;
        for (int i = 0; i < K; ++i) {
            const float* al = &dirA[(size_t)i * K];
            float* pr = &A[(size_t)i * K];
            switch (cfg.rng) {
                case RNG_MT19937: dirichlet_std(mt, al, pr, K); break;
                case RNG_PHILOX_SEQ: dirichlet_std(pseq, al, pr, K); break;
                case RNG_MT19937_RESTATED: { EngineSrc<std::mt19937> s(mt); dirichlet_src<EngineSrc<std::mt19937>, libm_math>(s, al, pr, K); break; }
                default:
                    if (cfg.math == MATH_DEV) dirichlet_ctr<hml_devmath>(HML_KIND_TRANS, (uint32_t)(i * K), al, pr, K);
                    else dirichlet_ctr<libm_math>(HML_KIND_TRANS, (uint32_t)(i * K), al, pr, K);
            }
        }
        const float off = cfg.t_off, dg = cfg.t_diag;
        for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) dirA[(size_t)i * K + j] = (i == j) ? dg : off;
    }

    // theta, pi, A from the (reset) priors (main.cpp:393-401)
    void sample_prior() {
        draw_theta();
        draw_pi();
        draw_A();
        epoch++;
        sample_prior_pending = false;
    }

    // ---------------------------------------------------------------- scheme tokens (main.cpp:391-454)
    void token_begin() { if (sample_prior_pending) sample_prior(); }
    void token_P() { token_begin(); sample_prior_pending = true; }
    void token_S() { token_begin(); thr = threshold_from_theta(); dynamic = false; }
    void token_D() { token_begin(); dynamic = true; }

    // createBlocks(theta) (BreakpointArray.hpp:196-199, Theta.hpp:227-234)
    float threshold_from_theta() const {
        float mv = std::numeric_limits<float>::infinity();
        for (int k = 0; k < nP(); ++k) mv = std::min(mv, var[k]);
        float l = m_logf((float)T);
        float arg = 2 * l * mv;
        return cfg.math == MATH_DEV ? HML_SQRTF(arg) : std::sqrt(arg);
    }

    // ---------------------------------------------------------------- categorical
    int categorical_seq(const float* wts, int K) {
        if (cfg.rng == RNG_MT19937) { std::discrete_distribution<size_t> d(wts, wts + K); return (int)d(mt); }
        if (cfg.rng == RNG_PHILOX_SEQ) { std::discrete_distribution<size_t> d(wts, wts + K); return (int)d(pseq); }
        // restated on mt19937
        uint32_t r0 = (uint32_t)mt(), r1 = (uint32_t)mt();
        return hml_categorical(wts, K, hml_canonical_f64(r0, r1));
    }
    int categorical_ctr(uint32_t kind, uint32_t index, const float* wts, int K) const {
        // backward rows: two consecutive blocks share one Philox block (hml_cat_uniform, hml_dist.h)
        if (kind == HML_KIND_CAT) return hml_categorical(wts, K, hml_cat_uniform(hml_make_key(cfg.seed, cfg.chain), epoch, index));
        hml_u32x4 o = hml_stream4(hml_make_key(cfg.seed, cfg.chain), kind, epoch, index, 0);
        return hml_categorical(wts, K, hml_canonical_f64(o.v[0], o.v[1]));
    }

    // ---------------------------------------------------------------- one sweep (A.10 / A.11)
    // sampleHMM body (HMM.hpp:99-121); method 'F' = StateSequence<ForwardBackward>::sample
    // (ForwardBackward.hpp:16-213), 'M' = StateSequence<Mixture>::sample (Mixture.hpp:31-144).
    void sweep(char method, bool record) {
        if (dynamic) thr = threshold_from_theta();
        if (method == 'F') sweep_fb(record); else sweep_mix(record);
        draw_theta();
        draw_pi();
        draw_A();
        epoch++;
        if (record && rec_params) append_params();
    }

    inline float emission_ip(float s, float sq, int k) const {
        // innerProduct (EFD.hpp:23-32): double inside, float result
        float r = (float)((2.0 * mu[k] * s - sq) / (2.0 * var[k]));
        if (!std::isfinite(r)) throw std::runtime_error("Result of Normal inner product is not finite!");
        return r;
    }
    inline float log_normalizer(int k) const {  // EFD.hpp:35-38
        return m_logf(sd[k]) + mu[k] * mu[k] / (2 * var[k]);
    }
    // Theta::logNormalizer(state) (Theta.hpp:148-158): float sum over the state's parameters, in dimension order
    inline float log_normalizer_state(int s) const {
        float r = 0;
        for (int d = 0; d < nD(); ++d) r += log_normalizer(map_sd(s, d));
        return r;
    }
    // innerProduct(y, theta.value(), theta.mapping(s)) (EFD.hpp:83-93): float sum over the dimensions, from 0
    inline float emission_ip_state(const float* bsum, const float* bsq, int s) const {
        float r = 0;
        for (int d = 0; d < nD(); ++d) r += emission_ip(bsum[d], bsq[d], map_sd(s, d));
        return r;
    }

    void sweep_fb(bool record) {
        const int K = cfg.K;
        std::vector<float> logA(K, 0.0f), logN(K);
        for (int s = 0; s < K; ++s) {
            if (cfg.self_trans) logA[s] = m_logf(A[(size_t)s * K + s]);
            logN[s] = log_normalizer_state(s);
        }
        trellis.clear();
        trellis.insert(trellis.end(), pi.begin(), pi.end());
        starts.clear();
        if (keep_probes) { lastE.clear(); fwd_rows.assign(pi.begin(), pi.end()); }
        std::vector<float> forward(K, 0.0f);
        float prevN = 1;
        size_t pos = 0;
        size_t t = 0;
        while (pos < T) {
            size_t end = next_end(pos);
            float bsum[HML_MAX_K], bsq[HML_MAX_K];
            block_stats_all(pos, end, bsum, bsq);
            starts.push_back((uint32_t)pos);
            ++t;
            float maxE = std::numeric_limits<float>::lowest();
            float N = (float)(end - pos);
            for (int s = 0; s < K; ++s) {
                float E = emission_ip_state(bsum, bsq, s) - N * logN[s];
                if (cfg.self_trans) E += (N - 1) * logA[s];
                forward[s] = E;
                maxE = std::max(E, maxE);
            }
            if (keep_probes) lastE.insert(lastE.end(), forward.begin(), forward.end());
            for (int s = 0; s < K; ++s) forward[s] = m_expf(forward[s] - maxE);
            float forwardSum = 0;
            const float* prev = &trellis[(t - 1) * K];
            for (int j = 0; j < K; ++j) {
                float tt = 0;
                for (int i = 0; i < K; ++i) tt += prev[i] * A[(size_t)i * K + j];
                forward[j] *= tt;
                forwardSum += forward[j];
            }
            if (forwardSum != 0) {
                for (int j = 0; j < K; ++j) forward[j] /= forwardSum;
            } else {
                warn_uniform++;
                for (int j = 0; j < K; ++j) forward[j] = (float)(1.0 / ((float)K));
            }
            if (keep_probes) fwd_rows.insert(fwd_rows.end(), forward.begin(), forward.end());
            if (cfg.self_trans) {
                float* back = &trellis[(t - 1) * K];
                for (int s = 0; s < K; ++s) back[s] *= m_expf((prevN - 1) * logA[s]);
            }
            trellis.insert(trellis.end(), forward.begin(), forward.end());
            prevN = N;
            pos = end;
        }
        starts.push_back((uint32_t)T);
        const size_t B = t;
        total_blocks += B;

        // backward sampling (ForwardBackward.hpp:133-162)
        q.resize(B);
        std::vector<float> row(K);
        int j;
        if (cfg.rng == RNG_CTR) j = categorical_ctr(HML_KIND_CAT, (uint32_t)B, &trellis[B * K], K);
        else j = categorical_seq(&trellis[B * K], K);
        q[B - 1] = (int16_t)j;
        for (size_t tt = B - 1; tt > 0; --tt) {
            float* r = &trellis[tt * K];
            for (int i = 0; i < K; ++i) {
                r[i] = r[i] * A[(size_t)i * K + j];
                if (r[i] < 0) throw std::runtime_error("Negative backward variable!");
            }
            if (cfg.rng == RNG_CTR) j = categorical_ctr(HML_KIND_CAT, (uint32_t)tt, r, K);
            else j = categorical_seq(r, K);
            q[tt - 1] = (int16_t)j;
        }
        count_pass(record, /*mixture=*/false);
    }

    void sweep_mix(bool record) {
        const int K = cfg.K;
        std::vector<float> logN(K);
        for (int s = 0; s < K; ++s) logN[s] = log_normalizer_state(s);
        starts.clear();
        q.clear();
        if (keep_probes) lastE.clear();
        std::vector<float> wts(K);
        size_t pos = 0, b = 0;
        while (pos < T) {
            size_t end = next_end(pos);
            float bsum[HML_MAX_K], bsq[HML_MAX_K];
            block_stats_all(pos, end, bsum, bsq);
            starts.push_back((uint32_t)pos);
            float maxE = std::numeric_limits<float>::lowest();
            const size_t N = end - pos;
            for (int s = 0; s < K; ++s) {
                float E = emission_ip_state(bsum, bsq, s) - N * logN[s];
                wts[s] = E;
                maxE = std::max(E, maxE);
            }
            if (keep_probes) lastE.insert(lastE.end(), wts.begin(), wts.end());
            for (int s = 0; s < K; ++s) wts[s] = m_expf(wts[s] - maxE);
            int st;
            if (cfg.rng == RNG_CTR) st = categorical_ctr(HML_KIND_MIX, (uint32_t)b, wts.data(), K);
            else st = categorical_seq(wts.data(), K);
            q.push_back((int16_t)st);
            pos = end; ++b;
        }
        starts.push_back((uint32_t)T);
        total_blocks += b;
        count_pass(record, /*mixture=*/true);
    }

    // posterior counting pass (ForwardBackward.hpp:170-211 / Mixture.hpp:113-141) and conjugate
    // updates (Conjugate.hpp:121-168,178-205)
    void count_pass(bool record, bool mixture) {
        const int K = cfg.K, P = nP(), D = nD();
        const size_t B = q.size();
        std::vector<uint64_t> trans((size_t)K * K, 0), occ(K, 0), nterms(P, 0);
        std::vector<float> sum_s(P, 0.0f), sum_q(P, 0.0f);
        fill_block_stats(B);
        if (cfg.reduce == REDUCE_REF) {
            std::vector<Kahan2> st(P);
            int prev = 0;
            for (size_t b = 0; b < B; ++b) {
                const size_t n = starts[b + 1] - starts[b];
                const int s = q[b];
                if (mixture) {
                    occ[s] += n;
                    trans[(size_t)s * K + s] += n - 1;
                } else {
                    const float N = (float)n;
                    trans[(size_t)s * K + s] = (uint64_t)((float)trans[(size_t)s * K + s] + (N - 1));
                    occ[s] = (uint64_t)((float)occ[s] + N);
                }
                // `+= 1` happens after the diagonal update in the reference
                trans[(size_t)prev * K + s] += 1;
                // stats[mapping[state][d]].add(y.suffStat(d), N) for every dimension in order (ForwardBackward.hpp:189-192)
                for (int d = 0; d < D; ++d) {
                    const int pp = map_sd(s, d);
                    st[pp].add(bsd_s[b * D + d], bsd_q[b * D + d]);
                    nterms[pp] += n;
                }
                prev = s;
            }
            for (int pp = 0; pp < P; ++pp) { sum_s[pp] = st[pp].sum(); sum_q[pp] = st[pp].sumSq(); }
        } else {
            // device order (hml_k_counts): block b belongs to chunk c = b / 256, wavefront w = (b / 64) % 4, lane l = b % 64 and
            // group g = c % 1024.  Accumulator (g, w, l) adds the terms of its blocks in increasing block order (a lane's
            // term for parameter p is the sum, in dimension order, of the block's statistics of the dimensions mapped to
            // p; blocks in other states add nothing); then a pairwise tree over the 64 lanes, the four wavefronts in
            // order, and a final pairwise tree over the groups.
            const size_t nchunks = (B + kReduceChunk - 1) / kReduceChunk;
            std::vector<double> gs((size_t)kReduceGroups * P, 0.0), gq((size_t)kReduceGroups * P, 0.0);
            {
                // (groups beyond the number of chunks stay empty: their sums are the +0.0 the vectors start with)
                const size_t active = nchunks < (size_t)kReduceGroups ? nchunks : (size_t)kReduceGroups;
                const size_t nacc = active * (kReduceChunk / 64) * 64;
                std::vector<double> as(nacc * P, 0.0), aq(nacc * P, 0.0);
                for (size_t b = 0; b < B; ++b) {
                    const size_t c = b / kReduceChunk, g = c % kReduceGroups, wl = b % kReduceChunk;   // wl = w * 64 + l
                    double* ps = &as[(g * kReduceChunk + wl) * P];
                    double* pq = &aq[(g * kReduceChunk + wl) * P];
                    if (D == 1) { ps[q[b]] = ps[q[b]] + (double)bsd_s[b]; pq[q[b]] = pq[q[b]] + (double)bsd_q[b]; }
                    else {
                        for (int pp = 0; pp < P; ++pp) {
                            double ts = 0.0, tq = 0.0;
                            bool any = false;
                            for (int d = 0; d < D; ++d) if (map_sd(q[b], d) == pp) { ts = ts + (double)bsd_s[b * D + d]; tq = tq + (double)bsd_q[b * D + d]; any = true; }
                            if (any) { ps[pp] = ps[pp] + ts; pq[pp] = pq[pp] + tq; }
                        }
                    }
                }
                std::vector<double> ls(64), lq(64);
                for (size_t g = 0; g < active; ++g)
                    for (int pp = 0; pp < P; ++pp) {
                        double cs = 0.0, cq = 0.0;
                        for (int wv = 0; wv < kReduceChunk / 64; ++wv) {
                            for (int l = 0; l < 64; ++l) { ls[l] = as[(g * kReduceChunk + wv * 64 + l) * P + pp]; lq[l] = aq[(g * kReduceChunk + wv * 64 + l) * P + pp]; }
                            for (int stride = 1; stride < 64; stride <<= 1)
                                for (int l = 0; l < 64; l += 2 * stride) { ls[l] = ls[l] + ls[l + stride]; lq[l] = lq[l] + lq[l + stride]; }
                            cs = cs + ls[0]; cq = cq + lq[0];
                        }
                        gs[g * P + pp] = cs;
                        gq[g * P + pp] = cq;
                    }
            }
            for (int pp = 0; pp < P; ++pp) {
                std::vector<double> a2(kReduceGroups), b2(kReduceGroups);
                for (int g = 0; g < kReduceGroups; ++g) { a2[g] = gs[(size_t)g * P + pp]; b2[g] = gq[(size_t)g * P + pp]; }
                for (int stride = 1; stride < kReduceGroups; stride <<= 1)
                    for (int g = 0; g < kReduceGroups; g += 2 * stride) { a2[g] = a2[g] + a2[g + stride]; b2[g] = b2[g] + b2[g + stride]; }
                sum_s[pp] = (float)a2[0]; sum_q[pp] = (float)b2[0];
            }
            int prev = 0;
            for (size_t b = 0; b < B; ++b) {
                const size_t n = starts[b + 1] - starts[b];
                const int s = q[b];
                occ[s] += n; trans[(size_t)s * K + s] += n - 1; trans[(size_t)prev * K + s] += 1;
                for (int d = 0; d < D; ++d) nterms[map_sd(s, d)] += n;
                prev = s;
            }
        }
        if (record) record_sweep();
        // tau_theta.addObservation per parameter (ForwardBackward.hpp:202-207, Conjugate.hpp:121-168)
        for (int pp = 0; pp < P; ++pp) {
            if (nterms[pp] > 0) nig_update(pp, sum_s[pp], sum_q[pp], nterms[pp]);
        }
        for (int i = 0; i < K; ++i) for (int j2 = 0; j2 < K; ++j2) dirA[(size_t)i * K + j2] += (float)trans[(size_t)i * K + j2];
        for (int i = 0; i < K; ++i) dirPi[i] += (float)occ[i];
        last_trans = trans; last_occ = occ; last_sum_s = sum_s; last_sum_q = sum_q; last_nterms = nterms;
    }
    std::vector<uint64_t> last_trans, last_occ, last_nterms;
    std::vector<float> last_sum_s, last_sum_q;

    void nig_update(int p, float sum, float sumSq, uint64_t counts) {
        if (sumSq < 0) throw std::runtime_error("Sum of squares is negative (" + std::to_string(sumSq) + ") for " + std::to_string(counts) + " observations!");
        const double N = (double)counts;
        const float xbar = sum / N;
        const float alpha = post_alpha[p], beta = post_beta[p], mu0 = post_mu0[p], nu = post_nu[p];
        float ssN = (sum * sum) / N;
        if (ssN > sumSq) ssN = sumSq;
        float na = alpha + N / 2.0;
        float nb = beta + ((sumSq + (N * nu / (N + nu)) * ((xbar - mu0) * (xbar - mu0))) - ssN) / 2.0;
        float nm = (nu * mu0 + sum) / (nu + N);
        float nn = nu + N;
        if (na <= 0) throw std::runtime_error("Alpha (" + std::to_string(na) + ") must be positive!");
        if (nb <= 0) throw std::runtime_error("Beta (" + std::to_string(nb) + ") must be positive!");
        if (nn <= 0) throw std::runtime_error("Nu (" + std::to_string(nn) + ")must be positive!");
        if (!std::isfinite(nm)) throw std::runtime_error("Mu0 (" + std::to_string(nm) + ")  must be finite!");
        post_alpha[p] = na; post_beta[p] = nb; post_mu0[p] = nm; post_nu[p] = nn;
    }

    // ---------------------------------------------------------------- recording (A.10-9, A.12)
    // Records::record (Records.hpp:155-235) + StateMarginals (StateMarginals.hpp:51-137) semantics:
    // segments = maximal runs of equal state; marginal boundaries = union of segment boundaries.
    void record_sweep() {
        const int K = cfg.K;
        const size_t B = q.size();
        if (rec_marginals && diff.empty()) { diff.assign((size_t)K * (T + 1), 0); boundary.assign(T + 1, 0); }
        char buf[64];
        size_t segStart = 0;
        bool firstSeg = true;
        std::vector<uint64_t> run_start;
        std::vector<int> run_state;
        for (size_t b = 0; b < B; ++b) {
            if (rec_blocks) {
                snprintf(buf, sizeof buf, "%s%zu", b == 0 ? "" : "\t", (size_t)(starts[b + 1] - starts[b]));
                out_blocks += buf;
            }
            const bool last = (b + 1 == B);
            if (last || q[b + 1] != q[b]) {
                const size_t s0 = starts[segStart], s1 = starts[b + 1];
                const int st = q[b];
                if (rec_marginals) {
                    diff[(size_t)st * (T + 1) + s0] += 1;
                    diff[(size_t)st * (T + 1) + s1] -= 1;
                    boundary[s0] = 1;
                    if (st > max_state_recorded) max_state_recorded = st;
                }
                if (rec_sequences) {
                    snprintf(buf, sizeof buf, "%s%zu:%d", firstSeg ? "" : "\t", s1 - s0, st);
                    out_sequences += buf;
                }
                if (rec_segments && rec_marginals) { run_start.push_back(s0); run_state.push_back(st); }
                firstSeg = false;
                segStart = b + 1;
            }
        }
        if (rec_segments) segments_line(run_start, run_state);
        if (rec_blocks) out_blocks += "\n";
        if (rec_sequences) out_sequences += "\n";
        if (rec_compression) {
            snprintf(buf, sizeof buf, "%g\n", ((double)T) / ((double)B));
            out_compression += buf;
        }
        n_recorded++;
    }
    // The `segments` line of a recorded sweep (Records.hpp:208-209): StateMarginals::nrSegments() and internalSize()
    // (StateMarginals.hpp:194-206) at the moment Records::record writes them - BEFORE the sweep's last run of equal states
    // goes into addRecord.  StateMarginals keeps one record per marginal segment in a rotating queue: the counts of the
    // states in ascending order, a state with a zero count left out, an index entry in front of every stored state that
    // does not follow the one stored before it (state 0 never needs one), one terminator (StateMarginals.hpp:71-115).
    // A record's length therefore depends only on the SET of states with a count.  At the time of writing the records of
    // the segments before the last run's start already hold this sweep's state, the others do not yet; the last run
    // ends at T and absorbs whole records (no split, StateMarginals.hpp:117-131), so the number of segments is final.
    static uint64_t record_length(uint64_t states) {
        return (uint64_t)__builtin_popcountll(states) + (uint64_t)__builtin_popcountll(states & ~(states << 1) & ~1ull) + 1;
    }
    void segments_line(const std::vector<uint64_t>& run_start, const std::vector<int>& run_state) {
        char buf[64];
        if (!rec_marginals) {   // addRecord is never called (Records.hpp:176,212): the queue stays {0}
            out_segments += "1\t1\n";
            return;
        }
        const size_t R = run_start.size(), M = mseg_start.size();
        const uint64_t last_run = run_start[R - 1];
        std::vector<uint64_t> ns, nm;
        ns.reserve(M + R); nm.reserve(M + R);
        uint64_t internal = 0;
        size_t i = 0, j = 0;   // run i and old segment j contain the position p
        uint64_t p = 0;
        while (p < T) {
            const uint64_t before = mseg_states[j], after = before | (1ull << run_state[i]);
            ns.push_back(p); nm.push_back(after);
            internal += record_length(p < last_run ? after : before);
            const uint64_t e_run = i + 1 < R ? run_start[i + 1] : T, e_seg = j + 1 < M ? mseg_start[j + 1] : T;
            p = std::min(e_run, e_seg);
            if (p == e_run) ++i;
            if (p == e_seg) ++j;
        }
        mseg_start.swap(ns); mseg_states.swap(nm);
        snprintf(buf, sizeof buf, "%zu\t%llu\n", mseg_start.size(), (unsigned long long)internal);
        out_segments += buf;
    }
    void append_params() {
        // Theta::str -> concat(Observation<NormalParam>::str) (Theta.hpp:215-219, Observation.hpp:205-210)
        for (int k = 0; k < nP(); ++k) {
            if (k) out_params += "\t";
            out_params += std::to_string(mu[k]) + "\t" + std::to_string(var[k]);
        }
        out_params += "\n";
    }

    // StateMarginals::save (StateMarginals.hpp:268-310)
    std::string marginals_text() const {
        std::string out;
        char buf[32];
        if (diff.empty()) {  // nothing recorded: one segment, no counts
            snprintf(buf, sizeof buf, "%zu\n", T);
            return std::string(buf);
        }
        const int K = cfg.K;
        const int nst = max_state_recorded + 1;
        std::vector<int32_t> cur(K, 0);
        size_t segStart = 0;
        for (size_t t = 0; t <= T; ++t) {
            const bool cut = (t == T) || (t > 0 && boundary[t]);
            if (cut) {
                snprintf(buf, sizeof buf, "%zu", t - segStart); out += buf;
                for (int s = 0; s < nst; ++s) { snprintf(buf, sizeof buf, "\t%d", cur[s]); out += buf; }
                out += "\n";
                segStart = t;
            }
            if (t == T) break;
            if (t == 0 || boundary[t])
                for (int s = 0; s < K; ++s) cur[s] += diff[(size_t)s * (T + 1) + t];
        }
        return out;
    }

    // dense marginal counts [K][T] (for parity with the device's dense export)
    void marginals_dense(std::vector<int32_t>& out) const {
        const int K = cfg.K;
        out.assign((size_t)K * T, 0);
        if (diff.empty()) return;
        for (int s = 0; s < K; ++s) {
            int32_t c = 0;
            for (size_t t = 0; t < T; ++t) { c += diff[(size_t)s * (T + 1) + t]; out[(size_t)s * T + t] = c; }
        }
    }
};

}  // namespace hml_oracle

#endif
