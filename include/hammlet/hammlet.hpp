// hammlet.hpp - the reference's in-process C++ surface (Emissions / Blocks / Statistics / Trellis /
// StateSequence / Theta / Transitions / Initial / Records / sampleHMM) as thin host classes over the
// C ABI of libhammlet_hip.so (include/hml.h).  All data and all arithmetic of the Gibbs sweep live on
// the GPU; these classes only hold the handle, forward the calls and fetch results.
//
// A driver written in the shape of the reference's src/main.cpp compiles against these names:
//     rng_t RNG(seed);                               // src/main.cpp:107-108
//     MaxletTransform(fin, inputValues, stats, nrDataDim);   HaarBreakpointWeights(inputValues);
//     Statistics<IntegralArray, Normal> ia(stats, nrDataDim);  Blocks<BreakpointArray> blocks(inputValues);
//     Emissions<S, B> y(ia, blocks);   autoPrior(var, p, y, stdEstimate);   Theta<NormalInverseGamma> theta(...);
//     StateSequence<ForwardBackward> q(RNG);  sampleHMM(y, q, theta, tau_theta, A, tau_A, pi, tau_pi, mapping, ...);
// Differences that follow from the device residency are noted at each class.
#ifndef HAMMLET_HPP
#define HAMMLET_HPP

#include <cmath>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <istream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../hml.h"

namespace hammlet {

typedef float real_t;          // reference src/includes.hpp:10
typedef int16_t marginal_t;    // reference src/includes.hpp:13

// tag classes (reference src/Tags.hpp)
class Normal {};
class NormalParam {};
typedef NormalParam NormalInverseGamma;
class NormalInverseGammaParam {};
class CategoricalParam {};
typedef CategoricalParam Dirichlet;
class DirichletParam {};
class CategoricalParamVector {};
typedef CategoricalParamVector DirichletVector;
class DirichletParamVector {};
class ForwardBackward {};
class Mixture {};
class IntegralArray {};
class BreakpointArray {};
enum MappingType { combinations, independent };

inline void hml_check(int rc) {
    if (rc != 0) throw std::runtime_error(hml_last_error());
}

// The reference hands one `rng_t& RNG` to every sampling object.  Here that object is the chain's device
// context: it owns the Philox key (seed, chain) and every device buffer.
//
// It also carries what lets a driver in the reference's shape run unedited although the device fixes the whole model in
// one call (hml_set_model) and draws theta, pi and A together:
//  * the pieces of the model arrive one by one - Theta's constructor knows the emission prior, Initial::sample(tau_pi)
//    and Transitions::sample(tau_A) (or sampleHMM / StateSequence::sample) the Dirichlet priors - and the device model
//    is created when the last one is known; prior draws requested before that are replayed in order;
//  * `theta.sample(tau_theta); pi.sample(tau_pi); A.sample(tau_A);` (src/main.cpp:397-399, src/HMM.hpp:112-116) is ONE
//    device draw: the first call that is not yet covered by a device draw triggers it, the other two are covered by it.
class rng_t {
    hml_ctx* mCtx = nullptr;
    static rng_t*& currentSlot() { static thread_local rng_t* cur = nullptr; return cur; }   // per host thread: one chain each

    // deferred model
    std::vector<real_t> mNig;
    size_t mStates = 0;
    bool mHaveTheta = false, mHaveA = false, mHavePi = false, mModelSet = false;
    real_t mOff = 0, mDiag = 0, mAlpha = 0;
    int mPendingPriorDraws = 0;
    int mSelfTrans = -1;       // what the device model currently uses (-1: the default of hml_set_model, on)
    bool mDynamic = true;      // block structure recomputed in every sweep (the device's mode)
    unsigned mCovered = 0;     // bit 0 theta, 1 pi, 2 A: sample() calls covered by the latest device draw

public:
    enum { THETA = 1, PI = 2, TRANS = 4 };
    explicit rng_t(uint64_t seed, int device = 0, uint32_t chain = 0) {
        hml_check(hml_create(&mCtx, device, seed, chain, nullptr));
        currentSlot() = this;
    }
    ~rng_t() {
        if (currentSlot() == this) currentSlot() = nullptr;
        hml_destroy(mCtx);
    }
    rng_t(const rng_t&) = delete;
    hml_ctx* ctx() const { return mCtx; }
    // objects that the reference constructs without an RNG argument (Statistics, Blocks) attach to this
    static rng_t& current() {
        if (!currentSlot()) throw std::runtime_error("No device context: construct rng_t RNG(seed) first!");
        return *currentSlot();
    }
    static bool hasCurrent() { return currentSlot() != nullptr; }

    // ---- model assembly
    void offerTheta(size_t nrStates, const std::vector<real_t>& nig) { mStates = nrStates; mNig = nig; mHaveTheta = true; }
    void offerTransitions(real_t off, real_t diag) { if (!mModelSet) { mOff = off; mDiag = diag; mHaveA = true; tryCreate(); } }
    void offerInitial(real_t alpha) { if (!mModelSet) { mAlpha = alpha; mHavePi = true; tryCreate(); } }
    void modelCreatedDirectly() { mModelSet = true; }
    bool modelSet() const { return mModelSet; }
    void tryCreate() {
        if (mModelSet || !(mHaveTheta && mHaveA && mHavePi)) return;
        hml_check(hml_set_model(mCtx, (int)mStates, mNig.data(), mOff, mDiag, mAlpha, 1));
        mModelSet = true;
        for (; mPendingPriorDraws > 0; --mPendingPriorDraws) hml_check(hml_sample_prior(mCtx));
    }
    void requireModel() {
        tryCreate();
        if (!mModelSet) throw std::runtime_error("The model is incomplete: emission, transition and initial priors are needed before sampling!");
    }
    void setSelfTransitions(bool on) {
        requireModel();
        if (mSelfTrans != (on ? 1 : 0)) { hml_check(hml_set_self_transitions(mCtx, on ? 1 : 0)); mSelfTrans = on ? 1 : 0; }
    }
    // `dynamic` of sampleHMM (src/HMM.hpp:74,99-102): the reference's driver only flips a bool on "D" and passes it on
    void blocksFixed() { mDynamic = false; }
    void setDynamic(bool on) {
        requireModel();
        if (on != mDynamic) { hml_check(hml_set_dynamic(mCtx, on ? 1 : 0)); mDynamic = on; }
    }
    // ---- draws
    void deviceDrewAll() { mCovered = THETA | PI | TRANS; }        // a sweep (or a prior draw) has drawn theta, pi and A
    void sampleCalled(unsigned who) {
        if (mCovered & who) { mCovered &= ~who; return; }
        // not covered: a draw from the (reset) priors, src/main.cpp:393-401
        if (mModelSet) hml_check(hml_sample_prior(mCtx));
        else ++mPendingPriorDraws;
        mCovered = (THETA | PI | TRANS) & ~who;
    }
};

template <typename T>
class SufficientStatistics;
template <>
class SufficientStatistics<Normal> {   // reference src/SufficientStatistics.hpp:49-142
    real_t mSum = 0, mSumSq = 0;

public:
    SufficientStatistics() {}
    SufficientStatistics(real_t v) : mSum(v), mSumSq(v * v) {}
    SufficientStatistics(real_t s, real_t q) : mSum(s), mSumSq(q) {}
    real_t sum() const { return mSum; }
    real_t sumSq() const { return mSumSq; }
    size_t nrDim() const { return 1; }
};

// GPU index used by objects that exist before the chain's context does (the text reader); set by the driver.
inline int& inputDevice() { static thread_local int dev = 0; return dev; }

// readValues: every value that `while ( input >> v )` extracts (reference src/wavelet.hpp:131), converted on the GPU
// chunk by chunk (hml_text_*; tokens the device cannot decide with proof go through the stream extraction on the host,
// so the values are the reference's bit for bit), appended to `values`.
inline void readValues(std::istream& input, std::vector<real_t>& values, const size_t nrDim = 1, const size_t reserveT = 0) {
    if (nrDim <= 0) throw std::runtime_error("Number of dimensions must be positive!");
    if (!input) throw std::runtime_error("Cannot read input file or stream!");
    hml_text* reader = nullptr;
    // staging buffers no larger than the input when its size is known (reserveT ~ bytes / 2): pinned memory is costly
    // to allocate, and a 10^5-value file should not pay for two 64 MiB buffers
    uint64_t chunk = 0;
    if (reserveT) {
        chunk = 2 * (uint64_t)reserveT + 4096;
        chunk = (chunk + 4095) / 4096 * 4096;
        if (chunk < (1u << 16)) chunk = 1u << 16;
        if (chunk > (64u << 20)) chunk = 0;   // the default
    }
    hml_check(hml_text_open(&reader, inputDevice(), chunk));
    struct Closer { hml_text* r; ~Closer() { hml_text_close(r); } } closer{reader};
    if (reserveT) hml_check(hml_text_reserve(reader, reserveT));
    for (;;) {
        char* buf = nullptr;
        uint64_t cap = 0;
        hml_check(hml_text_buffer(reader, &buf, &cap));
        input.read(buf, (std::streamsize)cap);
        const std::streamsize got = input.gcount();
        if (got <= 0) break;
        hml_check(hml_text_commit(reader, (uint64_t)got));
    }
    uint64_t n = 0;
    int stopped = 0;
    hml_check(hml_text_finish(reader, &n, &stopped));
    const size_t at = values.size();
    values.resize(at + n);
    hml_check(hml_text_values(reader, values.data() + at));
}

// MaxletTransform (reference src/wavelet.hpp:97-188), the reference's contract: reads the stream, leaves the maxlet
// coefficients (one per position) in `coeffs` and the per-position sufficient statistics (x, x*x) in `suffstats`.
// The observations go to the device here (K1-K3 build coefficients, weights and integral arrays at once) and the
// coefficients come back, so that a driver in the reference's shape can estimate the noise from them on the host
// (src/main.cpp:303-311).  The `hammlet` driver itself uses readValues + Statistics(values, nrDim) and never moves
// coefficients or weights to the host.
template <typename T>
void MaxletTransform(std::istream& input, std::vector<real_t>& coeffs, std::vector<SufficientStatistics<T>>& suffstats,
                     const size_t nrDim = 1, const size_t reserveT = 0) {
    if (!coeffs.empty()) throw std::runtime_error("Coefficient array must be empty!");
    if (!suffstats.empty()) throw std::runtime_error("Sufficient statistics array must be empty!");
    std::vector<real_t> values;
    readValues(input, values, nrDim, reserveT);
    if (values.size() % nrDim != 0)
        throw std::runtime_error("Input stream did not contain enough values to fill all dimensions at last position!");
    if (values.empty()) return;
    hml_ctx* ctx = rng_t::current().ctx();
    if (nrDim > 1) hml_check(hml_set_dimensions(ctx, (int)nrDim, 0));
    hml_check(hml_load_observations(ctx, values.data(), values.size()));
    coeffs.resize(values.size() / nrDim);
    hml_check(hml_get_coefficients(ctx, coeffs.data()));
    suffstats.reserve(values.size() + 1);
    for (const real_t v : values) suffstats.push_back(SufficientStatistics<T>(v));
}

// HaarBreakpointWeights (reference src/wavelet.hpp:68-93): the device computed the weights together with the
// transform; a vector of matching size (the maxlet coefficients MaxletTransform returned) receives them, so the host
// may still change them before the Blocks constructor takes them back (src/main.cpp:332-334).
inline void HaarBreakpointWeights(std::vector<real_t>& weights) {
    if (weights.empty()) throw std::runtime_error("Cannot compute Haar breakpoint weights, vector is empty!");
    if (!rng_t::hasCurrent()) return;
    hml_ctx* ctx = rng_t::current().ctx();
    double sigma = 0;
    if (hml_noise_sigma(ctx, &sigma) != 0) return;   // nothing loaded yet: the weights are built when the statistics are
    hml_check(hml_get_weights(ctx, weights.data()));
}

template <typename A, typename B>
class Statistics;
template <typename B>
class Blocks;
template <typename S, typename B>
class Emissions;
template <typename P>
class Theta;

// Statistics<IntegralArray, Normal> (reference src/Statistics/IntegralArray.hpp:28-233).  The constructor
// uploads the observations: K1-K3 build maxlet coefficients, breakpoint weights and the integral array.
template <>
class Statistics<IntegralArray, Normal> {
    rng_t& mDev;
    size_t mSize = 0, mNrDim = 1;
    SufficientStatistics<Normal> mCurrent;
    friend class Emissions<Statistics<IntegralArray, Normal>, Blocks<BreakpointArray>>;

public:
    Statistics(const Statistics&) = delete;
    // `values`: the vector MaxletTransform filled (swap-stolen like the reference's constructors)
    // (with nrDim > 1 the values of a position's dimensions follow each other, reference src/wavelet.hpp:131-137)
    Statistics(std::vector<real_t>& values, const size_t nrDim) : mDev(rng_t::current()), mNrDim(nrDim) {
        if (values.empty()) throw std::runtime_error("Input vector for breakpoint weights is empty!");
        if (nrDim > 1) hml_check(hml_set_dimensions(mDev.ctx(), (int)nrDim, 0));
        mSize = values.size() / nrDim;
        hml_check(hml_load_observations(mDev.ctx(), values.data(), values.size()));
        std::vector<real_t>().swap(values);
    }
    // several chains over the same observations (`hammlet -chains N`): the vector stays with the caller
    enum KeepInput { keepInput };
    Statistics(const std::vector<real_t>& values, const size_t nrDim, KeepInput) : mDev(rng_t::current()), mNrDim(nrDim) {
        if (values.empty()) throw std::runtime_error("Input vector for breakpoint weights is empty!");
        if (nrDim > 1) hml_check(hml_set_dimensions(mDev.ctx(), (int)nrDim, 0));
        mSize = values.size() / nrDim;
        hml_check(hml_load_observations(mDev.ctx(), values.data(), values.size()));
    }
    // a further chain on a GPU that already holds the construction of these observations (`hammlet -chains N` with more chains
    // than GPUs): shares it instead of building a copy (hml_attach_observations; include/hml.h)
    enum AttachInput { attachInput };
    Statistics(hml_ctx* source, const size_t size, const size_t nrDim, AttachInput) : mDev(rng_t::current()), mSize(size), mNrDim(nrDim) {
        hml_check(hml_attach_observations(mDev.ctx(), source));
    }
    // the reference's shape (src/main.cpp:340, src/Statistics/IntegralArray.hpp:136-191): the per-position statistics
    // MaxletTransform produced (swap-stolen).  The device already holds the integral arrays when MaxletTransform put
    // the observations there; statistics from elsewhere are uploaded now (x = sum of each one-observation statistic).
    Statistics(std::vector<SufficientStatistics<Normal>>& stats, const size_t nrDim) : mDev(rng_t::current()), mNrDim(nrDim) {
        if (stats.empty()) throw std::runtime_error("Input vector for breakpoint weights is empty!");
        if (stats.size() % nrDim != 0) throw std::runtime_error("Input stream did not contain enough values to fill all dimensions at last position!");
        mSize = stats.size() / nrDim;
        double sigma = 0;
        if (hml_noise_sigma(mDev.ctx(), &sigma) != 0) {
            std::vector<real_t> values(stats.size());
            for (size_t i = 0; i < stats.size(); ++i) values[i] = stats[i].sum();
            if (nrDim > 1) hml_check(hml_set_dimensions(mDev.ctx(), (int)nrDim, 0));
            hml_check(hml_load_observations(mDev.ctx(), values.data(), values.size()));
        }
        std::vector<SufficientStatistics<Normal>>().swap(stats);
    }
    size_t nrDim() const { return mNrDim; }
    size_t size() const { return mSize; }
    const SufficientStatistics<Normal>& suffStat(size_t) const { return mCurrent; }
    double noiseEstimate() const { double s; hml_check(hml_noise_sigma(mDev.ctx(), &s)); return s; }   // main.cpp:303-311
    rng_t& device() const { return mDev; }
};

// Blocks<BreakpointArray> (reference src/Blocks/BreakpointArray.hpp:24-307): the device holds the weights;
// an enumeration fetches the block starts of the current threshold once and iterates over them.
template <>
class Blocks<BreakpointArray> {
    rng_t& mDev;
    size_t mSize;
    std::vector<uint32_t> mStarts;
    std::vector<float> mSum, mSumSq;
    size_t mPos = 0;
    bool mIterating = false, mFinished = false;
    friend class Emissions<Statistics<IntegralArray, Normal>, Blocks<BreakpointArray>>;

public:
    explicit Blocks(const Statistics<IntegralArray, Normal>& stats) : mDev(stats.device()), mSize(stats.size()) {}
    // the reference's shape (src/main.cpp:341, src/Blocks/BreakpointArray.hpp:130-184): the breakpoint weights as the
    // host holds them (swap-stolen) - whatever the driver did to them since HaarBreakpointWeights is what counts
    explicit Blocks(std::vector<real_t>& weights) : mDev(rng_t::current()), mSize(weights.size()) {
        if (weights.empty()) throw std::runtime_error("Input vector for breakpoint weights is empty!");
        hml_check(hml_set_weights(mDev.ctx(), weights.data(), weights.size()));
        std::vector<real_t>().swap(weights);
    }
    void scaleWeights(real_t m) { hml_check(hml_scale_weights(mDev.ctx(), m)); }   // main.cpp:332-334
    void createBlocks(real_t threshold) {
        hml_check(hml_create_blocks(mDev.ctx(), threshold));
        fetch();
    }
    void fetch() {
        uint64_t B = 0;
        hml_check(hml_get_num_blocks(mDev.ctx(), &B));
        mStarts.resize(B + 1); mSum.resize(B); mSumSq.resize(B);
        hml_check(hml_get_blocks(mDev.ctx(), mStarts.data()));
        hml_check(hml_get_block_stats(mDev.ctx(), mSum.data(), mSumSq.data()));
        mFinished = false;
    }
    void initForward() { mPos = 0; mIterating = true; mFinished = false; }
    bool next() {
        if (mPos + 1 >= mStarts.size()) { mIterating = false; mFinished = true; return false; }
        ++mPos;
        return true;
    }
    size_t start() const { return mStarts[mPos - 1]; }
    size_t end() const { return mStarts[mPos]; }
    size_t blockSize() const { return end() - start(); }
    size_t size() const { return mSize; }
    size_t nrBlocks() const {
        if (mIterating) throw std::runtime_error("Cannot determine size of block structure before all blocks have been seen!");
        return mStarts.empty() ? 0 : mStarts.size() - 1;
    }
};

// Emissions<Statistics<S,T>, Blocks<B>> (reference src/Emissions.hpp:12-112)
template <>
class Emissions<Statistics<IntegralArray, Normal>, Blocks<BreakpointArray>> {
    typedef Statistics<IntegralArray, Normal> S;
    typedef Blocks<BreakpointArray> B;
    S& mStats;
    B& mBlocks;

public:
    Emissions(S& stats, B& blocks) : mStats(stats), mBlocks(blocks) {
        if (mStats.size() != mBlocks.size())
            throw std::runtime_error("Block structure and statistics have different number of data points!");
    }
    S& stats() { return mStats; }
    B& blocks() { return mBlocks; }
    const S& stats() const { return mStats; }
    const B& blocks() const { return mBlocks; }
    void createBlocks(real_t thresh) { mBlocks.createBlocks(thresh); }
    template <typename P>
    void createBlocks(const Theta<P>& theta);   // "S" token: fix the structure at the current theta
    size_t nrBlocks() const { return mBlocks.nrBlocks(); }
    size_t nrDim() const { return mStats.nrDim(); }   // reference src/Emissions.hpp:60-62
    size_t start() const { return mBlocks.start(); }
    size_t end() const { return mBlocks.end(); }
    size_t blockSize() const { return mBlocks.blockSize(); }
    size_t size() const { return mBlocks.size(); }
    void initForward() { mBlocks.initForward(); }
    bool next() {
        if (!mBlocks.next()) return false;
        mStats.mCurrent = SufficientStatistics<Normal>(mBlocks.mSum[mBlocks.mPos - 1], mBlocks.mSumSq[mBlocks.mPos - 1]);
        return true;
    }
    const SufficientStatistics<Normal>& suffStat(size_t dim) const { return mStats.suffStat(dim); }
    hml_ctx* ctx() const { return mStats.device().ctx(); }
};

// autoPrior (reference src/AutoPriors.hpp:86-110): {alpha, beta, mu0, nu}
template <typename E>
std::vector<real_t> autoPrior(real_t s2, real_t p, E& y, const double /*noiseStdev: held by the device*/) {
    std::vector<real_t> out(4);
    hml_check(hml_autoprior(y.ctx(), s2, p, out.data()));
    return out;
}

class Mapping {   // reference src/Mapping.hpp:53-137, "combinations": nrParams^nrDataDim states
    size_t mParams, mDims, mStates;

public:
    Mapping(size_t nrDataDim, size_t nrParams, MappingType) : mParams(nrParams), mDims(nrDataDim), mStates(1) {
        if (nrDataDim <= 0) throw std::runtime_error("Number of data dimensions must be positive!");
        if (nrParams <= 0) throw std::runtime_error("Number of parameters must be positive!");
        for (size_t d = 0; d < nrDataDim; ++d) mStates *= nrParams;
        if (mStates <= 1) throw std::runtime_error("Requested parameters would yield an HMM with less than 2 states!");
    }
    size_t nrStates() const { return mStates; }
    size_t nrParams() const { return mParams; }
    size_t nrDataDims() const { return mDims; }
    size_t operator()(size_t state, size_t dim) const {   // parameter of `state` for data dimension `dim`
        for (size_t d = 0; d < dim; ++d) state /= mParams;
        return state % mParams;
    }
};

// Hyper-parameter holders (reference src/ThetaHyperParam.hpp, TransitionHyperParam.hpp, InitialHyperParam.hpp):
// plain values; the posteriors themselves live on the device.
template <typename T>
class ThetaHyperParam {
    std::vector<std::vector<real_t>> mP;

public:
    explicit ThetaHyperParam(const std::vector<std::vector<real_t>>& hp) : mP(hp) {
        if (mP.empty()) throw std::runtime_error("Number of hyperparameters must be positive");
    }
    size_t nrParams() const { return mP.size(); }
    const std::vector<real_t>& prior(size_t d) const { return mP[d]; }
};
template <typename T>
class TransitionHyperParam {
public:
    size_t nrStates; real_t off, diag;
    TransitionHyperParam(size_t n, real_t o, real_t d) : nrStates(n), off(o), diag(d) {}
};
template <typename T>
class InitialHyperParam {
public:
    size_t nrStates; real_t alpha;
    InitialHyperParam(size_t n, real_t a) : nrStates(n), alpha(a) {}
};

template <typename T>
class Transitions {   // reference src/Transitions.hpp:23-90 (values are fetched from the device on demand)
    rng_t& mDev; size_t mK;
public:
    Transitions(size_t nrStates, rng_t& RNG) : mDev(RNG), mK(nrStates) {}
    size_t nrStates() const { return mK; }
    // Transitions::sample (reference src/Transitions.hpp:75-79): part of the device's joint draw (see rng_t)
    template <typename H>
    void sample(const TransitionHyperParam<H>& tau_A) { mDev.offerTransitions(tau_A.off, tau_A.diag); mDev.sampleCalled(rng_t::TRANS); }
    real_t operator()(size_t from, size_t to) const {
        std::vector<real_t> A(mK * mK), pi(mK);
        hml_check(hml_get_transitions(mDev.ctx(), A.data(), pi.data()));
        return A[from * mK + to];
    }
};
template <typename T>
class Initial {       // reference src/Initial.hpp:13-58
    rng_t& mDev; size_t mK;
public:
    Initial(size_t nrStates, rng_t& RNG) : mDev(RNG), mK(nrStates) {}
    size_t nrStates() const { return mK; }
    // Initial::sample (reference src/Initial.hpp:34-40): part of the device's joint draw (see rng_t)
    template <typename H>
    void sample(const InitialHyperParam<H>& tau_pi) { mDev.offerInitial(tau_pi.alpha); mDev.sampleCalled(rng_t::PI); }
    std::vector<real_t> valueVector() const {
        std::vector<real_t> A(mK * mK), pi(mK);
        hml_check(hml_get_transitions(mDev.ctx(), A.data(), pi.data()));
        return pi;
    }
};

// Theta<NormalInverseGamma> (reference src/Theta.hpp): constructing it fixes the model on the device and,
// like the reference's constructor (Theta.hpp:126-127), draws once from the prior.
template <>
class Theta<NormalParam> {
    rng_t& mDev; size_t mK, mP;   // states, emission parameters (mP^D = mK)
public:
    Theta(const Theta&) = delete;
    template <typename H, typename TA, typename TP>
    Theta(ThetaHyperParam<H>& tau_theta, const TransitionHyperParam<TA>& tau_A, const InitialHyperParam<TP>& tau_pi,
          bool useSelfTransitions, rng_t& RNG)
        : mDev(RNG), mK(tau_A.nrStates), mP(tau_theta.nrParams()) {
        hml_check(hml_set_model(mDev.ctx(), (int)mK, tau_theta.prior(0).data(), tau_A.off, tau_A.diag, tau_pi.alpha,
                                useSelfTransitions ? 1 : 0));
        mDev.modelCreatedDirectly();
    }
    // the reference's shapes (src/Theta.hpp:29-44, src/main.cpp:357-362): the emission prior and the mapping; the
    // device model is completed when the Dirichlet priors are known (see rng_t).  The constructor's own draw from the
    // prior (Theta.hpp:126-127) is the one hml_set_model makes.
    template <typename H>
    Theta(ThetaHyperParam<H>& tau_theta, const size_t nrDataDim, const MappingType mappingType, rng_t& RNG)
        : mDev(RNG), mK(Mapping(nrDataDim, tau_theta.nrParams(), mappingType).nrStates()), mP(tau_theta.nrParams()) {
        mDev.offerTheta(mK, tau_theta.prior(0));
    }
    template <typename H>
    Theta(ThetaHyperParam<H>& tau_theta, const size_t /*nrDataDim*/, const Mapping mapping, rng_t& RNG)
        : mDev(RNG), mK(mapping.nrStates()), mP(tau_theta.nrParams()) {
        mDev.offerTheta(mK, tau_theta.prior(0));
    }
    // Theta::sample (reference src/Theta.hpp:203-211): part of the device's joint draw (see rng_t)
    template <typename H>
    void sample(ThetaHyperParam<H>&) { mDev.sampleCalled(rng_t::THETA); }
    rng_t& device() const { return mDev; }
    size_t nrParams() const { return mP; }
    size_t nrStates() const { return mK; }
    std::vector<real_t> meanVar() const {
        std::vector<real_t> v(2 * mP);
        hml_check(hml_get_theta(mDev.ctx(), v.data()));
        return v;
    }
    // Theta::str(): to_string(mean) \t to_string(var) per state, tab-joined (Theta.hpp:215-219, Observation.hpp:205-210)
    std::string str(const std::string& sep = "\t") const {
        const std::vector<real_t> v = meanVar();
        std::string s;
        for (size_t k = 0; k < mP; ++k) {
            if (k) s += sep;
            s += std::to_string(v[2 * k]) + "\t" + std::to_string(v[2 * k + 1]);
        }
        return s;
    }
    real_t thresholdValue() const {   // Theta.hpp:227-234: smallest variance
        const std::vector<real_t> v = meanVar();
        real_t r = std::numeric_limits<real_t>::infinity();
        for (size_t k = 0; k < mP; ++k) r = std::min(r, v[2 * k + 1]);
        return r;
    }
    hml_ctx* ctx() const { return mDev.ctx(); }
};

template <typename P>
void Emissions<Statistics<IntegralArray, Normal>, Blocks<BreakpointArray>>::createBlocks(const Theta<P>& theta) {
    theta.device().requireModel();
    hml_check(hml_set_static_blocks(theta.ctx()));
    theta.device().blocksFixed();
}

// Trellis (reference src/Trellis.hpp:8-76).  In a sweep the rows are filled and sampled on the device; fetch() copies
// the last sweep's normalised forward rows (row 0 = pi) into this host container.  The container itself keeps the
// reference's interface - rows can be appended and a row can be sampled (std::discrete_distribution semantics, the
// uniform from the chain's Philox key) - for code that uses it on its own.
class Trellis {
    rng_t& mDev;
    std::vector<real_t> mVec;
    size_t mNrStates = 2;
    void assertRange(size_t d) const {
        if (d >= mNrStates) throw std::runtime_error("Trellis dimension index out of bounds!");
    }

public:
    Trellis(const Trellis&) = delete;
    explicit Trellis(rng_t& RNG) : mDev(RNG) {}
    Trellis(size_t nrStates, rng_t& RNG) : mDev(RNG), mNrStates(nrStates) {}
    void setNrStates(size_t K) { mNrStates = K; }
    void fetch() {
        uint64_t B = 0;
        hml_check(hml_get_num_blocks(mDev.ctx(), &B));
        mVec.resize((B + 1) * mNrStates);
        hml_check(hml_get_forward_rows(mDev.ctx(), mVec.data()));
    }
    real_t& operator()(size_t t, size_t d) { assertRange(d); return mVec[t * mNrStates + d]; }
    real_t operator()(size_t t, size_t d) const { assertRange(d); return mVec[t * mNrStates + d]; }
    real_t& back(size_t d) { return mVec[mVec.size() - mNrStates + d]; }
    real_t back(size_t d) const { return mVec[mVec.size() - mNrStates + d]; }
    size_t size() const {
        if (mNrStates == 0) throw std::runtime_error("Division by zero!");
        return mVec.size() / mNrStates;
    }
    void push_back(const std::vector<real_t>& vec) { mVec.insert(mVec.end(), vec.begin(), vec.end()); }
    size_t sample(size_t t) const {   // Trellis.hpp:61-66
        if ((t + 1) * mNrStates > mVec.size()) throw std::runtime_error("Trellis row index out of bounds!");
        uint32_t idx = 0;
        hml_check(hml_categorical_draw(mDev.ctx(), mVec.data() + t * mNrStates, (int)mNrStates, &idx));
        return idx;
    }
    void reserve(size_t N) { mVec.reserve(N * mNrStates); }
    void clear() { mVec.clear(); }
};

class Records;

// StateSequence<Tag> (reference src/StateSequence.hpp:18-106): sample() runs ONE Gibbs sweep of the given
// kind on the device (state sequence + the conjugate parameter draws that follow it in sampleHMM).
template <typename Tag>
class StateSequence {
    rng_t& mDev;
    std::vector<marginal_t> mStates;
    static char method();

public:
    StateSequence(const StateSequence&) = delete;
    explicit StateSequence(rng_t& RNG) : mDev(RNG) {}
    hml_ctx* ctx() const { return mDev.ctx(); }
    // StateSequence<Tag>::sample (reference src/StateSequence.hpp:44-65, StateSequence/ForwardBackward.hpp:16-32,
    // StateSequence/Mixture.hpp:31-50): one Gibbs sweep.  On the device the sweep ends with the conjugate draws of theta,
    // pi and A, so the `theta.sample(tau_theta); pi.sample(tau_pi); A.sample(tau_A);` that follow it in a loop written
    // like sampleHMM (src/HMM.hpp:111-116) are covered by this call.  With doRecord the sweep enters the marginals and
    // `records` appends the state-sequence, block and compression lines; `records.record(theta)` then adds the
    // parameters (HMM.hpp:118-120).
    template <typename E, typename ThetaType, typename TauThetaType, typename TransitionsType, typename TauAType,
              typename InitialType, typename TauPiType>
    void sample(E& y, const ThetaType& theta, TauThetaType& tau_theta, const TransitionsType& A, TauAType& tau_A,
                const InitialType& pi, TauPiType& tau_pi, const Mapping& mapping, Records& records, const bool doRecord,
                const bool useSelfTransitions);
    void fetch() {
        uint64_t B = 0;
        hml_check(hml_get_num_blocks(mDev.ctx(), &B));
        mStates.resize(B);
        hml_check(hml_get_states(mDev.ctx(), mStates.data()));
    }
    size_t size() const { return mStates.size(); }
    const std::vector<marginal_t>& states() const { return mStates; }
    marginal_t operator[](size_t s) const {
        if (s >= mStates.size()) throw std::runtime_error("State sequence index " + std::to_string(s) + " out of bounds!");
        return mStates[s];
    }
    std::string str() const {
        std::string s;
        for (size_t i = 0; i < mStates.size(); ++i) { if (i) s += " "; s += std::to_string(mStates[i]); }
        return s;
    }
    void clear() { std::vector<marginal_t>().swap(mStates); }
    static char methodChar() { return method(); }
};
template <> inline char StateSequence<ForwardBackward>::method() { return HML_METHOD_FB; }
template <> inline char StateSequence<Mixture>::method() { return HML_METHOD_MIXTURE; }

}  // namespace hammlet

#include "Records.hpp"

namespace hammlet {

template <typename Tag>
template <typename E, typename ThetaType, typename TauThetaType, typename TransitionsType, typename TauAType,
          typename InitialType, typename TauPiType>
void StateSequence<Tag>::sample(E& y, const ThetaType&, TauThetaType&, const TransitionsType&, TauAType& tau_A, const InitialType&,
                                TauPiType& tau_pi, const Mapping&, Records& records, const bool doRecord,
                                const bool useSelfTransitions) {
    hml_ctx* ctx = y.ctx();
    mDev.offerTransitions(tau_A.off, tau_A.diag);
    mDev.offerInitial(tau_pi.alpha);
    mDev.setSelfTransitions(useSelfTransitions);
    records.attach(ctx);
    hml_check(hml_set_recording(ctx, records.recordsMarginals() ? 1 : 0, nullptr, nullptr));
    hml_stats before{}, after{};
    hml_check(hml_get_stats(ctx, &before));
    hml_check(hml_iterate(ctx, method(), 1, doRecord ? 1 : 0));
    hml_check(hml_sync(ctx));
    hml_check(hml_get_stats(ctx, &after));
    for (uint64_t i = before.uniform_fallbacks; i < after.uniform_fallbacks; ++i)
        std::cout << "[WARNING] Uniform sampling of forward variables!" << std::endl;
    mDev.deviceDrewAll();
    fetch();
    if (doRecord) records.recordStates(ctx);
}

// sampleHMM (reference src/HMM.hpp:60-125).  The whole loop is device-resident; when per-sweep side files
// are requested the device calls back after every recorded sweep and `records` appends its lines.
template <typename Q, typename E, typename TH, typename TTH, typename TA, typename TTA, typename TP, typename TTP>
void sampleHMM(E& y, Q& q, TH& theta, TTH&, TA&, TTA& tau_A, TP&, TTP& tau_pi, const Mapping&, const size_t iterations, const size_t thinning,
               Records& records, const bool dynamic = true, const bool useSelfTransitions = true) {
    theta.device().offerTransitions(tau_A.off, tau_A.diag);
    theta.device().offerInitial(tau_pi.alpha);
    theta.device().setSelfTransitions(useSelfTransitions);
    if (thinning > iterations)
        std::cout << "[WARNING] Thinning parameter is larger than number of iterations. No data will be recorded!" << std::endl;
    hml_ctx* ctx = y.ctx();
    theta.device().requireModel();
    theta.device().setDynamic(dynamic);   // "S" fixed the structure through createBlocks(theta); "D" only flipped the driver's flag
    records.attach(ctx);
    struct Hook { Records* rec; Q* q; TH* theta; };
    Hook hook{&records, &q, &theta};
    const bool sideFiles = records.needsPerSweepData();
    hml_check(hml_set_recording(ctx, records.recordsMarginals() ? 1 : 0,
                                sideFiles ? +[](hml_ctx* c, uint64_t, void* user) {
                                    Hook* h = static_cast<Hook*>(user);
                                    h->rec->recordSweep(c, *h->theta);
                                } : (hml_record_cb) nullptr,
                                &hook));
    hml_stats before{}, after{};
    hml_check(hml_get_stats(ctx, &before));
    hml_check(hml_iterate(ctx, Q::methodChar(), iterations, thinning));
    hml_check(hml_sync(ctx));
    hml_check(hml_get_stats(ctx, &after));
    for (uint64_t i = before.uniform_fallbacks; i < after.uniform_fallbacks; ++i)
        std::cout << "[WARNING] Uniform sampling of forward variables!" << std::endl;
    hml_check(hml_set_recording(ctx, records.recordsMarginals() ? 1 : 0, nullptr, nullptr));
}

// sampleHMM for SEVERAL chains of one GPU in lockstep (`hammlet -chains N` with more chains than GPUs): every chain is
// prepared like sampleHMM prepares it, ONE hml_iterate_many runs `iterations` sweeps of all of them - chains over a shared
// construction go through one set of launches (include/hml.h) - and recorded sweeps call every chain's Records in chain
// order.  Same files per chain as sampleHMM chain by chain.  Reference: src/HMM.hpp:60-125, once per chain.
template <typename Q, typename E, typename TH, typename TTA, typename TTP>
void sampleHMMMany(const std::vector<E*>& ys, const std::vector<Q*>& qs, const std::vector<TH*>& thetas, const std::vector<TTA*>& tauAs,
                   const std::vector<TTP*>& tauPis, const size_t iterations, const size_t thinning, const std::vector<Records*>& records,
                   const bool dynamic = true, const bool useSelfTransitions = true) {
    const size_t n = ys.size();
    if (thinning > iterations)
        std::cout << "[WARNING] Thinning parameter is larger than number of iterations. No data will be recorded!" << std::endl;
    struct Hook { Records* rec; Q* q; TH* theta; };
    std::vector<Hook> hooks(n);
    std::vector<hml_ctx*> ctxs(n);
    std::vector<hml_stats> before(n);
    for (size_t k = 0; k < n; ++k) {
        TH& theta = *thetas[k];
        theta.device().offerTransitions(tauAs[k]->off, tauAs[k]->diag);
        theta.device().offerInitial(tauPis[k]->alpha);
        theta.device().setSelfTransitions(useSelfTransitions);
        theta.device().requireModel();
        theta.device().setDynamic(dynamic);
        ctxs[k] = ys[k]->ctx();
        records[k]->attach(ctxs[k]);
        hooks[k] = Hook{records[k], qs[k], thetas[k]};
        hml_check(hml_set_recording(ctxs[k], records[k]->recordsMarginals() ? 1 : 0,
                                    records[k]->needsPerSweepData() ? +[](hml_ctx* c, uint64_t, void* user) {
                                        Hook* h = static_cast<Hook*>(user);
                                        h->rec->recordSweep(c, *h->theta);
                                    } : (hml_record_cb) nullptr,
                                    &hooks[k]));
        hml_check(hml_get_stats(ctxs[k], &before[k]));
    }
    hml_check(hml_iterate_many(ctxs.data(), (int)n, Q::methodChar(), iterations, thinning));
    for (size_t k = 0; k < n; ++k) {
        hml_stats after{};
        hml_check(hml_sync(ctxs[k]));
        hml_check(hml_get_stats(ctxs[k], &after));
        for (uint64_t i = before[k].uniform_fallbacks; i < after.uniform_fallbacks; ++i)
            std::cout << "[WARNING] Uniform sampling of forward variables!" << std::endl;
        hml_check(hml_set_recording(ctxs[k], records[k]->recordsMarginals() ? 1 : 0, nullptr, nullptr));
    }
}

}  // namespace hammlet
#endif
