// hammlet - command-line driver of the MI355X-native sampler.  Same flags, same text-stream input, same
// output files and the same error format as the reference's driver (reference src/main.cpp:23-477;
// flag semantics doc/hammlet-manpage.md:33-175); everything between reading the input and writing the
// files runs on the GPU through libhammlet_hip.so.
//
// Extensions (not in the reference): `-O X` writes PREFIXmaxsegmentationSUFFIX; -raw FILE reads float32 values instead of text; -device N selects
// the GPU; -chain N selects the Philox sub-key of an independent chain; -chains N runs N independent chains (sub-keys
// chain .. chain+N-1), chain k on GPU (device + k) mod #GPUs in its own host thread, and pools their recorded marginals
// with one all-reduce over RCCL before PREFIXmarginalsSUFFIX is written (hml_allreduce_marginals); the per-sweep side
// files of chain k >= 1 are PREFIXchainK.{sequences,...}SUFFIX.
#include <condition_variable>
#include <cstdlib>
#include <ctime>
#include <exception>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hammlet/Parser.hpp"
#include "hammlet/hammlet.hpp"

using namespace hammlet;
using std::cerr;
using std::cout;
using std::endl;
using std::flush;
using std::string;
using std::vector;

static const char* kHelp =
    "hammlet (MI355X) - Bayesian HMM segmentation with dynamic Haar-wavelet compression\n\n"
    "  -f, -input-file FILE...        input files (default: standard input), whitespace-separated numbers\n"
    "  -raw FILE                      float32 input file (extension)\n"
    "  -o, -output-pattern PRE SUF    output files are PRE{marginals,...}SUF (default: hammlet- .csv)\n"
    "  -O, -output-data M S P B C G   marginals sequences parameters blocks compression segments\n"
    "                    X            maxsegmentation: the maxSegmentation tool's output for the marginals (extension)\n"
    "  -w, -overwrite                 allow overwriting output files\n"
    "  -s, -states K | C P D          number of states (default 3), or P parameters shared by P^D states over D dimensions\n"
    "  -e, -emissions normal VAR P    automatic prior: P(variance < VAR) = P (default normal 0.2 0.9)\n"
    "  -a, -auto-priors               (required) derive emission priors from the data\n"
    "  -t, -transitions OFF [DIAG]    Dirichlet prior of the transition rows (default 0.5 0.5)\n"
    "  -S, -no-self-transitions       do not model within-block self-transitions\n"
    "  -I, -initial-dist ALPHA        Dirichlet prior of the initial distribution (default 0.5)\n"
    "  -R, -random-seed N             seed (default: time)\n"
    "  -i, -iterations SCHEME         tokens: M n t | F n t | S | D | P (default M 500 0 S P F 200 0 F 300 3)\n"
    "  -m, -weight-multiplier F       multiply breakpoint weights (default 1)\n"
    "  -device N  -chain N            GPU and Philox sub-key of the chain (extensions)\n"
    "  -compat                        reference-compatible mode: the reference's mt19937 stream, libm arithmetic and\n"
    "                                 summation orders on the GPU - the same files as the reference for the same -R\n"
    "                                 (any model the default path takes: up to 64 states, -s C P D; about a hundred times\n"
    "                                 slower per sweep than the default path, ten times faster than the reference)\n"
    "  -chains N                      N independent chains, one per GPU, marginals pooled over RCCL (extension);\n"
    "                                 chains beyond the number of GPUs share a GPU and the construction it holds.\n"
    "                                 The pooled marginals / maxsegmentation files use common labels (states by\n"
    "                                 ascending mean); PREFIX[chainK.]relabelSUFFIX lists each chain's own label of\n"
    "                                 pooled state 0, 1, ... (its parameters / sequences files keep its own labels)\n"
    "  -v, -verbose   -g, -arguments   -h, -help\n";

// one entry of the sampling scheme (-i)
struct Step {
    string method;
    size_t iterations = 0, thinning = 0;
    bool incomplete = false;
};

// everything a chain needs besides the observations
struct Job {
    size_t T = 0, nrDataDim = 1, nrStates = 0, seed = 0;
    string opref, osuff;
    bool overwrite = false, useSelfTrans = true;
    real_t weightMultiplier = 1, trans = 0.5, selfTrans = 0.5, initialAlpha = 0.5;
    vector<vector<real_t>> thetaParams;
    vector<Step> scheme;
    std::map<string, bool> outputs;
};

// Meeting point of the chain threads of `-chains N` and the main thread: a chain arrives with its context once its
// scheme has run, the main thread pools the marginals of all of them (hml_allreduce_marginals) and lets them go on to
// write their files - or tells them not to when a chain failed.
class Rendezvous {
    std::mutex mMutex;
    std::condition_variable mCv;
    const int mExpected;
    vector<hml_ctx*> mCtx;
    int mAbandoned = 0;
    bool mReleased = false, mOk = false;

public:
    explicit Rendezvous(int n) : mExpected(n), mCtx(n, nullptr) {}
    // chain side: true = the marginals were pooled, write them
    bool arrive(int index, hml_ctx* ctx) {
        std::unique_lock<std::mutex> lock(mMutex);
        mCtx[index] = ctx;
        mCv.notify_all();
        mCv.wait(lock, [&] { return mReleased; });
        return mOk;
    }
    // a host thread that drives several chains: all of them arrive, one wait
    bool arrive(const vector<int>& indices, const vector<hml_ctx*>& ctxs) {
        std::unique_lock<std::mutex> lock(mMutex);
        for (size_t k = 0; k < indices.size(); ++k) mCtx[indices[k]] = ctxs[k];
        mCv.notify_all();
        mCv.wait(lock, [&] { return mReleased; });
        return mOk;
    }
    void abandon(int n = 1) {
        std::lock_guard<std::mutex> lock(mMutex);
        mAbandoned += n;
        mCv.notify_all();
    }
    // main side
    vector<hml_ctx*> waitForAll() {
        std::unique_lock<std::mutex> lock(mMutex);
        auto arrived = [&] { int n = 0; for (hml_ctx* c : mCtx) n += c != nullptr; return n; };
        mCv.wait(lock, [&] { return arrived() + mAbandoned >= mExpected; });
        vector<hml_ctx*> out;
        if (mAbandoned == 0) out = mCtx;
        return out;
    }
    void release(bool ok) {
        std::lock_guard<std::mutex> lock(mMutex);
        mReleased = true;
        mOk = ok;
        mCv.notify_all();
    }
};

// One chain from its device context to its output files.  `index` > 0 (chains of `-chains N` beyond the first): the
// per-sweep side files carry the infix "chainK." and the (pooled) marginals are left to chain 0.
// `source` (chains sharing a GPU): the context whose construction this chain attaches to (nullptr: it builds its own).
typedef Statistics<IntegralArray, Normal> StatsT;
typedef Blocks<BreakpointArray> BlocksT;
struct ChainRun {
    const Job& job;
    int index;
    bool verbose;
    rng_t RNG;
    Transitions<DirichletVector> A;
    Initial<Dirichlet> pi;
    TransitionHyperParam<DirichletParamVector> tau_A;
    InitialHyperParam<DirichletParam> tau_pi;
    Mapping mapping;
    Records records;
    std::unique_ptr<StatsT> ia;
    std::unique_ptr<BlocksT> waveletBlocks;
    std::unique_ptr<Emissions<StatsT, BlocksT>> y;
    std::unique_ptr<ThetaHyperParam<NormalInverseGammaParam>> tau_theta;
    std::unique_ptr<Theta<NormalInverseGamma>> theta;
    bool samplePrior = true, dynamic = true;

    ChainRun(const Job& job_, vector<real_t>& inputValues, bool steal, int device, uint32_t chainId, int index_, bool verbose_, hml_ctx* source)
        : job(job_), index(index_), verbose(verbose_), RNG((inputDevice() = device, job_.seed), device, chainId), A(job_.nrStates, RNG), pi(job_.nrStates, RNG),
          tau_A(job_.nrStates, job_.trans, job_.selfTrans), tau_pi(job_.nrStates, job_.initialAlpha),
          mapping(job_.nrDataDim, job_.thetaParams.size(), combinations),
          records(job_.T, index_ == 0 ? job_.opref : job_.opref + "chain" + std::to_string(index_) + ".", job_.osuff, job_.nrStates) {
        auto wants = [&](const char* o) { return job.outputs.at(o); };
        records.setRecordStateSequence(wants("sequences"), job.overwrite);
        records.setRecordTheta(wants("parameters"), job.overwrite);
        records.setRecordBlocks(wants("blocks"), job.overwrite);
        records.setRecordCompression(wants("compression"), job.overwrite);
        records.setRecordSegments(wants("segments"), job.overwrite);
        if (index == 0) {
            records.setRecordMarginals(wants("marginals"), job.overwrite);
            records.setRecordMaxSegmentation(wants("maxsegmentation"), job.overwrite);
        } else {
            records.setRecordMarginals(false);
            records.setAccumulateMarginals(wants("marginals") || wants("maxsegmentation"));   // for the pool
        }
        // upload + maxlet transform + weights + integral array (GPU); a lone chain takes the vector, several share it - and
        // chains that share a GPU share the construction itself
        if (source) ia.reset(new StatsT(source, job.T, job.nrDataDim, StatsT::attachInput));
        else ia.reset(steal ? new StatsT(inputValues, job.nrDataDim) : new StatsT(static_cast<const vector<real_t>&>(inputValues), job.nrDataDim, StatsT::keepInput));
        waveletBlocks.reset(new BlocksT(*ia));
        if (!source && job.weightMultiplier != 1) waveletBlocks->scaleWeights(job.weightMultiplier);   // (attached chains find the weights scaled)
    }
    // the rest of the set-up (after the construction is complete: an attaching chain may read it from now on)
    void model() {
        y.reset(new Emissions<StatsT, BlocksT>(*ia, *waveletBlocks));
        records.attach(y->ctx());
        vector<vector<real_t>> thetaParams = job.thetaParams;
        const double stdEstimate = ia->noiseEstimate();
        thetaParams[0] = autoPrior(thetaParams[0][0], thetaParams[0][1], *y, stdEstimate);
        for (auto& p : thetaParams) p = thetaParams[0];
        tau_theta.reset(new ThetaHyperParam<NormalInverseGammaParam>(thetaParams));
        theta.reset(new Theta<NormalInverseGamma>(*tau_theta, tau_A, tau_pi, job.useSelfTrans, RNG));
        if (verbose) cout << "Setting block structure to dynamic" << endl << flush;
    }
    // a token of the scheme up to (not including) the sweeps of "F" / "M" (reference main.cpp:383-452): a pending prior draw
    // happens when the next token starts, whatever it is.  Returns true when sweeps are to follow.
    bool token(const Step& st) {
        if (samplePrior) {
            if (verbose) cout << "Sampling prior" << endl << flush;
            hml_check(hml_sample_prior(RNG.ctx()));
            samplePrior = false;
        }
        if (st.method == "P") { samplePrior = true; return false; }
        if (st.method == "S") {
            if (verbose) cout << "Setting block structure to static" << endl << flush;
            y->createBlocks(*theta);
            dynamic = false;
            return false;
        }
        if (st.method == "D") {
            if (verbose) cout << "Setting block structure to dynamic" << endl << flush;
            dynamic = true;   // (sampleHMM switches the device back to per-sweep recompression)
            return false;
        }
        if (st.incomplete) throw std::runtime_error("Incomplete command line for -i!");
        if (st.method != "F" && st.method != "M") throw std::runtime_error("Unknown sampling type " + st.method + "!");
        if (verbose) cout << (st.method == "F" ? "Sampling Forward-Backward" : "Sampling mixture") << endl << flush;
        return true;
    }
    void sweeps(const Step& st) {
        if (st.method == "F") {
            StateSequence<ForwardBackward> q(RNG);
            sampleHMM(*y, q, *theta, *tau_theta, A, tau_A, pi, tau_pi, mapping, st.iterations, st.thinning, records, dynamic, job.useSelfTrans);
        } else {
            StateSequence<Mixture> q(RNG);
            sampleHMM(*y, q, *theta, *tau_theta, A, tau_A, pi, tau_pi, mapping, st.iterations, st.thinning, records, dynamic, job.useSelfTrans);
        }
    }
};

static void runChain(const Job& job, vector<real_t>& inputValues, bool steal, int device, uint32_t chainId, int index, bool verbose,
                     Rendezvous* rendezvous) {
    ChainRun run(job, inputValues, steal, device, chainId, index, verbose, nullptr);
    run.model();
    for (const Step& st : job.scheme)
        if (run.token(st)) run.sweeps(st);
    hml_check(hml_sync(run.RNG.ctx()));
    if (rendezvous && !rendezvous->arrive(index, run.RNG.ctx())) run.records.discardMarginals();   // pooling failed elsewhere
    run.records.close();
}

// The chains of `-chains N` that share ONE GPU, driven by one host thread in lockstep: the first builds the construction of the
// observations, the others attach to it (hml_attach_observations), and the sweeps of a scheme token run through
// hml_iterate_many - one set of launches for all of them where they are batched (include/hml.h).  Same files per chain as
// chain by chain.  indices[k] = the chain's index in the run (its Philox sub-key is `chain` + index).
static void runDeviceGroup(const Job& job, vector<real_t>& inputValues, int device, uint32_t chain, const vector<int>& indices, bool verbose,
                           Rendezvous* rendezvous) {
    vector<std::unique_ptr<ChainRun>> runs;
    for (size_t k = 0; k < indices.size(); ++k) {
        runs.emplace_back(new ChainRun(job, inputValues, /*steal*/ false, device, chain + (uint32_t)indices[k], indices[k], verbose && indices[k] == 0,
                                       k == 0 ? nullptr : runs[0]->RNG.ctx()));
        runs.back()->model();
    }
    for (const Step& st : job.scheme) {
        bool sweeps = false;
        for (auto& r : runs) sweeps = r->token(st) || sweeps;
        if (!sweeps) continue;
        const size_t n = runs.size();
        vector<Emissions<StatsT, BlocksT>*> ys(n);
        vector<Theta<NormalInverseGamma>*> thetas(n);
        vector<TransitionHyperParam<DirichletParamVector>*> tauAs(n);
        vector<InitialHyperParam<DirichletParam>*> tauPis(n);
        vector<Records*> recs(n);
        for (size_t k = 0; k < n; ++k) { ys[k] = runs[k]->y.get(); thetas[k] = runs[k]->theta.get(); tauAs[k] = &runs[k]->tau_A; tauPis[k] = &runs[k]->tau_pi; recs[k] = &runs[k]->records; }
        const bool dynamic = runs[0]->dynamic;
        if (st.method == "F") {
            vector<std::unique_ptr<StateSequence<ForwardBackward>>> qs;
            vector<StateSequence<ForwardBackward>*> qp(n);
            for (size_t k = 0; k < n; ++k) { qs.emplace_back(new StateSequence<ForwardBackward>(runs[k]->RNG)); qp[k] = qs.back().get(); }
            sampleHMMMany(ys, qp, thetas, tauAs, tauPis, st.iterations, st.thinning, recs, dynamic, job.useSelfTrans);
        } else {
            vector<std::unique_ptr<StateSequence<Mixture>>> qs;
            vector<StateSequence<Mixture>*> qp(n);
            for (size_t k = 0; k < n; ++k) { qs.emplace_back(new StateSequence<Mixture>(runs[k]->RNG)); qp[k] = qs.back().get(); }
            sampleHMMMany(ys, qp, thetas, tauAs, tauPis, st.iterations, st.thinning, recs, dynamic, job.useSelfTrans);
        }
    }
    vector<hml_ctx*> ctxs;
    for (auto& r : runs) { hml_check(hml_sync(r->RNG.ctx())); ctxs.push_back(r->RNG.ctx()); }
    const bool pooled = !rendezvous || rendezvous->arrive(indices, ctxs);
    for (auto& r : runs) { if (!pooled) r->records.discardMarginals(); r->records.close(); }
}

int main(int argc, const char* argv[]) {
    try {
        Parser args(argc, argv);
        args.registerFlags({"-v", "-verbose"});
        args.registerFlags({"-g", "-arguments"});
        args.registerFlags({"-h", "-help", "--help"});
        args.registerFlags({"-f", "-input-file"});
        args.registerFlags({"-o", "-output-pattern"}, "hammlet- .csv");
        args.registerFlags({"-O", "-output-data"}, "marginals");
        args.registerFlags({"-w", "-overwrite"});
        args.registerFlags({"-s", "-states"}, "3");
        args.registerFlags({"-e", "-emissions"}, "normal 0.2 0.9");
        args.registerFlags({"-a", "-auto-priors"});
        args.registerFlags({"-t", "-transitions"}, "0.5 0.5");
        args.registerFlags({"-S", "-no-self-transitions"});
        args.registerFlags({"-I", "-initial-dist"}, "0.5");
        args.registerFlags({"-R", "-random-seed"}, std::to_string(time(0)));
        args.registerFlags({"-i", "-iterations"}, "M 500 0 S P F 200 0 F 300 3");
        args.registerFlags({"-m", "-weight-multiplier"}, "1");
        // extensions (registered last so that `-g` prints the reference's lines first)
        args.registerFlags({"-raw"});
        args.registerFlags({"-device"}, "0");
        args.registerFlags({"-chain"}, "0");
        args.registerFlags({"-chains"}, "1");
        args.registerFlags({"-compat"});
        args.parseArgs();

        if (args.isSet("-g")) args.print();
        const bool verbose = args.isSet("-v");
        const bool overwrite = args.isSet("-w");
        if (args.isSet("-h")) {
            cout << endl << kHelp << endl;
            return 0;
        }

        // output pattern: without -o, "-f name.ext" yields "name-" ".ext"
        string opref, osuff;
        if (!args.isSet("-o") && args.isSet("-f")) {
            const string filename = args.parse<string>("-f");
            const size_t i = filename.find_last_of(".");
            opref = filename.substr(0, i) + "-";
            osuff = filename.substr(i);
        } else {
            opref = args.parse<string>("-o", 0);
            osuff = args.parse<string>("-o", 1);
        }

        const size_t rng_seed = args.parse<size_t>("-R", 0);
        // -compat: the reference-compatible mode of the library (include/hml.h, option "compat"): the reference's own
        // std::mt19937 stream, libm arithmetic and summation orders, so that -R SEED writes the reference's files
        if (args.isSet("-compat")) setenv("HML_COMPAT", "1", 1);
        const int device = args.parse<int>("-device");
        const uint32_t chain = args.parse<uint32_t>("-chain");
        const int nrChains = args.parse<int>("-chains");
        if (nrChains < 1) throw std::runtime_error("Number of chains must be positive!");

        // states: "-s K", or "-s C P D": P emission parameters shared by P^D states over D data dimensions whose values
        // follow each other in the input (reference main.cpp:114-137)
        size_t nrParams, nrDataDim = 1;
        if (args.nrTokens("-s") == 1) {
            nrParams = args.parse<size_t>("-s", 0);
        } else {
            const string m = args.parse<string>("-s", 0);
            if (m != "C" && m != "combinations") throw std::runtime_error("Unknown mapping type " + m + "!");
            nrParams = args.parse<size_t>("-s", 1);
            if (args.nrTokens("-s") >= 3) nrDataDim = args.parse<size_t>("-s", 2);
        }
        Mapping mapping(nrDataDim, nrParams, combinations);
        const size_t nrStates = mapping.nrStates();
        // (more than 16 states: the library's default path takes the number of states at run time there - hml_k_wide.h - up to 64)

        // first token = off-diagonal, second = diagonal (reference main.cpp:144-149)
        const real_t trans = args.parse<real_t>("-t", 0);
        real_t selfTrans = trans;
        if (args.nrTokens("-t") > 1) selfTrans = args.parse<real_t>("-t", 1);
        TransitionHyperParam<DirichletParamVector> tau_A(nrStates, trans, selfTrans);
        const bool useSelfTrans = !args.isSet("-S");
        const real_t initialAlpha = args.parse<real_t>("-I", 0);
        InitialHyperParam<DirichletParam> tau_pi(nrStates, initialAlpha);
        const real_t weightMultiplier = args.parse<real_t>("-m");

        vector<vector<real_t>> thetaParams;
        if (!args.isSet("-a")) throw std::runtime_error("Manual theta priors not implemented, use -a!");
        const vector<real_t> thp = args.parseVector<real_t>("-e", 1, 3);
        for (size_t i = 0; i < nrParams; ++i) thetaParams.push_back(thp);

        if (verbose) {
            cout << "Data dimensions: " << nrDataDim << endl;
            cout << "Emission distributions: " << nrParams << endl;
            cout << "States: " << nrStates << endl;
            string scheme;
            for (const string& t : args.tokens("-i")) scheme += (scheme.empty() ? "" : " ") + t;
            cout << "Sampling scheme: " << scheme << endl;
            cout << "Random seed: " << rng_seed << endl;
        }

        Parser outputArgs = args.subparser("-output-data");
        outputArgs.registerFlags({"M", "marginals"});
        outputArgs.registerFlags({"S", "sequences"});
        outputArgs.registerFlags({"P", "parameters"});
        outputArgs.registerFlags({"B", "blocks"});
        outputArgs.registerFlags({"C", "compression"});
        outputArgs.registerFlags({"D", "mapping"});
        outputArgs.registerFlags({"G", "segments"});
        outputArgs.registerFlags({"X", "maxsegmentation"});   // extension
        outputArgs.parseArgs();

        // ---- input
        inputDevice() = device;
        vector<real_t> inputValues;   // the observations (the device computes coefficients, weights and statistics)
        if (args.isSet("-raw")) {
            const string fname = args.parse<string>("-raw");
            std::ifstream fin(fname, std::ios::binary);
            if (!fin) throw std::runtime_error("Cannot read from input file " + fname + "!");
            fin.seekg(0, std::ios::end);
            const size_t n = (size_t)fin.tellg() / sizeof(float);
            fin.seekg(0);
            inputValues.resize(n);
            fin.read(reinterpret_cast<char*>(inputValues.data()), n * sizeof(float));
        } else if (args.isSet("-f")) {
            for (const string& fname : args.parseVector<string>("-f")) {
                if (verbose) cout << "Reading " + fname << endl << flush;
                std::ifstream fin(fname);
                if (!fin) throw std::runtime_error("Cannot read from input file " + fname + "!");
                // upper estimate of the number of values (the reference counts the lines, main.cpp:277): a value and
                // its separator take at least two bytes
                fin.seekg(0, std::ios::end);
                const std::streamoff bytes = fin.tellg();
                fin.seekg(0);
                readValues(fin, inputValues, nrDataDim, bytes > 0 ? (size_t)bytes / 2 + 1 : 0);
            }
        } else {
            if (verbose) cout << "Reading from standard input" << endl << flush;
            readValues(std::cin, inputValues, nrDataDim);
        }
        if (verbose) cout << "Output will be written to " + opref + "*" + osuff << endl << flush;
        // (the reference counts the coefficients, one per position; here the vector still holds the D values of every position)
        if (inputValues.size() % nrDataDim != 0)
            throw std::runtime_error("Input stream did not contain enough values to fill all dimensions at last position!");
        const size_t T = inputValues.size() / nrDataDim;
        if (verbose) cout << "Number of data points: " + std::to_string(T) << endl << flush;

        if (inputValues.empty()) throw std::runtime_error("Cannot compute Haar breakpoint weights, vector is empty!");
        if (verbose) cout << "Calculating Haar breakpoint weights" << endl << flush;

        // ---- sampling scheme, read once (reference main.cpp:364-377 validates the triples before anything runs; an
        // incomplete triple or an unknown method only fails when the loop reaches it, main.cpp:424-451)
        {
            size_t n = 0;
            for (const string& c : args.tokens("-i"))
                if (c != "P" && c != "S" && c != "D") n++;
            if (n % 3 != 0) throw std::runtime_error("Parameters for -i, excluding \"P\", \"S\" and \"D\", must be multiples of 3!");
        }
        vector<Step> scheme;
        {
            const size_t nrTokens = args.nrTokens("-i");
            for (size_t i = 0; i < nrTokens;) {
                Step st;
                st.method = args.parse<string>("-i", i);
                if (st.method == "P" || st.method == "S" || st.method == "D") { i++; }
                else if (i + 2 >= nrTokens) { st.incomplete = true; i = nrTokens; }
                else {
                    // (conversion errors surface here, before the first sweep; the reference parses them when the token is reached)
                    st.iterations = args.parse<size_t>("-i", i + 1);
                    st.thinning = args.parse<size_t>("-i", i + 2);
                    i += 3;
                }
                scheme.push_back(st);
            }
        }

        Job job;
        job.T = T; job.nrDataDim = nrDataDim; job.nrStates = nrStates; job.seed = rng_seed;
        job.opref = opref; job.osuff = osuff; job.overwrite = overwrite;
        job.weightMultiplier = weightMultiplier; job.useSelfTrans = useSelfTrans;
        job.thetaParams = thetaParams; job.trans = trans; job.selfTrans = selfTrans; job.initialAlpha = initialAlpha;
        job.scheme = scheme;
        for (const char* o : {"sequences", "parameters", "blocks", "compression", "marginals", "segments", "maxsegmentation"})
            job.outputs[o] = outputArgs.isSet(o);

        if (nrChains <= 1) {
            // the device context is created once every argument has been parsed and the input has been read
            runChain(job, inputValues, /*steal*/ true, device, chain, /*index*/ 0, verbose, nullptr);
        } else {
            // ---- chain-parallel (extension): chain k on GPU (device + k) mod #GPUs, each driven by its own host thread;
            // nothing is exchanged while sampling; the recorded marginals are pooled by one all-reduce (RCCL) at the end
            int nDev = 1;
            hml_check(hml_device_count(&nDev));
            Rendezvous rv(nrChains);
            // chain k lives on GPU (device + k) mod #GPUs; the chains of one GPU are driven by ONE host thread in lockstep and
            // share the construction of the observations
            std::map<int, vector<int>> byDevice;
            for (int k = 0; k < nrChains; ++k) byDevice[(device + k) % nDev].push_back(k);
            vector<std::thread> threads;
            vector<std::exception_ptr> errors(byDevice.size());
            size_t gi = 0;
            for (auto& kv : byDevice) {
                const int dev = kv.first;
                const vector<int> idx = kv.second;
                const size_t slot = gi++;
                threads.emplace_back([&, dev, idx, slot] {
                    try {
                        if (idx.size() == 1) runChain(job, inputValues, /*steal*/ false, dev, chain + (uint32_t)idx[0], idx[0], verbose && idx[0] == 0, &rv);
                        else runDeviceGroup(job, inputValues, dev, chain, idx, verbose, &rv);
                    } catch (...) {
                        errors[slot] = std::current_exception();
                        rv.abandon((int)idx.size());
                    }
                });
            }
            // all chains have sampled (or one has failed): pool, then let them write their files
            vector<hml_ctx*> ctxs = rv.waitForAll();
            std::exception_ptr poolError;
            // (nothing to pool when neither the marginals nor their arg-max segmentation were asked for)
            const bool wantsPool = job.outputs.at("marginals") || job.outputs.at("maxsegmentation");
            if ((int)ctxs.size() == nrChains && wantsPool) {
                if (verbose) cout << "Pooling the marginals of " << nrChains << " chains" << endl << flush;
                try {
                    // The pooled files (PREFIXmarginalsSUFFIX, PREFIXmaxsegmentationSUFFIX) use COMMON labels - states by
                    // ascending mean - while every chain's parameters / sequences / segments files keep the chain's own
                    // labels: PREFIX[chainK.]relabelSUFFIX holds, tab-separated, the chain's label of pooled state 0, 1, ...
                    vector<int32_t> perms((size_t)nrChains * job.nrStates);
                    hml_check(hml_allreduce_marginals_perm(ctxs.data(), nrChains, perms.data()));
                    for (int k = 0; k < nrChains; ++k) {
                        const string fn = (k == 0 ? job.opref : job.opref + "chain" + std::to_string(k) + ".") + "relabel" + job.osuff;
                        if (!job.overwrite) { std::ifstream probe(fn); if (probe.good()) throw std::runtime_error("File " + fn + " already exists!"); }
                        std::ofstream out(fn);
                        if (!out) throw std::runtime_error("Cannot open file " + fn + " for writing!");
                        for (size_t j = 0; j < job.nrStates; ++j) out << (j ? "\t" : "") << perms[(size_t)k * job.nrStates + j];
                        out << "\n";
                    }
                } catch (...) { poolError = std::current_exception(); }
            }
            rv.release(poolError == nullptr && (int)ctxs.size() == nrChains);
            for (auto& t : threads) t.join();
            for (auto& e : errors) if (e) std::rethrow_exception(e);
            if (poolError) std::rethrow_exception(poolError);
        }
        if (verbose) cout << "Exit HaMMLET" << endl << flush;
        return 0;
    } catch (std::exception& e) {
        cout << flush;
        cerr << endl << flush << "[ERROR] " << e.what() << endl;
        cerr << "Terminating HaMMLET. The rest is silence." << endl << flush;
        return 1;
    }
}
